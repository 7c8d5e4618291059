// nn_kernels.hip -- K2: the conv + 3xGRU + dense network (src/rnn.c:44-60) on the
// vector path: one 384-thread workgroup per stream, thread = output row / hidden unit,
// int8 weights consumed with v_dot4 (s8 x s8 plus a per-row offset that reproduces the
// x86 s8 x u8 accumulator exactly), float layers as one FMA chain per output in the
// reference's AVX2 order.  Follows oracle/rn_oracle.c (lin_float / lin_int8 / gru_step).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rn_dev.h"

#define NN_THREADS 384
typedef int v4i_t __attribute__((ext_vector_type(4)));

// ---- x86-profile activations (src/vec_avx.h:398-445) with the captured rcpps table ----
__device__ __forceinline__ float rcp_x86(float x, const uint16_t *__restrict__ lut) { return rn_rcp_x86(x, lut); }

__device__ __forceinline__ float tanh_x86(float x, const uint16_t *__restrict__ lut) {
  const float N0 = 952.52801514f, N1 = 96.39235687f, N2 = 0.60863042f;
  const float D0 = 952.72399902f, D1 = 413.36801147f, D2 = 11.88600922f;
  float x2 = x * x;
  float num = fmaf(fmaf(N2, x2, N1), x2, N0);
  float den = fmaf(fmaf(D2, x2, D1), x2, D0);
  num = num * x;
  den = rcp_x86(den, lut);
  num = num * den;
  num = (1.f < num) ? 1.f : num;
  return (-1.f > num) ? -1.f : num;
}

__device__ __forceinline__ float sigmoid_x86(float x, const uint16_t *__restrict__ lut) {
  const float N0 = 238.13200378f, N1 = 6.02452230f, N2 = 0.00950985f;
  const float D0 = 952.72399902f, D1 = 103.34200287f, D2 = 0.74287558f;
  float x2 = x * x;
  float num = fmaf(fmaf(N2, x2, N1), x2, N0);
  float den = fmaf(fmaf(D2, x2, D1), x2, D0);
  num = num * x;
  den = rcp_x86(den, lut);
  num = fmaf(num, den, .5f);
  num = (1.f < num) ? 1.f : num;
  return (0.f > num) ? 0.f : num;
}

// AVX2 vector_ps_to_epi8 (src/vec_avx.h:326-341) -> u8, then re-centred by -128 so that it
// fits the signed operand of v_dot4_i32_i8; the row offset 128*sum(w) restores s8 x u8.
__device__ __forceinline__ int quant_s8(float x) {
  float xf = fmaf(x, 127.f, 127.f);
  int xi = (xf >= -2147483648.f && xf < 2147483648.f) ? (int)rintf(xf) : INT32_MIN;
  int u16 = xi < 0 ? 0 : (xi > 65535 ? 65535 : xi);
  int s16 = (int)(int16_t)(uint16_t)u16;
  int u8 = s16 < 0 ? 0 : (s16 > 255 ? 255 : s16);
  return u8 - 128;
}

__device__ __forceinline__ int pack4(const float *x) {
  return (quant_s8(x[0]) & 0xff) | ((quant_s8(x[1]) & 0xff) << 8) | ((quant_s8(x[2]) & 0xff) << 16) |
         ((quant_s8(x[3]) & 0xff) << 24);
}

// one output row of a block-sparse (or dense, cols==null) int8 layer:
// float(acc_x86) * scale + subias   (src/vec_avx.h:778-877, src/nnet_arch.h:145-151)
__device__ __forceinline__ float int8_row(const RnLinearDev &l, int row, const int *xq) {
  const int grp = row >> 3, sub = row & 7;
  const int *wd = reinterpret_cast<const int *>(l.w);
  int acc = 0;
  if (l.cols) {
    const int b0 = l.grp_start[grp], b1 = l.grp_start[grp + 1];
    for (int b = b0; b < b1; b++) acc = __builtin_amdgcn_sdot4(wd[b * 8 + sub], xq[l.cols[b] >> 2], acc, false);
  } else {
    const int nb = l.nin >> 2;
    const int *wg = wd + (size_t)grp * nb * 8 + sub;
    for (int jb = 0; jb < nb; jb++) acc = __builtin_amdgcn_sdot4(wg[jb * 8], xq[jb], acc, false);
  }
  acc += l.rowsum128[row];
  return (float)acc * l.scale[row] + l.bias[row];
}

// NR output rows of int8 layers at once -- float(acc_x86) * scale + subias (src/vec_avx.h:778-877, src/nnet_arch.h:145-151).
// (the latency-oriented kernel below; the throughput kernel keeps int8_row: with two workgroups per CU its one-load-at-a-time
// rows overlap each other, and the byte extraction here costs it more issue slots than the wide loads save: 3.6 M against
// 4.5 M frames/s at 4096 streams.)  Row r comes from layer l[r] with the quantised input xq[r].  The accumulator is an integer, so the order of its terms is
// free: a row's blocks are read from the row-major copy (rn_dev.h: wrow / cq / grp4) four per 16-byte load, U chunks of all NR
// rows requested before the first dot product is issued.  (Thread = row with one block per 4-byte load and one load in
// flight took 107 us for one stream's network: first the L2 round trip per block, then -- with the loads batched -- the
// texture addresser, which spends as long on a 4-byte load as on a 16-byte one.)  Chunks past a row's end are clamped to its
// last chunk and enter with weight 0; padding blocks inside the last chunk are zero in the copy.
template <int NR, int U>
__device__ __forceinline__ void int8_rows(const RnLinearDev *const (&l)[NR], const int (&row)[NR], const int *const (&xq)[NR], float (&out)[NR]) {
  int nch[NR], acc[NR], maxn = 0;
  const v4i_t *wr[NR];
  const uint32_t *cq[NR];
#pragma unroll
  for (int r = 0; r < NR; r++) {
    const int grp = row[r] >> 3, g0 = l[r]->grp4[grp];
    nch[r] = l[r]->grp4[grp + 1] - g0;
    wr[r] = reinterpret_cast<const v4i_t *>(l[r]->wrow) + (g0 * 8 + (row[r] & 7) * nch[r]);
    cq[r] = l[r]->cq + g0;
    maxn = max(maxn, nch[r]);
    acc[r] = 0;
  }
  for (int i = 0; i < maxn; i += U) {
    v4i_t w[NR][U];
    uint32_t c[NR][U];
#pragma unroll
    for (int r = 0; r < NR; r++)
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int k = max(min(i + u, nch[r] - 1), 0);
        w[r][u] = wr[r][k];
        c[r][u] = cq[r][k];
      }
#pragma unroll
    for (int r = 0; r < NR; r++)
#pragma unroll
      for (int u = 0; u < U; u++) {
        const bool ok = i + u < nch[r];
#pragma unroll
        for (int j = 0; j < 4; j++)
          acc[r] = __builtin_amdgcn_sdot4(ok ? w[r][u][j] : 0, xq[r][(c[r][u] >> (8 * j)) & 0xff], acc[r], false);
      }
  }
#pragma unroll
  for (int r = 0; r < NR; r++) out[r] = (float)(acc[r] + l[r]->rowsum128[row[r]]) * l[r]->scale[row[r]] + l[r]->bias[row[r]];
}

struct NnLds {
  float tmp1[196];  // conv1 input  [t-2 | t-1 | t]   (src/nnet.c:118-119)
  float c1[128];
  float tmp2[384];  // conv2 input
  float cat[1536];  // conv2 out | gru1 | gru2 | gru3  (src/rnn.c:53-55)
  int xq[96];       // quantised layer input, 4 per dword
  int hq[96];       // quantised recurrent state
};

extern "C" __global__ void __launch_bounds__(NN_THREADS)
rn_nn_vector_kernel(RnGroupDev g, RnModelDev m, RnTablesDev tb) {
  __shared__ NnLds L;
  const int s = blockIdx.x, t = threadIdx.x;
  const uint16_t *lut = tb.rcp16;
  if (g.silence[s]) {  // src/denoise.c:474: the network and its state are untouched on silent frames
    if (t < RN_NB_BANDS) g.gains[(size_t)s * RN_NB_BANDS + t] = 0;
    if (t == 0) g.vad[s] = 0;
    return;
  }
  float *c1s = g.conv1_state + (size_t)s * 130;
  float *c2s = g.conv2_state + (size_t)s * 256;
  if (t < 130) L.tmp1[t] = c1s[t];
  if (t < 65) L.tmp1[130 + t] = g.features[(size_t)s * 68 + t];
  if (t >= 128) L.tmp2[t - 128] = c2s[t - 128];  // 256 history values
  __syncthreads();
  // conv1: float, 195 -> 128, one FMA chain per output (src/vec_avx.h:672-730), tanh
  if (t < 128) {
    float acc = 0;
    for (int j = 0; j < RN_CONV1_K; j++) acc = fmaf(m.conv1.fw[j * 128 + t], L.tmp1[j], acc);
    float v = tanh_x86(acc + m.conv1.bias[t], lut);
    L.c1[t] = v;
    L.tmp2[256 + t] = v;
  }
  if (t >= 192 && t < 192 + 130) c1s[t - 192] = L.tmp1[65 + t - 192];  // history <- last two frames
  __syncthreads();
  if (t < 96) L.xq[t] = pack4(L.tmp2 + 4 * t);
  if (t >= 128) c2s[t - 128] = L.tmp2[t];  // conv2 history <- tmp2[128..383]
  __syncthreads();
  // conv2: int8 dense 384 -> 384, tanh
  {
    float v = tanh_x86(int8_row(m.conv2, t, L.xq), lut);
    L.cat[t] = v;
  }
  __syncthreads();
  // three GRUs (src/nnet.c:65-94); thread = hidden unit
  for (int k = 0; k < 3; k++) {
    float *st = g.gru_state + ((size_t)k * g.n_stride + s) * RN_GRU;
    const float *xin = L.cat + k * RN_GRU;  // conv2 out, then the previous GRU's new state
    const float h_old = st[t];
    L.cat[(k + 1) * RN_GRU + t] = h_old;
    __syncthreads();
    if (t < 96) L.xq[t] = pack4(xin + 4 * t);
    else if (t >= 128 && t < 224) L.hq[t - 128] = pack4(L.cat + (k + 1) * RN_GRU + 4 * (t - 128));
    __syncthreads();
    const RnLinearDev &wi = m.gru_in[k], &wr = m.gru_rec[k];
    float zi = int8_row(wi, t, L.xq), ri = int8_row(wi, RN_GRU + t, L.xq), hi = int8_row(wi, 2 * RN_GRU + t, L.xq);
    float zr = int8_row(wr, t, L.hq), rr = int8_row(wr, RN_GRU + t, L.hq), hr = int8_row(wr, 2 * RN_GRU + t, L.hq);
    zr += wr.diag[t] * h_old;  // src/nnet_arch.h:153-161
    rr += wr.diag[RN_GRU + t] * h_old;
    hr += wr.diag[2 * RN_GRU + t] * h_old;
    float z = sigmoid_x86(zi + zr, lut);
    float r = sigmoid_x86(ri + rr, lut);
    float h = tanh_x86(hi + hr * r, lut);
    h = z * h_old + (1 - z) * h;
    __syncthreads();
    L.cat[(k + 1) * RN_GRU + t] = h;
    st[t] = h;
  }
  __syncthreads();
  // dense_out (1536 -> 32, FMA chains, sigmoid) and vad_dense (1536 -> 1: the scalar tail of
  // sgemv, src/vec_avx.h:732-736, unfused mul+add)
  if (t < RN_NB_BANDS) {
    float acc = 0;
    for (int j = 0; j < RN_CAT; j++) acc = fmaf(m.dense_out.fw[j * RN_NB_BANDS + t], L.cat[j], acc);
    g.gains[(size_t)s * RN_NB_BANDS + t] = sigmoid_x86(acc + m.dense_out.bias[t], lut);
  } else if (t == 64) {
    float acc = 0;
    for (int j = 0; j < RN_CAT; j++) acc = acc + m.vad_dense.fw[j] * L.cat[j];
    g.vad[s] = sigmoid_x86(acc + m.vad_dense.bias[0], lut);
  }
}

extern "C" hipError_t rn_launch_nn_vector(const RnGroupDev *g, const RnModelDev *m, const RnTablesDev *tb,
                                          hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  RN_LAUNCH(rn_nn_vector_kernel, dim3(g->n_streams), dim3(NN_THREADS), 0, st, e0, e1, *g, *m, *tb);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// K2 for a handful of streams (the one-stream states behind rnnoise_process_frame, include/rnnoise.h:94): the same network
// and the same bits as rn_nn_vector_kernel, arranged for LATENCY.  One workgroup of 7 waves per stream:
//   * waves 0-5 (thread = output row / hidden unit) run conv1 -> conv2 -> GRU x 3 with the prefetched row products above;
//   * wave 6 runs the two 1536-step chains of dense_out / vad_dense (serial by definition: one fmaf -- resp. mul + add -- per
//     input, in input order) BESIDE the GRU layers: the chain over cat segment s (conv2 output, then each GRU's new state)
//     runs while the next GRU layer is computed, so only the last 384 steps are exposed;
//   * every float weight a chain touches is in LDS before the chain needs it, brought there by LDS-DMA (no registers, no
//     instruction waits for it until its consumer does): conv1's 195 x 128 matrix by all waves at the start, vad_dense's 1536
//     weights and dense_out's 384 x 32 segments by wave 6, two segments ahead of their chain (two 48 KB buffers that take
//     over conv1's space).  A chain step is then an LDS read and an FMA, not an L2 round trip.
// ---------------------------------------------------------------------------------------------
#define ONE_THREADS 448
struct OneLds {
  NnLds n;
  float vadw[RN_CAT];
  float big[25088];  // conv1 weights (24,960 floats, 98 DMA pieces) at the start; then two buffers of 384 x 32 dense_out weights
};

// One LDS-DMA piece: 64 lanes x 16 bytes from per-lane global addresses to 1 KB of LDS at lds_dst (wave-uniform byte address).
// Issued from asm, so hipcc neither counts it nor waits for it: the waits are the explicit s_waitcnt vmcnt(0) below.
__device__ __forceinline__ void one_dma_1k(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}
__device__ __forceinline__ unsigned one_lds_addr(const void *p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const void *)p;
}

extern "C" __global__ void __launch_bounds__(ONE_THREADS)
rn_nn_one_kernel(RnGroupDev g, RnModelDev m, RnTablesDev tb) {
  extern __shared__ __attribute__((aligned(16))) char one_smem[];
  OneLds &O = *reinterpret_cast<OneLds *>(one_smem);
  NnLds &L = O.n;
  const int s = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const bool chain_wave = t >= RN_GRU;           // wave 6
  const int ct = t - RN_GRU;                     // its lane: 0..31 dense_out outputs, 32 the vad chain
  const uint16_t *lut = tb.rcp16;
  if (g.silence[s]) {  // src/denoise.c:474
    if (t < RN_NB_BANDS) g.gains[(size_t)s * RN_NB_BANDS + t] = 0;
    if (t == 0) g.vad[s] = 0;
    return;
  }
  float *c1s = g.conv1_state + (size_t)s * 130;
  float *c2s = g.conv2_state + (size_t)s * 256;
  {  // conv1 weights -> LDS: 98 pieces of 1 KB, 14 per wave (the last piece's tail lanes re-read the last 16 bytes)
    const char *src = reinterpret_cast<const char *>(m.conv1.fw);
    const unsigned dst = one_lds_addr(O.big);
    constexpr int last = RN_CONV1_K * 128 * 4 - 16;
#pragma unroll
    for (int i = 0; i < 14; i++) {
      const int piece = wave + 7 * i;
      one_dma_1k(src + min(piece * 1024 + lane * 16, last), dst + piece * 1024);
    }
    if (chain_wave) {
#pragma unroll
      for (int i = 0; i < RN_CAT * 4 / 1024; i++)
        one_dma_1k(reinterpret_cast<const char *>(m.vad_dense.fw) + i * 1024 + lane * 16, one_lds_addr(O.vadw) + i * 1024);
    }
    if (t < 130) L.tmp1[t] = c1s[t];
    if (t < 65) L.tmp1[130 + t] = g.features[(size_t)s * 68 + t];
    if (t >= 128 && t < RN_GRU) L.tmp2[t - 128] = c2s[t - 128];  // 256 history values
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  // conv1: float, 195 -> 128, one FMA chain per output (src/vec_avx.h:672-730), tanh
  if (t < 128) {
    float acc = 0;
#pragma unroll 15
    for (int j = 0; j < RN_CONV1_K; j++) acc = fmaf(O.big[j * 128 + t], L.tmp1[j], acc);
    const float v = tanh_x86(acc + m.conv1.bias[t], lut);
    L.c1[t] = v;
    L.tmp2[256 + t] = v;
  }
  if (t >= 192 && t < 192 + 130) c1s[t - 192] = L.tmp1[65 + t - 192];  // history <- last two frames
  __syncthreads();  // (conv1's weights have been consumed: both dense_out buffers are free)
  if (t < 96) L.xq[t] = pack4(L.tmp2 + 4 * t);
  if (t >= 128 && t < RN_GRU) c2s[t - 128] = L.tmp2[t];  // conv2 history <- tmp2[128..383]
  // dense_out weights of cat segment `seg` (384 x 32 floats, contiguous, 48 pieces) -> LDS buffer seg & 1, by wave 6
  auto fetch_segment = [&](int seg) {
    const char *src = reinterpret_cast<const char *>(m.dense_out.fw) + (size_t)seg * 49152 + lane * 16;
    const unsigned dst = one_lds_addr(O.big) + (seg & 1) * 49152;
#pragma unroll
    for (int i = 0; i < 48; i++) one_dma_1k(src + i * 1024, dst + i * 1024);
  };
  // the chains of wave 6 over one segment
  float dacc = 0, vacc = 0;
  auto chain_segment = [&](int seg) {
    const float *w = O.big + (seg & 1) * 12288, *x = L.cat + seg * RN_GRU, *vw = O.vadw + seg * RN_GRU;
    if (ct < RN_NB_BANDS) {
#pragma unroll 16
      for (int j = 0; j < RN_GRU; j++) dacc = fmaf(w[j * RN_NB_BANDS + ct], x[j], dacc);
    } else if (ct == RN_NB_BANDS) {  // the scalar tail of sgemv (src/vec_avx.h:732-736): unfused mul + add
#pragma unroll 16
      for (int j = 0; j < RN_GRU; j++) vacc = vacc + vw[j] * x[j];
    }
  };
  if (chain_wave) {
    fetch_segment(0);
    fetch_segment(1);
  }
  __syncthreads();
  // conv2: int8 dense 384 -> 384, tanh
  if (!chain_wave) {
    const RnLinearDev *const l1[1] = {&m.conv2};
    const int r1[1] = {t};
    const int *const x1[1] = {L.xq};
    float o1[1];
    int8_rows<1, 8>(l1, r1, x1, o1);
    L.cat[t] = tanh_x86(o1[0], lut);
  }
  // three GRUs (src/nnet.c:65-94); thread = hidden unit
#pragma unroll
  for (int k = 0; k < 3; k++) {
    float h_old = 0;
    float *st = g.gru_state + ((size_t)k * g.n_stride + s) * RN_GRU;
    if (!chain_wave) {
      h_old = st[t];
      L.cat[(k + 1) * RN_GRU + t] = h_old;
    }
    __syncthreads();  // segment k of cat is final (conv2 output / the previous layer's new state)
    if (t < 96) L.xq[t] = pack4(L.cat + k * RN_GRU + 4 * t);
    else if (t >= 128 && t < 224) L.hq[t - 128] = pack4(L.cat + (k + 1) * RN_GRU + 4 * (t - 128));
    __syncthreads();
    float h = 0;
    if (chain_wave) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's own DMA pieces: segment k's weights are in LDS
      chain_segment(k);
      if (k + 2 <= 3) fetch_segment(k + 2);            // into the buffer the chain has just left
    } else {
      const RnLinearDev &wi = m.gru_in[k], &wr = m.gru_rec[k];
      const RnLinearDev *const l6[6] = {&wi, &wi, &wi, &wr, &wr, &wr};
      const int r6[6] = {t, RN_GRU + t, 2 * RN_GRU + t, t, RN_GRU + t, 2 * RN_GRU + t};
      const int *const x6[6] = {L.xq, L.xq, L.xq, L.hq, L.hq, L.hq};
      float o6[6];
      int8_rows<6, 4>(l6, r6, x6, o6);
      const float zi = o6[0], ri = o6[1], hi = o6[2];
      float zr = o6[3], rr = o6[4], hr = o6[5];
      zr += wr.diag[t] * h_old;  // src/nnet_arch.h:153-161
      rr += wr.diag[RN_GRU + t] * h_old;
      hr += wr.diag[2 * RN_GRU + t] * h_old;
      const float z = sigmoid_x86(zi + zr, lut);
      const float r = sigmoid_x86(ri + rr, lut);
      h = tanh_x86(hi + hr * r, lut);
      h = z * h_old + (1 - z) * h;
    }
    __syncthreads();  // every reader of the old state (hq) is done
    if (!chain_wave) {
      L.cat[(k + 1) * RN_GRU + t] = h;
      st[t] = h;
    }
  }
  __syncthreads();
  if (chain_wave) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    chain_segment(3);
    if (ct < RN_NB_BANDS) g.gains[(size_t)s * RN_NB_BANDS + ct] = sigmoid_x86(dacc + m.dense_out.bias[ct], lut);
    else if (ct == RN_NB_BANDS) g.vad[s] = sigmoid_x86(vacc + m.vad_dense.bias[0], lut);
  }
}

extern "C" hipError_t rn_launch_nn_one(const RnGroupDev *g, const RnModelDev *m, const RnTablesDev *tb, hipStream_t st, hipEvent_t e0,
                                       hipEvent_t e1) {
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(rn_nn_one_kernel),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(OneLds));
  if (attr != hipSuccess) return attr;
  RN_LAUNCH(rn_nn_one_kernel, dim3(g->n_streams), dim3(ONE_THREADS), sizeof(OneLds), st, e0, e1, *g, *m, *tb);
  return hipGetLastError();
}

// MFMA path: see nn_mfma.hip
