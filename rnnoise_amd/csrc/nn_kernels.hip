// nn_kernels.hip -- K2: the conv + 3xGRU + dense network (src/rnn.c:44-60) on the
// vector path: one 384-thread workgroup per stream, thread = output row / hidden unit,
// int8 weights consumed with v_dot4 (s8 x s8 plus a per-row offset that reproduces the
// x86 s8 x u8 accumulator exactly), float layers as one FMA chain per output in the
// reference's AVX2 order.  Follows oracle/rn_oracle.c (lin_float / lin_int8 / gru_step).
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include "rn_dev.h"

#define NN_THREADS 384
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef float v4f_t __attribute__((ext_vector_type(4)));

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

// NR integer row products of int8 layers at once (the latency-oriented kernel below; the throughput kernel keeps int8_row:
// with two workgroups per CU its one-load-at-a-time rows overlap each other, and the byte extraction here costs it more issue
// slots than the wide loads save: 3.6 M against 4.5 M frames/s at 4096 streams).  Row r comes from layer l[r] with the quantised
// input xq[r]; acc[r] = sum over the row's blocks of w . x (add rowsum128, scale and bias with int8_finish).  The accumulator
// is an integer, so the order of its terms is free: a row's blocks are read from the row-major copy (rn_dev.h: wrow / cq /
// grp4) four per 16-byte load, U chunks of all NR rows requested before the first dot product is issued, and a row may be
// split between `nparts` threads (part p takes chunks [nch p / nparts, nch (p + 1) / nparts)).  Thread = row with one block
// per 4-byte load and one load in flight took 107 us for one stream's network: first the L2 round trip per block, then --
// with the loads batched -- the texture addresser, which spends as long on a 4-byte load as on a 16-byte one.
// Chunks past the range are clamped to its last chunk and dropped; padding blocks inside a group's last chunk are zero.
struct RowSrc {  // where a thread's rows come from: the row-major copy of one int8 layer + its epilogue vectors
  const int *grp4, *wrow, *rowsum128;
  const uint32_t *cq;
  const float *scale, *bias, *diag;
};
__device__ __forceinline__ RowSrc row_src(const RnLinearDev &l) { return RowSrc{l.grp4, l.wrow, l.rowsum128, l.cq, l.scale, l.bias, l.diag}; }
// (field-wise select: the layer structs live in kernel-argument memory, their fields arrive by scalar loads; selecting the
//  STRUCT per lane would turn every field access into a vector load from that memory)
__device__ __forceinline__ RowSrc row_src_select(bool second, const RnLinearDev &a, const RnLinearDev &b) {
  return RowSrc{second ? b.grp4 : a.grp4, second ? b.wrow : a.wrow, second ? b.rowsum128 : a.rowsum128, second ? b.cq : a.cq,
                second ? b.scale : a.scale, second ? b.bias : a.bias, second ? b.diag : a.diag};
}
// g0[r], g1[r] = grp4[row >> 3], grp4[(row >> 3) + 1], fetched by the caller AHEAD of the phase (with the epilogue vectors:
// every dependent global round trip costs one stream's network 1.5-2 us, and a GRU layer had eight of them in series --
// group bounds, weights, row sums / scales / biases, diagonal, three table lookups of the activations)
// The four lanes of a quad hold four rows of ONE 8-row group (callers map threads that way), so they walk the same chunks and
// need the same four activation dwords per chunk: each lane reads ONE of them from LDS (the block lane & 3 of the chunk) and
// the quad shares them by DPP broadcast -- a quarter of the LDS instructions.  (With one read per block the row waves kept
// the CU's LDS pipe busy for ~10 k cycles per GRU layer, and the dense_out / vad chains on their own waves, which live on
// LDS operands, crawled behind them: 32 k cycles per layer, THE critical path.)
template <int NR, int U>
__device__ __forceinline__ void int8_rows(const RowSrc &l, const int (&row)[NR], const int (&g0a)[NR], const int (&g1a)[NR], const int *xq,
                                          int part, int nparts, int (&acc)[NR]) {
  int n[NR], maxn = 0;
  const v4i_t *wr[NR];
  const uint32_t *cq[NR];
  const int qsh = 8 * (threadIdx.x & 3);
#pragma unroll
  for (int r = 0; r < NR; r++) {
    const int g0 = g0a[r], nch = g1a[r] - g0;
    const int lo = nch * part / nparts, hi = nch * (part + 1) / nparts;
    n[r] = hi - lo;
    wr[r] = reinterpret_cast<const v4i_t *>(l.wrow) + ((g0 + lo) * 8 + (row[r] & 7));
    cq[r] = l.cq + g0 + lo;
    maxn = max(maxn, n[r]);
    acc[r] = 0;
  }
  for (int i = 0; i < maxn; i += U) {  // (uniform within a quad: its four lanes share their groups)
    v4i_t w[NR][U];
    uint32_t c[NR][U];
    int x[NR][U];
#pragma unroll
    for (int r = 0; r < NR; r++)
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int k = max(min(i + u, n[r] - 1), 0);
        w[r][u] = wr[r][k * 8];
        c[r][u] = cq[r][k];
      }
#pragma unroll
    for (int r = 0; r < NR; r++)
#pragma unroll
      for (int u = 0; u < U; u++) x[r][u] = xq[(c[r][u] >> qsh) & 0xff];
#pragma unroll
    for (int r = 0; r < NR; r++)
#pragma unroll
      for (int u = 0; u < U; u++) {
        int t4 = 0;
        {
          const int xv = x[r][u];
          const v4i_t wv = w[r][u];
          // v_dot4c_i32_i8 is a VOP2: the quad broadcast rides on its first operand as a DPP modifier, one instruction per block
          // instead of three (zeroed register, v_mov_b32_dpp, v_dot4c; hipcc's DPP combiner does not form it).  From inline asm the
          // hazard recogniser does not see a DPP instruction, so the five wait states a DPP operand may need after a VALU or
          // EXEC write are spelled out (without them the kernel faulted).
          asm volatile("s_nop 4\n\tv_dot4c_i32_i8_dpp %0, %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                       "v_dot4c_i32_i8_dpp %0, %1, %3 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                       "v_dot4c_i32_i8_dpp %0, %1, %4 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                       "v_dot4c_i32_i8_dpp %0, %1, %5 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:0\n\ts_nop 1"
                       : "+v"(t4)
                       : "v"(xv), "v"(wv[0]), "v"(wv[1]), "v"(wv[2]), "v"(wv[3]));
        }
        acc[r] += (i + u < n[r]) ? t4 : 0;
      }
  }
}
// float(acc_x86) * scale + subias (src/vec_avx.h:778-877, src/nnet_arch.h:145-151), the three vectors prefetched
struct RowEpi { int rowsum; float scale, bias; };
__device__ __forceinline__ RowEpi row_epi(const RowSrc &l, int row) { return RowEpi{l.rowsum128[row], l.scale[row], l.bias[row]}; }
__device__ __forceinline__ float int8_finish(const RowEpi &e, int acc) { return (float)(acc + e.rowsum) * e.scale + e.bias; }

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
// K2 for a handful of streams (the one-stream states behind rnnoise_process_frame, include/rnnoise.h:94, and batches of up to
// 512 streams): the same network and the same bits as rn_nn_vector_kernel, arranged for LATENCY.  One workgroup of 14 waves
// per stream:
//   * waves 0-5: thread u = the three input-matrix rows of hidden unit u (and conv2's row u, and the unit's gates); waves 6-11:
//     thread u = the three recurrent rows (+ diagonal), handed to the gate thread through LDS.  Row products come from the
//     row-major int8 copy, four blocks per 16-byte load; a quad's four lanes are four rows of one group and share their
//     activation reads; what a layer needs besides weights (old state, group bounds, row sums / scales / biases, diagonal)
//     is requested one layer ahead; the rcpps table of the activations lives in LDS;
//   * wave 12 runs the 32 dense_out chains, wave 13 the vad_dense chain (serial by definition: one fmaf -- resp. mul + add --
//     per input, in input order) BESIDE the GRU layers: the chain over cat segment s (conv2 output, then each GRU's new state)
//     runs while the next GRU layer is computed, so only the last 384 steps are exposed; operands sixteen steps at a time,
//     the next sixteen in flight;
//   * every float weight a chain touches is in LDS before the chain needs it, brought there by LDS-DMA (no registers, no
//     instruction waits for it until its consumer does): conv1's 195 x 128 matrix by all waves at the start, vad_dense's 1536
//     weights by wave 13, dense_out's 384 x 32 segments by wave 12, two segments ahead of their chain (two 48 KB buffers that
//     take over conv1's space).  A chain step is then an LDS read and an FMA, not an L2 round trip.
// 107 us (thread = row, one 4-byte load in flight) -> 40 us for one stream; where the time went is in DESIGN.md section 9.
// ---------------------------------------------------------------------------------------------
#define ONE_ROW_THREADS 768   // 12 waves of row threads
#define ONE_THREADS 896       // + wave 12: the dense_out chains (lane = output), wave 13: the vad chain
struct OneLds {
  NnLds n;
  uint16_t lut[4096];  // rcpps table (rn_dev.h: rcp16)
  float ex[3][RN_GRU];  // a layer's recurrent sums (z, r, h rows), handed from the recurrent-row threads to the gate threads
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
// the value of lane ^ 1 (quad_perm [1, 0, 3, 2])
__device__ __forceinline__ int one_pair_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false); }
__device__ __forceinline__ float one_pair_f(float v) { return __int_as_float(one_pair_i(__float_as_int(v))); }

extern "C" __global__ void __launch_bounds__(ONE_THREADS)
rn_nn_one_kernel(RnGroupDev g, RnModelDev m, RnTablesDev tb, RnRows rows) {
  extern __shared__ __attribute__((aligned(16))) char one_smem[];
  OneLds &O = *reinterpret_cast<OneLds *>(one_smem);
  NnLds &L = O.n;
  // (rows: a launch group of the one-frame API, rn_dev.h -- the block's pool row, its VAD goes to the row's pinned frame block)
  const bool listed = rows.n > 0;
  const int s = listed ? RN_ROW_OF(rows.e[blockIdx.x]) : (int)blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  float *vad_dst = listed ? rows.io + (size_t)s * RN_ROW_IO + 2 * RN_FRAME_SIZE + 4 : g.vad + s;
  const bool chain_wave = t >= ONE_ROW_THREADS;  // waves 12 (dense_out) and 13 (vad_dense)
  const bool vad_wave = wave == 13;
  const int ct = lane & 31;                      // wave 12: dense_out output of this lane (the upper half repeats the lower)
  // row threads: waves 0-5 the rows of the input matrices (and conv2), waves 6-11 those of the recurrent matrices; a quad = four
  // consecutive rows of one 8-row group (int8_rows shares their activation reads)
  const int half = t >= RN_GRU && t < ONE_ROW_THREADS, u = t - (t >= RN_GRU ? RN_GRU : 0);
  const uint16_t *lut = O.lut;  // the rcpps table in LDS: three dependent lookups per unit and layer must not be L2 trips
  if (g.silence[s]) {  // src/denoise.c:474
    if (t < RN_NB_BANDS) g.gains[(size_t)s * RN_NB_BANDS + t] = 0;
    if (t == 0) *vad_dst = 0;
    return;
  }
  float *c1s = g.conv1_state + (size_t)s * 130;
  float *c2s = g.conv2_state + (size_t)s * 256;
  // (instrumented build: shader-clock taps of thread 0 -> slots RN_DBG_CLK2 + 0..7, of the chain wave -> + 8..13; tools/nn_one_taps.py)
#if RN_INSTRUMENT
  float *dbg = (g.debug && (t == 0 || t == ONE_ROW_THREADS)) ? g.debug + (size_t)s * RN_DBG_FLOATS + RN_DBG_CLK2 + (t ? 8 : 0) : nullptr;
  unsigned long long clk_prev = g.debug ? __builtin_amdgcn_s_memtime() : 0;
  int tap_i = 0;
#define ONE_TAP() do { if (g.debug) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); if (dbg && tap_i < 8) dbg[tap_i] = (float)(now_ - clk_prev); tap_i++; clk_prev = now_; } } while (0)
#else
#define ONE_TAP() do { } while (0)
#endif
  {  // conv1 weights -> LDS: 98 pieces of 1 KB over 13 waves (the last piece's tail lanes re-read the last 16 bytes)
    const char *src = reinterpret_cast<const char *>(m.conv1.fw);
    const unsigned dst = one_lds_addr(O.big);
    constexpr int last = RN_CONV1_K * 128 * 4 - 16;
#pragma unroll
    for (int i = 0; i < 7; i++) {
      const int piece = wave + 14 * i;  // wave-uniform; 14 x 7 = 98
      one_dma_1k(src + min(piece * 1024 + lane * 16, last), dst + piece * 1024);
    }
    if (vad_wave) {
#pragma unroll
      for (int i = 0; i < RN_CAT * 4 / 1024; i++)
        one_dma_1k(reinterpret_cast<const char *>(m.vad_dense.fw) + i * 1024 + lane * 16, one_lds_addr(O.vadw) + i * 1024);
    } else if (wave < 8) {
      one_dma_1k(reinterpret_cast<const char *>(tb.rcp16) + wave * 1024 + lane * 16, one_lds_addr(O.lut) + wave * 1024);
    }
    if (t < 130) L.tmp1[t] = c1s[t];
    if (t < 65) L.tmp1[130 + t] = g.features[(size_t)s * 68 + t];
    if (t >= 128 && t < RN_GRU) L.tmp2[t - 128] = c2s[t - 128];  // 256 history values
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  ONE_TAP();  // 0: weights of conv1 and the inputs are in LDS
  // (conv2's group bounds and epilogue vectors start their trip now: consumed two barriers later)
  const RowSrc c2 = row_src(m.conv2);
  const bool c2_thread = t < RN_GRU;  // (conv2: one thread per row; the recurrent-row waves sit it out)
  const int c2g0 = c2_thread ? c2.grp4[u >> 3] : 0, c2g1 = c2_thread ? c2.grp4[(u >> 3) + 1] : 0;
  const RowEpi c2e = c2_thread ? row_epi(c2, u) : RowEpi{0, 0.f, 0.f};
  // conv1: float, 195 -> 128, one FMA chain per output (src/vec_avx.h:672-730), tanh
  if (t < 128) {
    const float b1 = m.conv1.bias[t];
    float acc = 0;
#pragma unroll 39
    for (int j = 0; j < RN_CONV1_K; j++) acc = fmaf(O.big[j * 128 + t], L.tmp1[j], acc);
    const float v = tanh_x86(acc + b1, lut);
    L.c1[t] = v;
    L.tmp2[256 + t] = v;
  }
  if (t >= 192 && t < 192 + 130) c1s[t - 192] = L.tmp1[65 + t - 192];  // history <- last two frames
  __syncthreads();  // (conv1's weights have been consumed: both dense_out buffers are free)
  ONE_TAP();  // 1: conv1
  if (t < 96) L.xq[t] = pack4(L.tmp2 + 4 * t);
  if (t >= 128 && t < RN_GRU) c2s[t - 128] = L.tmp2[t];  // conv2 history <- tmp2[128..383]
  // dense_out weights of cat segment `seg` (fw4 order, 48 KB contiguous = 48 pieces) -> LDS buffer seg & 1.  Segment 0 by the
  // dense_out wave during conv2; segment k + 1 by the twelve row waves, four pieces each, at the start of GRU layer k (its
  // buffer was last read by chain k - 1, one barrier ago).  (One wave issuing all 48 pieces after its chain spent 7 k cycles
  // in the issue alone -- the memory pipeline takes a 1 KB piece every ~150 cycles from one wave -- and everybody waited for it
  // at the layer's barrier.)
  // (lane-number derivatives that are needed once per layer, or once at the end, are re-derived from the thread id through an opaque
  //  copy where they are used: kept in registers from the top they pushed the kernel 5 registers over the 128 its 14 waves allow)
  auto late_lane = [&] {
    int v = (int)threadIdx.x;
    asm volatile("" : "+v"(v));
    return v & 63;
  };
  auto fetch_pieces = [&](int seg, int first, int count) {
    const char *src = reinterpret_cast<const char *>(m.dense_out.fw4) + (size_t)seg * 49152 + late_lane() * 16;
    const unsigned dst = one_lds_addr(O.big) + (seg & 1) * 49152;
    for (int i = first; i < first + count; i++) one_dma_1k(src + i * 1024, dst + i * 1024);
  };
  // The chains over one segment of cat, 384 steps each, wave-uniform code (wave 12: lanes = the 32 dense_out outputs, one
  // fmaf per step; wave 13: every lane the same vad chain, the scalar tail of sgemv, src/vec_avx.h:732-736: unfused mul + add).
  // Operands come from LDS sixteen steps at a time, the next sixteen requested before the current ones are consumed: left to
  // the compiler every step waited out its own LDS round trip (48 cycles x 768 steps per layer -- THE critical path of the
  // first version, longer than the GRU rows beside it).
  float cacc = 0;
  // (two copies of the loop behind a wave-uniform branch: with the mul + add / fmaf choice INSIDE the loop the compiler computed
  //  both and selected per step -- a v_cndmask on VCC in the dependency chain, 16-19 clocks each: 73 cycles per step)
  // w4: the chain's weights, four consecutive steps per 16 bytes (dense_out: rn_dev.h fw4 order, one 16-byte slot per output
  // and group of four inputs; vad_dense: its 1536 weights as they are, the same slot for every lane)
  auto chain_steps = [&](const v4f_t *w4, int ws4, const float *x, auto step) {
    v4f_t wa[4], wb[4], xa[4], xb[4];
    auto fetch = [&](v4f_t (&wv)[4], v4f_t (&xv)[4], int b) {
#pragma unroll
      for (int i = 0; i < 4; i++) wv[i] = w4[(4 * b + i) * ws4];
#pragma unroll
      for (int i = 0; i < 4; i++) xv[i] = *reinterpret_cast<const v4f_t *>(x + 16 * b + 4 * i);
    };
    auto run = [&](const v4f_t (&wv)[4], const v4f_t (&xv)[4]) {
#pragma unroll
      for (int i = 0; i < 16; i++) cacc = step(wv[i >> 2][i & 3], xv[i >> 2][i & 3], cacc);
    };
    fetch(wa, xa, 0);
#pragma unroll 1
    for (int b = 0; b < RN_GRU / 16; b += 2) {  // (fenced: the scheduler otherwise sinks every fetch to its first use)
      fetch(wb, xb, b + 1);
      __builtin_amdgcn_sched_barrier(0);
      run(wa, xa);
      __builtin_amdgcn_sched_barrier(0);
      if (b + 2 < RN_GRU / 16) fetch(wa, xa, b + 2);
      __builtin_amdgcn_sched_barrier(0);
      run(wb, xb);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto chain_segment = [&](int seg) {
    const float *x = L.cat + seg * RN_GRU;
    if (vad_wave) chain_steps(reinterpret_cast<const v4f_t *>(O.vadw + seg * RN_GRU), 1, x, [](float w, float xv, float a) { return a + w * xv; });
    else chain_steps(reinterpret_cast<const v4f_t *>(O.big + (seg & 1) * 12288) + ct, RN_NB_BANDS, x, [](float w, float xv, float a) { return fmaf(w, xv, a); });
  };
  if (chain_wave && !vad_wave) fetch_pieces(0, 0, 48);
  __syncthreads();
  // what a GRU layer reads from global memory besides its weights -- old state, group bounds, epilogue vectors, diagonal --
  // is requested one layer ahead (here for layer 0; inside layer k for k + 1): a dependent round trip costs 1.5-3.5 us
  struct Smalls {
    float h_old;
    int g0[3], g1[3];
    RowEpi e[3];
    float dg[3];
  };
  auto smalls_fetch = [&](int k) {
    Smalls q = {};
    if (!chain_wave) {
      const RowSrc wm = row_src_select(half, m.gru_in[k], m.gru_rec[k]);
      q.h_old = g.gru_state[((size_t)k * g.n_stride + s) * RN_GRU + u];
#pragma unroll
      for (int r = 0; r < 3; r++) {
        const int row = r * RN_GRU + u;
        q.g0[r] = wm.grp4[row >> 3];
        q.g1[r] = wm.grp4[(row >> 3) + 1];
        q.e[r] = row_epi(wm, row);
        if (half) q.dg[r] = wm.diag[row];
      }
    }
    return q;
  };
  Smalls sm = smalls_fetch(0);
  // conv2: int8 dense 384 -> 384, tanh
  if (c2_thread) {
    const int r1[1] = {u}, g0[1] = {c2g0}, g1[1] = {c2g1};
    int a1[1];
    int8_rows<1, 6>(c2, r1, g0, g1, L.xq, 0, 1, a1);
    L.cat[u] = tanh_x86(int8_finish(c2e, a1[0]), lut);
  }
  ONE_TAP();  // 2: conv2 (rows) / chain wave: segments 0 and 1 requested
  // three GRUs (src/nnet.c:65-94): thread u of waves 0-5 the three input-matrix rows of hidden unit u and, after the exchange,
  // its gates; thread u of waves 6-11 the three recurrent rows (+ diagonal), handed over through LDS
  if (chain_wave) __builtin_amdgcn_s_setprio(3);  // (the chains are one long dependency: let their LDS reads and FMAs issue first)
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float h_old = sm.h_old;
    float *st = g.gru_state + ((size_t)k * g.n_stride + s) * RN_GRU;
    if (!chain_wave && !half) L.cat[(k + 1) * RN_GRU + u] = h_old;
    __syncthreads();  // segment k of cat is final (conv2 output / the previous layer's new state)
    if (k == 0) ONE_TAP();  // 3: layer 0: barrier
    {  // (the thread id through an opaque copy: see late_lane)
      int tp = (int)threadIdx.x;
      asm volatile("" : "+v"(tp));
      if (tp < 96) L.xq[tp] = pack4(L.cat + k * RN_GRU + 4 * tp);
      else if (tp >= 128 && tp < 224) L.hq[tp - 128] = pack4(L.cat + (k + 1) * RN_GRU + 4 * (tp - 128));
    }
    __syncthreads();
    if (k == 0) ONE_TAP();  // 4: layer 0: quantised inputs packed, barrier
#if RN_INSTRUMENT
    const unsigned long long phase_clk = g.debug ? __builtin_amdgcn_s_memtime() : 0;
#endif
    Smalls nx = {};
    int a3[3] = {0, 0, 0};
    if (chain_wave) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's own DMA pieces: segment k's weights are in LDS
      if (k == 0) ONE_TAP();  // chain wave 5: layer 0: waited for its DMA
      chain_segment(k);
      if (k == 0) ONE_TAP();  // chain wave 6: layer 0: chain segment
    } else {
      fetch_pieces(k + 1, 4 * wave, 4);  // (asm-issued: drained by the explicit wait below, before the layer's barrier)
      if (k < 2) nx = smalls_fetch(k + 1);
      const RowSrc wm = row_src_select(half, m.gru_in[k], m.gru_rec[k]);
      const int r3[3] = {u, RN_GRU + u, 2 * RN_GRU + u};
      int8_rows<3, 3>(wm, r3, sm.g0, sm.g1, half ? L.hq : L.xq, 0, 1, a3);
      if (k == 0) ONE_TAP();  // 5: layer 0: row products
      if (half) {  // recurrent sums + diagonal (src/nnet_arch.h:153-161) -> LDS
#pragma unroll
        for (int r = 0; r < 3; r++) O.ex[r][u] = int8_finish(sm.e[r], a3[r]) + sm.dg[r] * h_old;
      }
    }
    if (!chain_wave) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the next segment have landed
#if RN_INSTRUMENT
    if (k == 0 && g.debug && lane == 0)  // arrival of every wave at layer 0's exchange barrier, clocks since its pack barrier
      g.debug[(size_t)s * RN_DBG_FLOATS + RN_DBG_CLK2 + 16 + wave] = (float)(__builtin_amdgcn_s_memtime() - phase_clk);
#endif
    __syncthreads();  // the recurrent sums are in LDS (and every reader of the old state image is long done)
    if (k == 0) ONE_TAP();  // 6: layer 0: exchange barrier (waits for the slowest wave, the chains included)
    if (!chain_wave && !half) {
      const float zi = int8_finish(sm.e[0], a3[0]), ri = int8_finish(sm.e[1], a3[1]), hi = int8_finish(sm.e[2], a3[2]);
      const float z = sigmoid_x86(zi + O.ex[0][u], lut);
      const float r = sigmoid_x86(ri + O.ex[1][u], lut);
      float h = tanh_x86(hi + O.ex[2][u] * r, lut);
      h = z * h_old + (1 - z) * h;
      L.cat[(k + 1) * RN_GRU + u] = h;
      st[u] = h;
    }
    if (k == 0) ONE_TAP();  // 7: layer 0: gates
    sm = nx;
  }
  __syncthreads();
  if (chain_wave) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    chain_segment(3);
    const int ll = late_lane(), lct = ll & 31;
    if (!vad_wave && ll < RN_NB_BANDS) g.gains[(size_t)s * RN_NB_BANDS + lct] = sigmoid_x86(cacc + m.dense_out.bias[lct], lut);
    else if (vad_wave && ll == 0) *vad_dst = sigmoid_x86(cacc + m.vad_dense.bias[0], lut);
  }
  ONE_TAP();  // 6: last chain segment + outputs
#undef ONE_TAP
}

// the 125 KB of dynamic LDS are an opt-in per DEVICE (a process may hold pools and batches on several GPUs): once per device,
// on the device the launch goes to
static hipError_t nn_one_opt_in() {
  static std::atomic<int> opted[64];
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64 || !opted[dev].load(std::memory_order_acquire)) {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(rn_nn_one_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)sizeof(OneLds));
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) opted[dev].store(1, std::memory_order_release);
  }
  return hipSuccess;
}
extern "C" hipError_t rn_launch_nn_one(const RnGroupDev *g, const RnModelDev *m, const RnTablesDev *tb, hipStream_t st, hipEvent_t e0,
                                       hipEvent_t e1) {
  const hipError_t attr = nn_one_opt_in();
  if (attr != hipSuccess) return attr;
  RN_LAUNCH(rn_nn_one_kernel, dim3(g->n_streams), dim3(ONE_THREADS), sizeof(OneLds), st, e0, e1, *g, *m, *tb, RnRows{});
  return hipGetLastError();
}
// K2 of a launch group of the one-frame API (rn_dev.h: RnRows): one workgroup per listed row
extern "C" hipError_t rn_launch_nn_rows(const RnGroupDev *g, const RnModelDev *m, const RnTablesDev *tb, const RnRows *rows, hipStream_t st) {
  const hipError_t attr = nn_one_opt_in();
  if (attr != hipSuccess) return attr;
  hipLaunchKernelGGL(rn_nn_one_kernel, dim3(rows->n), dim3(ONE_THREADS), sizeof(OneLds), st, *g, *m, *tb, *rows);
  return hipGetLastError();
}

// MFMA path: see nn_mfma.hip
