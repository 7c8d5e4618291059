// nn_mfma.hip -- K2, batched MFMA path: the network (src/rnn.c:44-60) recast as
// (16 streams x K) . (K x outputs) matrix products, one workgroup (4 waves) per tile of 16
// streams.
//
//   int8 layers (conv2, 6 GRU matrices): v_mfma_i32_16x16x64_i8 on the block-sparse weights
//     zero-filled to dense and pre-swizzled into A-fragment order (shim.cpp: stage_mfma), the
//     activations quantised exactly like the x86 path (u8, src/vec_avx.h:326-341) and
//     re-centred to s8 = u8-128; acc_x86 = acc_mfma + 128*rowsum(w).  Integer => exact.
//   float layers (conv1, dense_out): v_mfma_f32_16x16x4_f32, which on gfx950 is bitwise a
//     k-ordered fmaf chain (tools/mfma_probe.hip, cdna_hip_programming.md section 3) = the AVX2
//     sgemv order (src/vec_avx.h:672-730).  vad_dense is the unfused scalar tail
//     (vec_avx.h:732-736) and stays on the VALU, lane = stream.
//
// Fragment maps (verified on hardware by the probe): A lane l -> row l&15, k-group l>>4;
// B lane l -> column (stream) l&15, k-group l>>4; C/D lane l, reg r -> row 4*(l>>4)+r, col l&15.
// Results are bit-identical to the vector path and to the oracle.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rn_dev.h"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define TS 16          // streams per workgroup
#define NWAVES 8       // 2 waves per SIMD: one wave's weight loads / epilogue overlap the other's MFMAs
#define NTHREADS (64 * NWAVES)
#define KT 6           // 384 / 64 k-tiles of every int8 layer
#define CHUNK 256      // inputs per staged chunk of the dense_out / vad chains
#define CH_STRIDE 260  // floats per stream and chunk in LDS (16-byte aligned rows, 2-way bank conflicts at most)

// ---- x86-profile activations (same arithmetic as nn_kernels.hip; LUT staged in LDS) ----
__device__ __forceinline__ float rcp_x86(float x, const uint32_t *lut) {
  uint32_t b = __float_as_uint(x);
  return __uint_as_float(lut[(b >> 12) & 0x7ff] - ((b & 0x7f800000u) - 0x3f800000u));
}
__device__ __forceinline__ float tanh_x86(float x, const uint32_t *lut) {  // src/vec_avx.h:398-416
  const float N0 = 952.52801514f, N1 = 96.39235687f, N2 = 0.60863042f;
  const float D0 = 952.72399902f, D1 = 413.36801147f, D2 = 11.88600922f;
  float x2 = x * x;
  float num = fmaf(fmaf(N2, x2, N1), x2, N0);
  float den = fmaf(fmaf(D2, x2, D1), x2, D0);
  num = num * x;
  num = num * rcp_x86(den, lut);
  num = (1.f < num) ? 1.f : num;
  return (-1.f > num) ? -1.f : num;
}
__device__ __forceinline__ float sigmoid_x86(float x, const uint32_t *lut) {  // src/vec_avx.h:426-445
  const float N0 = 238.13200378f, N1 = 6.02452230f, N2 = 0.00950985f;
  const float D0 = 952.72399902f, D1 = 103.34200287f, D2 = 0.74287558f;
  float x2 = x * x;
  float num = fmaf(fmaf(N2, x2, N1), x2, N0);
  float den = fmaf(fmaf(D2, x2, D1), x2, D0);
  num = num * x;
  num = fmaf(num, rcp_x86(den, lut), .5f);
  num = (1.f < num) ? 1.f : num;
  return (0.f > num) ? 0.f : num;
}
__device__ __forceinline__ int quant_s8(float x) {  // src/vec_avx.h:326-341, then -128
  float xf = fmaf(x, 127.f, 127.f);
  int xi = (xf >= -2147483648.f && xf < 2147483648.f) ? (int)rintf(xf) : INT32_MIN;
  int u16 = xi < 0 ? 0 : (xi > 65535 ? 65535 : xi);
  int s16 = (int)(int16_t)(uint16_t)u16;
  int u8 = s16 < 0 ? 0 : (s16 > 255 ? 255 : s16);
  return u8 - 128;
}
__device__ __forceinline__ int pack4(float a, float b, float c, float d) {
  return (quant_s8(a) & 0xff) | ((quant_s8(b) & 0xff) << 8) | ((quant_s8(c) & 0xff) << 16) | ((quant_s8(d) & 0xff) << 24);
}

// byte offset of activation (stream n, input k) inside a B-fragment-ordered buffer:
// [k/64][lane = n + 16*((k%64)/16)][k%16]
__device__ __forceinline__ int frag_off(int n, int k) { return (((k >> 6) * 64 + n + 16 * ((k >> 4) & 3)) << 4) + (k & 15); }

struct MfmaLds {
  uint32_t lut[2048];             // rcpps table
  float vadw[RN_CAT];             // vad_dense weights: the lane = stream chain of wave 2 must not wait for L2 at every step
  union {
    struct {
      float tmp1[TS][197];          // conv1 input [t-2|t-1|t], padded row
      int8_t xq[2][KT * 64 * 16];   // quantised layer input, B-fragment order (double buffer)
      int8_t hq[KT * 64 * 16];      // quantised recurrent state
    };
    float stage[2][TS][CH_STRIDE];  // dense_out / vad phase: f32 activations, 128 inputs per chunk, double buffer
  };
};

// one int8 output-row tile: 6 MFMAs over K=384, A straight from the pre-swizzled weights
// (B fragments are re-read from LDS per use -- conflict-free 16-byte reads -- rather than held in 48 VGPRs)
__device__ __forceinline__ v4i int8_tile(const int8_t *__restrict__ wmf, int rt, int lane, const int8_t *bq) {
  v4i acc = {0, 0, 0, 0};
  const v4i *a = reinterpret_cast<const v4i *>(wmf) + (size_t)rt * KT * 64 + lane;
  const v4i *b = reinterpret_cast<const v4i *>(bq) + lane;
#pragma unroll
  for (int kt = 0; kt < KT; kt++) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[kt * 64], b[kt * 64], acc, 0, 0, 0);
  return acc;
}

// float(acc_x86)*scale + subias for the 4 rows a lane owns (src/nnet_arch.h:145-151)
__device__ __forceinline__ v4f int8_finish(const RnLinearDev &l, int row0, v4i acc) {
  const v4i rs = *reinterpret_cast<const v4i *>(l.rowsum128 + row0);
  const v4f sc = *reinterpret_cast<const v4f *>(l.scale + row0);
  const v4f sb = *reinterpret_cast<const v4f *>(l.bias + row0);
  v4f o;
#pragma unroll
  for (int r = 0; r < 4; r++) o[r] = (float)(acc[r] + rs[r]) * sc[r] + sb[r];
  return o;
}

extern "C" __global__ void __launch_bounds__(NTHREADS)  // (forcing <=128 VGPRs for 4 WGs/CU spills and is slower: measured)
rn_nn_mfma_kernel(RnGroupDev g, RnModelDev m, RnTablesDev tb) {
  __shared__ __attribute__((aligned(16))) MfmaLds L;
  // The tile is one long dependency chain and the analysis kernel of the next frame queues behind the LDS it
  // holds: let its waves win instruction arbitration against the co-resident analysis waves (4096 streams:
  // +2.4 % mean over 4 alternating A/B runs, run-to-run noise +-3 %; no effect at 65,536).
  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 15, gq = lane >> 4;
  const int N = g.n_streams, s0 = blockIdx.x * TS;
  const int sn = (s0 + n < N) ? s0 + n : N - 1;              // this lane's stream (clamped for loads)
  const bool live = (s0 + n < N) && !g.silence[sn];          // silent streams keep state (src/denoise.c:474)
  const uint32_t *lut = L.lut;
  float *dbg = (g.debug && tid == 0) ? g.debug + (size_t)s0 * RN_DBG_FLOATS + RN_DBG_CLK2 : nullptr;
  unsigned long long clk_prev = g.debug ? __builtin_amdgcn_s_memtime() : 0;
#define CLK_TAP(idx)                                           \
  do {                                                         \
    if (g.debug) {                                             \
      unsigned long long now_ = __builtin_amdgcn_s_memtime();  \
      if (dbg) dbg[idx] = (float)(now_ - clk_prev);            \
      clk_prev = now_;                                         \
    }                                                          \
  } while (0)

  for (int i = tid; i < 2048; i += NTHREADS) L.lut[i] = tb.rcp_lut[i];
  for (int i = tid; i < RN_CAT; i += NTHREADS) L.vadw[i] = m.vad_dense.fw[i];
  // ---- conv1 input: [conv1_state(130) | features(65) | 0] per stream ----
  for (int e = tid; e < TS * 196; e += NTHREADS) {
    const int q = e / 196, k = e - q * 196, s = (s0 + q < N) ? s0 + q : N - 1;
    float v = 0;
    if (k < 130) v = g.conv1_state[(size_t)s * 130 + k];
    else if (k < 195) v = g.features[(size_t)s * 68 + (k - 130)];
    L.tmp1[q][k] = v;
  }
  // conv2 history: quantise old[0..255] into xq[0] (k = 0..255); keep old[128..255] to shift the state
  constexpr int HC = 1024 / NTHREADS;  // 16 streams x 64 chunks of 4 floats, spread over the workgroup
  v4f hist[HC];
  int hq_[HC], hk_[HC];
#pragma unroll
  for (int c = 0; c < HC; c++) {
    const int chunk = tid + c * NTHREADS, q = chunk >> 6, k = (chunk & 63) << 2;  // 16 streams x 64 chunks of 4
    const int s = (s0 + q < N) ? s0 + q : N - 1;
    hist[c] = *reinterpret_cast<const v4f *>(g.conv2_state + (size_t)s * 256 + k);
    hq_[c] = q;
    hk_[c] = k;
    *reinterpret_cast<int *>(L.xq[0] + frag_off(q, k)) = pack4(hist[c][0], hist[c][1], hist[c][2], hist[c][3]);
  }
  __syncthreads();
  // state shifts (src/nnet.c:122): conv1 history <- tmp1[65..194]; conv2 history[0..127] <- old[128..255]
  for (int e = tid; e < TS * 130; e += NTHREADS) {
    const int q = e / 130, k = e - q * 130;
    if (s0 + q < N && !g.silence[s0 + q]) g.conv1_state[(size_t)(s0 + q) * 130 + k] = L.tmp1[q][65 + k];
  }
#pragma unroll
  for (int c = 0; c < HC; c++)
    if (hk_[c] >= 128 && s0 + hq_[c] < N && !g.silence[s0 + hq_[c]])
      *reinterpret_cast<v4f *>(g.conv2_state + (size_t)(s0 + hq_[c]) * 256 + hk_[c] - 128) = hist[c];

  CLK_TAP(0);  // loads, history quantisation, state shifts
  // ---- conv1: f32 MFMA, 195(+1) -> 128 = 8 row tiles, spread over the waves ----
  for (int rt = wave; rt < 8; rt += NWAVES) {
    v4f acc = {0, 0, 0, 0};
    const float *fw = m.conv1.fw + 16 * rt + n;
    for (int j = 0; j < 49; j++) {
      const int k = 4 * j + gq;
      const float b = L.tmp1[n][k];
      const float a = (k < RN_CONV1_K) ? fw[k * 128] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    const int row0 = 16 * rt + 4 * gq;
    const v4f bs = *reinterpret_cast<const v4f *>(m.conv1.bias + row0);
    v4f c1;
#pragma unroll
    for (int r = 0; r < 4; r++) c1[r] = tanh_x86(acc[r] + bs[r], lut);
    *reinterpret_cast<int *>(L.xq[0] + frag_off(n, 256 + row0)) = pack4(c1[0], c1[1], c1[2], c1[3]);
    if (live) *reinterpret_cast<v4f *>(g.conv2_state + (size_t)sn * 256 + 128 + row0) = c1;
  }
  __syncthreads();

  CLK_TAP(1);  // conv1
  // ---- conv2: int8 dense 384 -> 384, tanh; wave w owns row tiles w, w+4, ... ----
  {
    for (int rt = wave; rt < 24; rt += NWAVES) {
      const int row0 = 16 * rt + 4 * gq;
      v4f o = int8_finish(m.conv2, row0, int8_tile(m.conv2.wmf, rt, lane, L.xq[0]));
#pragma unroll
      for (int r = 0; r < 4; r++) o[r] = tanh_x86(o[r], lut);
      if (s0 + n < N) *reinterpret_cast<v4f *>(g.nn_act + (size_t)sn * RN_GRU + row0) = o;  // f32 copy for dense_out
      *reinterpret_cast<int *>(L.xq[1] + frag_off(n, row0)) = pack4(o[0], o[1], o[2], o[3]);
    }
  }

  // ---- three GRUs (src/nnet.c:65-94); wave w owns hidden-unit tiles w, w+4, ... ----
  int cur = 1;
  CLK_TAP(2);  // conv2
  for (int k = 0; k < 3; k++) {
    float *st = g.gru_state + (size_t)k * g.n_stride * RN_GRU;
    for (int e = tid; e < TS * 96; e += NTHREADS) {  // quantise the old state into hq
      const int q = e / 96, c4 = (e - q * 96) << 2, s = (s0 + q < N) ? s0 + q : N - 1;
      const v4f h = *reinterpret_cast<const v4f *>(st + (size_t)s * RN_GRU + c4);
      *reinterpret_cast<int *>(L.hq + frag_off(q, c4)) = pack4(h[0], h[1], h[2], h[3]);
    }
    __syncthreads();
    const RnLinearDev &wi = m.gru_in[k], &wr = m.gru_rec[k];
    for (int u = wave; u < 24; u += NWAVES) {
      const int unit0 = 16 * u + 4 * gq;
      const v4f h_old = *reinterpret_cast<const v4f *>(st + (size_t)sn * RN_GRU + unit0);
      v4f gi[3], gr[3];
#pragma unroll
      for (int gate = 0; gate < 3; gate++) {
        gi[gate] = int8_finish(wi, gate * RN_GRU + unit0, int8_tile(wi.wmf, gate * 24 + u, lane, L.xq[cur]));
        gr[gate] = int8_finish(wr, gate * RN_GRU + unit0, int8_tile(wr.wmf, gate * 24 + u, lane, L.hq));
        const v4f dg = *reinterpret_cast<const v4f *>(wr.diag + gate * RN_GRU + unit0);
#pragma unroll
        for (int r = 0; r < 4; r++) gr[gate][r] += dg[r] * h_old[r];  // src/nnet_arch.h:153-161
      }
      v4f hn;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const float z = sigmoid_x86(gi[0][r] + gr[0][r], lut);
        const float rg = sigmoid_x86(gi[1][r] + gr[1][r], lut);
        const float hh = tanh_x86(gi[2][r] + gr[2][r] * rg, lut);
        hn[r] = z * h_old[r] + (1 - z) * hh;
      }
      if (live) *reinterpret_cast<v4f *>(st + (size_t)sn * RN_GRU + unit0) = hn;
      *reinterpret_cast<int *>(L.xq[cur ^ 1] + frag_off(n, unit0)) = pack4(hn[0], hn[1], hn[2], hn[3]);
    }
    cur ^= 1;
    __syncthreads();
    CLK_TAP(3 + k);  // GRU k
  }

  // ---- dense_out (1536 -> 32, f32 MFMA chains, waves 0-1) and vad_dense (wave 2, lane = stream) ----
  // cat = [conv2 out | gru1 | gru2 | gru3] (src/rnn.c:53-55); silent streams are computed on
  // their unchanged state and discarded.
  // The chains are serial over the 1536 inputs.  Their activation operand is staged through LDS in
  // chunks of 128 inputs by the whole workgroup (one coalesced 16-byte load per thread and chunk, issued
  // a full chunk ahead and parked in a register), because fetched straight from the per-stream rows every
  // chain step costs 16 scattered L1 accesses -- and the vector L1 is what this kernel saturates first.
  // Measured with the phase taps at 65,536 streams: what paced this phase was the lane = stream VAD chain waiting for its
  // weights in L2 at every step (77k -> 54k clocks per tile with the weights in LDS), not the MFMA chain's weights.
  // Dead ends (this round): running the chains segment by segment inside the GRU phases (the tile loop then spills at
  // 128 VGPRs: K2 1.00 -> 1.14-1.28 ms); one 384-input chunk per barrier (172 VGPRs or spills: 1.13 ms).
  {
    const int pq = tid >> 5, pc = (tid & 31) << 2;              // producer role: stream pq, floats pc + 128 j .. +3 of a chunk, j = 0, 1
    const int ps = (s0 + pq < N) ? s0 + pq : N - 1;
    auto chunk_src = [&](int c) {                                // chunk c = inputs 128c .. 128c+127 of cat
      // half-chunk h = 2c + j covers inputs 128 h .. 128 h + 127 of cat: segment h / 3, offset 128 (h % 3)
      return [=](int j) {
        const int h = 2 * c + j, seg = h / 3, k0 = (h - 3 * seg) * 128;
        return *reinterpret_cast<const v4f *>((seg == 0 ? g.nn_act : g.gru_state + (size_t)(seg - 1) * g.n_stride * RN_GRU) +
                                              (size_t)ps * RN_GRU + k0 + pc);
      };
    };
    constexpr int NCH = 4 * RN_GRU / CHUNK;  // 12
    v4f park[2] = {chunk_src(0)(0), chunk_src(0)(1)};
    *reinterpret_cast<v4f *>(&L.stage[0][pq][pc]) = park[0];  // xq / hq / tmp1 are dead: the last GRU barrier is behind us
    *reinterpret_cast<v4f *>(&L.stage[0][pq][pc + 128]) = park[1];
    park[0] = chunk_src(1)(0);
    park[1] = chunk_src(1)(1);
    v4f dacc = {0, 0, 0, 0};
    float vacc = 0;
    // dense_out weights: the MFMA-ordered copy (shim.cpp: stage_linear) -- 16 bytes = this lane's operands of four
    // consecutive steps, WDEPTH loads (4 x WDEPTH steps, > 1000 cycles of chain) in flight
    constexpr int WDEPTH = 8, NW4 = RN_CAT / 16;  // 96 groups of four steps
    static_assert((CHUNK / 16) % WDEPTH == 0, "a chunk is a whole number of weight-buffer rounds");
    const v4f *wq = reinterpret_cast<const v4f *>(m.dense_out.fwm) + (size_t)(wave & 1) * NW4 * 64 + lane;
    v4f wbuf[WDEPTH];
    if (wave < 2) {
#pragma unroll
      for (int u = 0; u < WDEPTH; u++) wbuf[u] = wq[u * 64];
    }
    __syncthreads();
    for (int c = 0; c < NCH; c++) {
      if (c + 1 < NCH) {
        *reinterpret_cast<v4f *>(&L.stage[(c + 1) & 1][pq][pc]) = park[0];
        *reinterpret_cast<v4f *>(&L.stage[(c + 1) & 1][pq][pc + 128]) = park[1];
      }
      if (c + 2 < NCH) {
        park[0] = chunk_src(c + 2)(0);
        park[1] = chunk_src(c + 2)(1);
      }
      const float(*sx)[CH_STRIDE] = L.stage[c & 1];
      if (wave < 2) {  // 64 MFMA steps: k = 256c + 16*t4 + 4e + gq
        const float *bx = &sx[n][gq];
#pragma unroll 8
        for (int t4 = 0; t4 < CHUNK / 16; t4++) {
          const int gidx = (CHUNK / 16) * c + t4;
          const v4f a = wbuf[t4 % WDEPTH];  // slot t4 % WDEPTH holds group gidx
          if (gidx + WDEPTH < NW4) wbuf[t4 % WDEPTH] = wq[(size_t)(gidx + WDEPTH) * 64];
#pragma unroll
          for (int e = 0; e < 4; e++) dacc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], bx[16 * t4 + 4 * e], dacc, 0, 0, 0);
        }
      } else if (wave == 2 && lane < TS) {  // unfused mul-then-add, src/vec_avx.h:732-736
        const v4f *w = reinterpret_cast<const v4f *>(L.vadw + CHUNK * c);
        const v4f *x = reinterpret_cast<const v4f *>(sx[lane]);
#pragma unroll 4
        for (int j = 0; j < CHUNK / 4; j++) {
          const v4f wj = w[j], xj = x[j];
#pragma unroll
          for (int e = 0; e < 4; e++) vacc = vacc + wj[e] * xj[e];
        }
      }
      __syncthreads();
    }
    if (wave < 2) {
      const int row0 = 16 * wave + 4 * gq;
      const v4f bs = *reinterpret_cast<const v4f *>(m.dense_out.bias + row0);
      v4f o;
#pragma unroll
      for (int r = 0; r < 4; r++) o[r] = live ? sigmoid_x86(dacc[r] + bs[r], lut) : 0.f;
      if (s0 + n < N) *reinterpret_cast<v4f *>(g.gains + (size_t)sn * RN_NB_BANDS + row0) = o;
    } else if (wave == 2 && lane < TS) {
      const int q = s0 + lane, sq = q < N ? q : N - 1;
      const bool lv = q < N && !g.silence[sq];
      if (q < N) g.vad[sq] = lv ? sigmoid_x86(vacc + m.vad_dense.bias[0], lut) : 0.f;
    }
  }
  CLK_TAP(6);  // dense_out / vad (wave 0's view)
}

extern "C" hipError_t rn_launch_nn_mfma(const RnGroupDev *g, const RnModelDev *m, const RnTablesDev *tb, hipStream_t st,
                                        hipEvent_t e0, hipEvent_t e1) {
  if (!m->conv2.wmf || !g->nn_act || !m->dense_out.fwm) return hipErrorNotSupported;
  RN_LAUNCH(rn_nn_mfma_kernel, dim3((g->n_streams + TS - 1) / TS), dim3(NTHREADS), 0, st, e0, e1, *g, *m, *tb);
  return hipGetLastError();
}
extern "C" int rn_nn_mfma_available(void) { return 1; }
