// nn_layers.hip -- K2 for large batches: one GRU layer (src/nnet.c:65-94) for 64 streams per workgroup.
//
// The fused tile kernel (nn_mfma.hip) walks all seven int8 matrices for 16 streams: every weight fragment a wave
// fetches from L2 feeds ONE MFMA, and its phases are latency-bound (rocprof: 61-66 % of wave time waiting).  Here
// a workgroup holds the quantised inputs of GM = 4 tiles in LDS and every weight fragment feeds four MFMAs, one per
// tile: a quarter of the L2 weight traffic per stream, four independent accumulator chains per wave.  The layer's input
// arrives as the B-fragment image the previous launch left in HBM (act_q), its output leaves the same way.
// Arithmetic per element is the fused kernel's, so the bits are too (tests/test_gpu_parity.py runs both).
#include "nn_common.h"

#define GM 4  // 16-stream tiles per workgroup
#ifndef GW
#define GW 8  // waves per workgroup
#endif
#define GTHREADS (64 * GW)

struct GruLds {
  uint32_t lut[2048];            // rcpps table
  int8_t xq[GM][KT * 64 * 16];   // layer input images
  int8_t hq[GM][KT * 64 * 16];   // quantised recurrent state
};

// acc[gate][t] += W(row tile 24 gate + u) . image[t]: the three gates of a unit tile share the layer input, so one B
// fragment read from LDS feeds three MFMAs and one A fragment from L2 four.  (Measured with the 1 x 4 blocking of the
// first version: a 16x16x64 MFMA takes 16 cycles on its SIMD, its 1 KB B fragment 8 cycles of the CU's one LDS port --
// four SIMDs re-reading B per MFMA are LDS-bound at half the MFMA rate.)
// Addressing is (uniform base, unsigned 32-bit offset) throughout this file: the SGPR-base + VGPR-offset form of global_load.
// The A fragments come from L2 (~600 cycles): a rolling buffer keeps them AD k-steps ahead of their MFMAs, across the
// boundary between the input and the recurrent matrix (step = 0..5 input, 6..11 recurrent).
#define AD 2
struct AFrags {
  v4i f[AD + 1][3];
};
__device__ __forceinline__ void a_fetch(AFrags &A, int step, const int8_t *__restrict__ wi, const int8_t *__restrict__ wr, unsigned a0) {
  const v4i *a = reinterpret_cast<const v4i *>(step < KT ? wi : wr);
  const int kt = step < KT ? step : step - KT;
#pragma unroll
  for (int gate = 0; gate < 3; gate++) A.f[step % (AD + 1)][gate] = a[a0 + (unsigned)((gate * 24 * KT + kt) * 64)];
}
// k-steps [s0, s0 + KT) of the rolling sequence: acc[gate][t] += A(step)[gate] . image[t]
__device__ __forceinline__ void int8_gates(v4i acc[3][GM], AFrags &A, int s0, const int8_t *__restrict__ wi, const int8_t *__restrict__ wr,
                                           unsigned a0, int lane, const int8_t (*bq)[KT * 64 * 16]) {
  asm volatile("" : "+v"(lane));  // (the images do not change inside the kernel: keep the compiler from hoisting all 48 B fragments)
#pragma unroll
  for (int kt = 0; kt < KT; kt++) {
    const int step = s0 + kt;
    if (step + AD < 2 * KT) a_fetch(A, step + AD, wi, wr, a0);
    __builtin_amdgcn_sched_barrier(0);  // (else the scheduler sinks the fetch to its use to save registers: every A fragment an exposed L2 trip)
    v4i bf[GM];
#pragma unroll
    for (int t = 0; t < GM; t++) bf[t] = reinterpret_cast<const v4i *>(bq[t])[kt * 64 + lane];
#pragma unroll
    for (int gate = 0; gate < 3; gate++)
#pragma unroll
      for (int t = 0; t < GM; t++)
        acc[gate][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A.f[step % (AD + 1)][gate], bf[t], acc[gate][t], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

extern "C" __global__ void __launch_bounds__(GTHREADS) rn_nn_gru_kernel(RnGroupDev g, RnModelDev m, RnTablesDev tb, int layer) {
  __shared__ __attribute__((aligned(16))) GruLds L;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 15, gq = lane >> 4;
  const int N = g.n_streams, n_tiles = (N + TS - 1) / TS, tile0 = blockIdx.x * GM;
  const uint32_t *lut = L.lut;
  float *st = g.gru_state + (size_t)layer * g.n_stride * RN_GRU;
  const int8_t *xin = g.act_q[layer & 1];
  int8_t *xout = g.act_q[(layer + 1) & 1];

  // (tests / profiling: shader-clock taps of wave 0, slots RN_DBG_CLK2 + 7 + 3 * layer + {0: prologue, 1: loads issued -> barrier, 2: tiles})
  float *dbg = (g.debug && tid == 0) ? g.debug + (size_t)tile0 * TS * RN_DBG_FLOATS + RN_DBG_CLK2 + 7 + 3 * layer : nullptr;
  const unsigned long long clk0 = g.debug ? __builtin_amdgcn_s_memtime() : 0;
  int sn[GM], sil[GM];
  bool live[GM];
#pragma unroll
  for (int t = 0; t < GM; t++) {
    const int s = (tile0 + t) * TS + n;
    sn[t] = s < N ? s : N - 1;
    sil[t] = g.silence[(unsigned)sn[t]];
  }
  // all loads of the prologue in flight at once (constant trip counts, fully unrolled): one HBM round trip, not twelve
  {
    constexpr int NX = GM * KT * 64 / GTHREADS, NH = GM * TS * 96 / GTHREADS;  // 3 and 12 16-byte loads per thread
    static_assert(NX * GTHREADS == GM * KT * 64 && NH * GTHREADS == GM * TS * 96, "prologue tiling");
    v4i xi[NX];
    v4f hs[NH];
#pragma unroll
    for (int j = 0; j < NX; j++) {
      const int i = tid + j * GTHREADS, t = i / (KT * 64), o = i - t * (KT * 64), tile = (tile0 + t < n_tiles) ? tile0 + t : n_tiles - 1;
      xi[j] = reinterpret_cast<const v4i *>(xin)[(unsigned)(tile * (KT * 64) + o)];
    }
#pragma unroll
    for (int j = 0; j < NH; j++) {
      const int e = tid + j * GTHREADS, q = e / 96, c = e - q * 96, s = (tile0 * TS + q < N) ? tile0 * TS + q : N - 1;
      hs[j] = reinterpret_cast<const v4f *>(st)[(unsigned)(s * 96 + c)];
    }
#pragma unroll
    for (int j = 0; j < (2048 + GTHREADS - 1) / GTHREADS; j++)
      if (tid + j * GTHREADS < 2048) L.lut[tid + j * GTHREADS] = tb.rcp_lut[tid + j * GTHREADS];
#pragma unroll
    for (int j = 0; j < NX; j++) {
      const int i = tid + j * GTHREADS, t = i / (KT * 64), o = i - t * (KT * 64);
      reinterpret_cast<v4i *>(L.xq[t])[o] = xi[j];
    }
#pragma unroll
    for (int j = 0; j < NH; j++) {  // quantise the old state
      const int e = tid + j * GTHREADS, q = e / 96, c4 = (e - q * 96) << 2;
      *reinterpret_cast<int *>(L.hq[q >> 4] + frag_off(q & 15, c4)) = pack4(hs[j][0], hs[j][1], hs[j][2], hs[j][3]);
    }
  }
  const unsigned long long clk1 = g.debug ? __builtin_amdgcn_s_memtime() : 0;
  __syncthreads();
  const unsigned long long clk2 = g.debug ? __builtin_amdgcn_s_memtime() : 0;

#pragma unroll
  for (int t = 0; t < GM; t++) live[t] = (tile0 + t) * TS + n < N && !sil[t];  // silent streams keep their state (src/denoise.c:474)
  const RnLinearDev &wi = m.gru_in[layer], &wr = m.gru_rec[layer];
#pragma unroll 1
  for (int u = wave; u < 24; u += GW) {
    const int unit0 = 16 * u + 4 * gq;
    v4i acc[3][GM];
    v4f gi[3][GM], h_old[GM];
#pragma unroll
    for (int t = 0; t < GM; t++) h_old[t] = reinterpret_cast<const v4f *>(st)[(unsigned)(sn[t] * (RN_GRU / 4) + (unit0 >> 2))];
#pragma unroll
    for (int gate = 0; gate < 3; gate++)
#pragma unroll
      for (int t = 0; t < GM; t++) acc[gate][t] = v4i{0, 0, 0, 0};
    const unsigned a0 = (unsigned)(u * KT * 64 + lane);
    AFrags A;
#pragma unroll
    for (int step = 0; step < AD; step++) a_fetch(A, step, wi.wmf, wr.wmf, a0);
    int8_gates(acc, A, 0, wi.wmf, wr.wmf, a0, lane, L.xq);
#pragma unroll
    for (int gate = 0; gate < 3; gate++) {  // float(acc_x86)*scale + subias (src/nnet_arch.h:145-151)
      const unsigned row4 = (unsigned)(gate * RN_GRU + unit0) >> 2;
      const v4i rs = reinterpret_cast<const v4i *>(wi.rowsum128)[row4];
      const v4f sc = reinterpret_cast<const v4f *>(wi.scale)[row4];
      const v4f sb = reinterpret_cast<const v4f *>(wi.bias)[row4];
#pragma unroll
      for (int t = 0; t < GM; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          gi[gate][t][r] = (float)(acc[gate][t][r] + rs[r]) * sc[r] + sb[r];
          acc[gate][t][r] = 0;
        }
    }
    int8_gates(acc, A, KT, wi.wmf, wr.wmf, a0, lane, L.hq);
    v4f gr[3][GM];
#pragma unroll
    for (int gate = 0; gate < 3; gate++) {
      const unsigned row4 = (unsigned)(gate * RN_GRU + unit0) >> 2;
      const v4i rs = reinterpret_cast<const v4i *>(wr.rowsum128)[row4];
      const v4f sc = reinterpret_cast<const v4f *>(wr.scale)[row4];
      const v4f sb = reinterpret_cast<const v4f *>(wr.bias)[row4];
      const v4f dg = reinterpret_cast<const v4f *>(wr.diag)[row4];
#pragma unroll
      for (int t = 0; t < GM; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          gr[gate][t][r] = (float)(acc[gate][t][r] + rs[r]) * sc[r] + sb[r];
          gr[gate][t][r] += dg[r] * h_old[t][r];  // src/nnet_arch.h:153-161
        }
    }
#pragma unroll
    for (int t = 0; t < GM; t++) {
      v4f hn;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const float z = sigmoid_x86(gi[0][t][r] + gr[0][t][r], lut);
        const float rg = sigmoid_x86(gi[1][t][r] + gr[1][t][r], lut);
        const float hh = tanh_x86(gi[2][t][r] + gr[2][t][r] * rg, lut);
        hn[r] = z * h_old[t][r] + (1 - z) * hh;
      }
      if (live[t]) reinterpret_cast<v4f *>(st)[(unsigned)(sn[t] * (RN_GRU / 4) + (unit0 >> 2))] = hn;
      if (layer < 2 && tile0 + t < n_tiles)
        reinterpret_cast<int *>(xout)[(unsigned)((tile0 + t) * (KT * 64 * 16) + frag_off(n, unit0)) >> 2] = pack4(hn[0], hn[1], hn[2], hn[3]);
    }
  }
  if (dbg && tile0 * TS < N) {
    const unsigned long long clk3 = __builtin_amdgcn_s_memtime();
    dbg[0] = (float)(clk1 - clk0);
    dbg[1] = (float)(clk2 - clk1);
    dbg[2] = (float)(clk3 - clk2);
  }
}

extern "C" hipError_t rn_launch_nn_gru_layer(const RnGroupDev *g, const RnModelDev *m, const RnTablesDev *tb, int layer, hipStream_t st) {
  const int n_tiles = (g->n_streams + TS - 1) / TS;
  hipLaunchKernelGGL(rn_nn_gru_kernel, dim3((n_tiles + GM - 1) / GM), dim3(GTHREADS), 0, st, *g, *m, *tb, layer);
  return hipGetLastError();
}
