// nn_mfma.hip -- K2, batched MFMA path: the network (src/rnn.c:44-60) recast as
// (16 streams x K) . (K x outputs) matrix products, one workgroup (4 waves) per tile of 16
// streams.
//
//   int8 layers (conv2, 6 GRU matrices): v_mfma_i32_16x16x64_i8 on the block-sparse weights
//     zero-filled to dense and pre-swizzled into A-fragment order (model.cpp: stage_linear), the
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
#include "nn_common.h"
#include <stdlib.h>
#include <atomic>

#define CHUNK 256      // inputs per staged chunk of the dense_out / vad chains
#define CH_STRIDE 260  // floats per stream and chunk in LDS (16-byte aligned rows, 2-way bank conflicts at most)

struct MfmaLds {
  uint16_t lut[4096];             // rcpps table (rn_dev.h: rcp16)
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

// the front kernel (RN_NN_MODE 1: conv1, conv2) touches neither the recurrent image, nor the dense-phase staging, nor the
// vad weights: 32.6 KB instead of 47.6 -- four workgroups per CU instead of three (it is bound by the latency of a tile)
struct FrontLds {
  uint16_t lut[4096];
  float tmp1[TS][197];
  int8_t xq[2][KT * 64 * 16];
};

// ... and with the output chains' staging (RN_NN_MODE 2): the f32 conv2 outputs of the tile, 24 KB, over the conv1 input (dead behind
// conv1's barrier): 44 KB -- three workgroups per CU, which is what the kernel's registers allow anyway
struct FrontFoldLds {
  uint16_t lut[4096];
  union {
    float tmp1[TS][197];
    float stage[RN_GRU / 4][TS][4];
  };
  int8_t xq[2][KT * 64 * 16];
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

// MODE 0: the whole network for one tile.  Large batches run it layer by layer instead (nn_layers.hip): MODE 1 = the
// front (conv1, conv2; leaves the quantised conv2 output as a B-fragment image in act_q[0]), then three launches of the
// 64-stream GRU layer kernel, then the 64-stream dense kernel on the f32 activations the others left in HBM.
// Two workgroups per CU (4 waves per SIMD, <= 128 VGPRs) is what the 47 KB of LDS is sized for; left to itself the
// allocator drifts between 121 and 162 VGPRs with unrelated edits, and above 128 only one workgroup fits.
// "At least 4" costs 8 spilled registers, "exactly 4" 36.  (Forcing <= 64 VGPRs for 4 workgroups per CU is slower: measured.)
extern "C" __global__ void __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(4)))
rn_nn_mfma_kernel(RnGroupDev g, RnModelDev m, RnTablesDev tb) {
#define RN_NN_MODE 0
#include "nn_tile_body.inc"
#undef RN_NN_MODE
}
// The same with SIXTEEN waves per tile, for one-frame calls on batches in which every tile has a CU to itself (n_tiles <= CUs: up to
// 4,096 streams on MI355X).  The tile is a chain of barrier-separated phases, each as long as its slowest wave: the GRU phases give a
// wave 24 / NWAVES unit tiles (36 MFMAs each, every weight fragment an L2 round trip), conv2 as many row tiles -- 3 with eight waves,
// 2 with sixteen (waves 8..15 one).  1,024 threads = 4 waves per SIMD at the same <= 128 VGPRs: the workgroup takes every register of
// its CU, so inside a pipelined multi-frame call -- where at these sizes the analysis and synthesis waves of the neighbouring frames
// live on the CU's other half -- it is the slower form (4,096 streams: 21.5 against 25.3 M frames/s), and alone the faster one
// (0.0750 -> 0.0686 ms, one frame per call 0.237 -> 0.2305 ms per step; profiles/r6_late_ab.txt).  Same arithmetic per element, same
// bits (tests/test_gpu_parity.py runs both forms at every small size).
#undef NWAVES
#define NWAVES 16
extern "C" __global__ void __launch_bounds__(NTHREADS) rn_nn_mfma16_kernel(RnGroupDev g, RnModelDev m, RnTablesDev tb) {
#define RN_NN_MODE 0
#include "nn_tile_body.inc"
#undef RN_NN_MODE
}
#undef NWAVES
#define NWAVES 8
extern "C" __global__ void __launch_bounds__(NTHREADS) rn_nn_front_kernel(RnGroupDev g, RnModelDev m, RnTablesDev tb) {
#define RN_NN_MODE 1
#include "nn_tile_body.inc"
#undef RN_NN_MODE
}
#if RN_INSTRUMENT
// LAB (profiles/r6_dense_fold.txt): the front with the output chains taken through the conv2 segment (nn_gru.h: gru_body FOLD)
extern "C" __global__ void __launch_bounds__(NTHREADS) rn_nn_front_fold_kernel(RnGroupDev g, RnModelDev m, RnTablesDev tb) {
#define RN_NN_MODE 2
#include "nn_tile_body.inc"
#undef RN_NN_MODE
}
// the same at <= 64 VGPRs (A/B of the instrumented build, $RNNOISE_AMD_FRONT64=1): the front's 33 KB of LDS allow four workgroups per
// CU, its 66 -> 72 allocated registers x 2 waves per SIMD only three
extern "C" __global__ void __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(8, 8)))
rn_nn_front64_kernel(RnGroupDev g, RnModelDev m, RnTablesDev tb) {
#define RN_NN_MODE 1
#include "nn_tile_body.inc"
#undef RN_NN_MODE
}
#endif
extern "C" hipError_t rn_launch_nn_mfma(const RnGroupDev *g, const RnModelDev *m, const RnTablesDev *tb, hipStream_t st,
                                        hipEvent_t e0, hipEvent_t e1, int alone) {
  if (!m->conv2.wmf || !g->nn_act || !m->dense_out.fwm || !m->conv1.fwm) return hipErrorNotSupported;
  const int n_tiles = (g->n_streams + TS - 1) / TS;
  // sixteen waves per tile in a call that runs nothing beside the network while every tile has a CU to itself, eight otherwise;
  // $RNNOISE_AMD_TILE_WAVES = 8 | 16 forces one (same bits: tests run both)
  static const int forced = [] { const char *e = getenv("RNNOISE_AMD_TILE_WAVES"); return e ? atoi(e) : 0; }();
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  static std::atomic<int> cus[64];  // (per-device cache, relaxed: every writer stores the same value)
  int ncu = cus[dev].load(std::memory_order_relaxed);
  if (!ncu) {
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
    cus[dev].store(ncu, std::memory_order_relaxed);
  }
  if (forced == 16 || (forced != 8 && alone && n_tiles <= ncu))
    RN_LAUNCH(rn_nn_mfma16_kernel, dim3(n_tiles), dim3(1024), 0, st, e0, e1, *g, *m, *tb);
  else
    RN_LAUNCH(rn_nn_mfma_kernel, dim3(n_tiles), dim3(NTHREADS), 0, st, e0, e1, *g, *m, *tb);
  return hipGetLastError();
}
extern "C" hipError_t rn_launch_nn_gru_layer(const RnGroupDev *g, const RnModelDev *m, const RnTablesDev *tb, int layer, hipStream_t st,
                                             hipEvent_t e0, hipEvent_t e1);
extern "C" hipError_t rn_launch_nn_dense(const RnGroupDev *g, const RnModelDev *m, const RnTablesDev *tb, hipStream_t st, hipEvent_t e0,
                                         hipEvent_t e1);
extern "C" int rn_nn_layers_fold(void);  // nn_layers.hip
// launches of the layer-wise network: front, three GRU layers, dense -- or, with the output chains folded into the layer launches, four
extern "C" int rn_nn_layers_launches(void) { return rn_nn_layers_fold() ? 4 : 5; }
// the network layer by layer (front, three GRU layers at 64 streams per workgroup, [dense]); ev[i] = the optional (start, stop) events
// of launch i: each kernel is timed on its own, the durations add up to the network's
extern "C" hipError_t rn_launch_nn_layers(const RnGroupDev *g, const RnModelDev *m, const RnTablesDev *tb, hipStream_t st,
                                          hipEvent_t ev[5][2]) {
  if (!m->conv2.wmf || !g->nn_act || !m->dense_out.fwm || !m->conv1.fwm || !g->act_q[0] || g->n_streams != g->n_stride) return hipErrorNotSupported;
  if ((size_t)g->n_streams * RN_GRU * 4 >= (1ull << 32)) return hipErrorNotSupported;  // 32-bit offsets in nn_layers.hip
  const dim3 grid((g->n_streams + TS - 1) / TS);
  const bool fold = rn_nn_layers_fold();
#if RN_INSTRUMENT
  static const bool front64 = [] { const char *e = RN_LAB_ENV("FRONT64"); return e && atoi(e) == 1; }();
  if (fold) RN_LAUNCH(rn_nn_front_fold_kernel, grid, dim3(NTHREADS), 0, st, ev[0][0], ev[0][1], *g, *m, *tb);
  else if (front64) RN_LAUNCH(rn_nn_front64_kernel, grid, dim3(NTHREADS), 0, st, ev[0][0], ev[0][1], *g, *m, *tb);
  else
#endif
    RN_LAUNCH(rn_nn_front_kernel, grid, dim3(NTHREADS), 0, st, ev[0][0], ev[0][1], *g, *m, *tb);
  for (int k = 0; k < 3; k++) {
    hipError_t e = rn_launch_nn_gru_layer(g, m, tb, k, st, ev[1 + k][0], ev[1 + k][1]);
    if (e != hipSuccess) return e;
  }
  return fold ? hipGetLastError() : rn_launch_nn_dense(g, m, tb, st, ev[4][0], ev[4][1]);
}
extern "C" int rn_nn_mfma_available(void) { return 1; }
