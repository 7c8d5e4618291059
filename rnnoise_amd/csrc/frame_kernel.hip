// frame_kernel.hip -- the reference's own call, rnnoise_process_frame (src/denoise.c:457-504), as ONE launch: the four stages of
// a frame step for the rows of a launch group of the one-frame API (rn_dev.h: RnRows; dropin.cpp: the combiner), one 14-wave
// workgroup per row.
//
//   stage          waves      what                                              LDS
//   K0             0          hp_one_body        (hp_kernel.hip)                stage arena (behind the network's)
//   (beside it)    1..13      nn_one_prefetch    conv1 matrix, rcpps table, vad weights -> the network's LDS, by LDS-DMA
//   K1             0          analysis_body<1>   (dsp_kernels.hip)              stage arena
//   K2             0..13      nn_one_body        (nn_kernels.hip)               OneLds
//   K3             0          synthesis_body     (dsp_kernels.hip)              stage arena
//
// The stages are the bodies of the stand-alone latency kernels, included as they are (RN_FUSED_BUILD hides those files' own
// kernels and launchers): same arithmetic, same bits.  What the fusion buys is the three dependent-dispatch gaps between four
// tiny kernels (4-5 us each on the GPU side) and the network's prologue, which now runs under the high-pass filter; between
// stages the hand-off is a workgroup barrier, global memory written by one stage and read by the next inside one workgroup
// (one CU, one L1).  The workgroup is compiled for 14 waves (128 VGPRs): the analysis and synthesis bodies fit that.
#define RN_FUSED_BUILD 1
#include "hp_kernel.hip"
#include "dsp_kernels.hip"
#include "nn_kernels.hip"

#define FRAME_STAGE_OFF ((int)((sizeof(OneLds) + 255) & ~(size_t)255))
#define FRAME_STAGE_BYTES \
  (sizeof(HpOneLds) > sizeof(AnalysisLds) ? (sizeof(HpOneLds) > sizeof(SynthLds) ? sizeof(HpOneLds) : sizeof(SynthLds)) \
                                          : (sizeof(AnalysisLds) > sizeof(SynthLds) ? sizeof(AnalysisLds) : sizeof(SynthLds)))
#define FRAME_LDS_BYTES (FRAME_STAGE_OFF + ((FRAME_STAGE_BYTES + 255) & ~(size_t)255))
static_assert(FRAME_LDS_BYTES <= 160 * 1024, "one frame workgroup per CU");

extern "C" __global__ void __launch_bounds__(ONE_THREADS)
rn_frame_rows_kernel(RnGroupDev g, RnModelDev m, RnTablesDev tb, RnRows rows) {
  extern __shared__ __attribute__((aligned(16))) char frame_smem[];
  OneLds &O = *reinterpret_cast<OneLds *>(frame_smem);
  const uint32_t re = rows.e[blockIdx.x];
  const int s = (int)(re & 255u), slot = (int)((re >> 8) & 7u), parity = (int)((re >> 12) & 3u);
  const int prev = (parity + RN_SPEC_SLOTS - 1) % RN_SPEC_SLOTS;
  float *row = rows.io + (size_t)s * RN_ROW_IO;  // in[480] | pad[4] | out[480] | vad | pad[2] | done
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (wave == 0) hp_one_body(g, row, s, slot, 0, *reinterpret_cast<HpOneLds *>(frame_smem + FRAME_STAGE_OFF), lane);
  else nn_one_prefetch(m, tb, O, wave - 1, ONE_THREADS / 64 - 1, lane);
  __syncthreads();
  if (wave == 0) analysis_body<false, 1>(g, tb, slot, parity, RnTrainArgs{}, s, FRAME_STAGE_OFF);
  __syncthreads();
  nn_one_body(g, m, tb, s, row + 2 * RN_FRAME_SIZE + 4, O, true);
  __syncthreads();
  if (wave == 0)
    synthesis_body(g, tb, row + RN_FRAME_SIZE + 4, false, s, parity, prev, reinterpret_cast<uint32_t *>(row + RN_ROW_IO - 1), re >> 16,
                   FRAME_STAGE_OFF);
}

// one launch per group of the one-frame API; the LDS opt-in is per device (a process may hold pools on several GPUs)
extern "C" hipError_t rn_launch_frame_rows(const RnGroupDev *g, const RnModelDev *m, const RnTablesDev *tb, const RnRows *rows, hipStream_t st) {
  static std::atomic<int> opted[64];
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64 || !opted[dev].load(std::memory_order_acquire)) {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(rn_frame_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FRAME_LDS_BYTES);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) opted[dev].store(1, std::memory_order_release);
  }
  hipLaunchKernelGGL(rn_frame_rows_kernel, dim3(rows->n), dim3(ONE_THREADS), FRAME_LDS_BYTES, st, *g, *m, *tb, *rows);
  return hipGetLastError();
}
