// state_kernels.hip -- portable per-stream state (include/rn_layout.h: the 25,128 live bytes of the reference's
// DenoiseState, src/denoise.c:68-88) <-> the batch's structure-of-arrays layout (rn_dev.h), on the device.
// One launch moves `g.n_streams` states; the host side needs one memcpy per direction instead of one per field.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rn_dev.h"

// flat[s][RN_STATE_FLOATS] <- stream s of the view.  newest_slot = pitch-ring slot of the latest frame,
// last = spectra slot of the latest frame (the reference's delayed_*).
extern "C" __global__ void __launch_bounds__(256)
rn_state_gather_kernel(RnGroupDev g, float *__restrict__ flat, int newest_slot, int last) {
  const size_t s = blockIdx.x, N = g.n_stride;
  float *f = flat + s * RN_STATE_FLOATS;
  const int ring0 = RN_RING0(newest_slot);
  const float *ring = g.pitch_ring + s * RN_RING_SIZE;
  for (int w = threadIdx.x; w < RN_STATE_FLOATS; w += blockDim.x) {
    float v;
    if (w < RN_OFF_SYNTHESIS) v = ring[(ring0 + (RN_PITCH_BUF_SIZE - RN_FRAME_SIZE) + w) % RN_RING_SIZE];  // analysis_mem = tail of pitch_buf
    else if (w < RN_OFF_PITCH_BUF) v = g.synth_mem[s * RN_FRAME_SIZE + (w - RN_OFF_SYNTHESIS)];
    else if (w < RN_OFF_LAST_GAIN) v = ring[(ring0 + (w - RN_OFF_PITCH_BUF)) % RN_RING_SIZE];
    else if (w == RN_OFF_LAST_GAIN) v = g.last_gain[s];
    else if (w == RN_OFF_LAST_PERIOD) v = __int_as_float(g.last_period[s]);
    else if (w < RN_OFF_LASTG) v = g.mem_hp[2 * s + (w - RN_OFF_MEM_HP)];
    else if (w < RN_OFF_CONV1) v = g.lastg[s * RN_NB_BANDS + (w - RN_OFF_LASTG)];
    else if (w < RN_OFF_CONV2) v = g.conv1_state[s * 130 + (w - RN_OFF_CONV1)];
    else if (w < RN_OFF_GRU1) v = g.conv2_state[s * 256 + (w - RN_OFF_CONV2)];
    else if (w < RN_OFF_DELAYED_X) {
      const int k = (w - RN_OFF_GRU1) / RN_GRU, i = (w - RN_OFF_GRU1) % RN_GRU;
      v = g.gru_state[(k * N + s) * RN_GRU + i];
    } else if (w < RN_OFF_DELAYED_P) v = g.spec_X[last][s * RN_SPEC_STRIDE + (w - RN_OFF_DELAYED_X)];
    else if (w < RN_OFF_DELAYED_EX) v = g.spec_P[last][s * RN_SPEC_STRIDE + (w - RN_OFF_DELAYED_P)];
    else v = g.spec_E[last][s * 96 + (w - RN_OFF_DELAYED_EX)];
    f[w] = v;
  }
}

// stream s of the view <- flat[s][RN_STATE_FLOATS] (analysis_mem is implied by pitch_buf and not stored)
extern "C" __global__ void __launch_bounds__(256)
rn_state_scatter_kernel(RnGroupDev g, const float *__restrict__ flat, int newest_slot, int last) {
  const size_t s = blockIdx.x, N = g.n_stride;
  const float *f = flat + s * RN_STATE_FLOATS;
  const int ring0 = RN_RING0(newest_slot);
  float *ring = g.pitch_ring + s * RN_RING_SIZE;
  for (int p = threadIdx.x; p < RN_RING_SIZE; p += blockDim.x) {  // the 1152 ring positions outside pitch_buf are zeroed
    const int i = (p - ring0 + RN_RING_SIZE) % RN_RING_SIZE;
    ring[p] = i < RN_PITCH_BUF_SIZE ? f[RN_OFF_PITCH_BUF + i] : 0.f;
  }
  for (int w = RN_OFF_SYNTHESIS + threadIdx.x; w < RN_STATE_FLOATS; w += blockDim.x) {
    const float v = f[w];
    if (w < RN_OFF_PITCH_BUF) g.synth_mem[s * RN_FRAME_SIZE + (w - RN_OFF_SYNTHESIS)] = v;
    else if (w < RN_OFF_LAST_GAIN) continue;
    else if (w == RN_OFF_LAST_GAIN) g.last_gain[s] = v;
    else if (w == RN_OFF_LAST_PERIOD) g.last_period[s] = __float_as_int(v);
    else if (w < RN_OFF_LASTG) g.mem_hp[2 * s + (w - RN_OFF_MEM_HP)] = v;
    else if (w < RN_OFF_CONV1) g.lastg[s * RN_NB_BANDS + (w - RN_OFF_LASTG)] = v;
    else if (w < RN_OFF_CONV2) g.conv1_state[s * 130 + (w - RN_OFF_CONV1)] = v;
    else if (w < RN_OFF_GRU1) g.conv2_state[s * 256 + (w - RN_OFF_CONV2)] = v;
    else if (w < RN_OFF_DELAYED_X) {
      const int k = (w - RN_OFF_GRU1) / RN_GRU, i = (w - RN_OFF_GRU1) % RN_GRU;
      g.gru_state[(k * N + s) * RN_GRU + i] = v;
    } else if (w < RN_OFF_DELAYED_P) g.spec_X[last][s * RN_SPEC_STRIDE + (w - RN_OFF_DELAYED_X)] = v;
    else if (w < RN_OFF_DELAYED_EX) g.spec_P[last][s * RN_SPEC_STRIDE + (w - RN_OFF_DELAYED_P)] = v;
    else g.spec_E[last][s * 96 + (w - RN_OFF_DELAYED_EX)] = v;
  }
}

extern "C" hipError_t rn_launch_state_gather(const RnGroupDev *g, float *flat, int newest_slot, int last, hipStream_t st) {
  hipLaunchKernelGGL(rn_state_gather_kernel, dim3(g->n_streams), dim3(256), 0, st, *g, flat, newest_slot, last);
  return hipGetLastError();
}
extern "C" hipError_t rn_launch_state_scatter(const RnGroupDev *g, const float *flat, int newest_slot, int last, hipStream_t st) {
  hipLaunchKernelGGL(rn_state_scatter_kernel, dim3(g->n_streams), dim3(256), 0, st, *g, flat, newest_slot, last);
  return hipGetLastError();
}
