// state_kernels.hip -- portable per-stream state (include/rn_layout.h: the 25,128 live bytes of the reference's
// DenoiseState, src/denoise.c:68-88) <-> the batch's structure-of-arrays layout (rn_dev.h), on the device.
// One launch moves `g.n_streams` states (a 1024-thread workgroup each: the per-word field dispatch is latency, 17 us with 256
// threads); the host side needs one memcpy per direction instead of one per field.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rn_dev.h"

// flat[s][RN_STATE_FLOATS] <- stream s of the view.  newest_slot = pitch-ring slot of the latest frame,
// last = spectra slot of the latest frame (the reference's delayed_*).
extern "C" __global__ void __launch_bounds__(1024)
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
extern "C" __global__ void __launch_bounds__(1024)
rn_state_scatter_kernel(RnGroupDev g, const float *__restrict__ flat, int newest_slot, int last) {
  const size_t s = blockIdx.x, N = g.n_stride;
  const float *f = flat + s * RN_STATE_FLOATS;
  const int ring0 = RN_RING0(newest_slot);
  float *ring = g.pitch_ring + s * RN_RING_SIZE;
  for (int p = threadIdx.x; p < RN_RING_SIZE; p += blockDim.x) {  // the 1152 ring positions outside pitch_buf are zeroed
    const int i = (p - ring0 + RN_RING_SIZE) % RN_RING_SIZE;
    ring[p] = i < RN_PITCH_BUF_SIZE ? f[RN_OFF_PITCH_BUF + i] : 0.f;
  }
  // the decimated ring is derived data (rn_dev.h: RN_XRING_SLOT): sample q = the high-pass kernel's expression over ring positions
  // 2q-1, 2q, 2q+1 as just written (zeros outside pitch_buf)
  float *xring = g.xlp_ring + s * RN_XRING_SIZE;
  auto at = [&](int p) {
    const int i = (p - ring0 + 2 * RN_RING_SIZE) % RN_RING_SIZE;
    return i < RN_PITCH_BUF_SIZE ? f[RN_OFF_PITCH_BUF + i] : 0.f;
  };
  for (int q = threadIdx.x; q < RN_XRING_SIZE; q += blockDim.x) xring[q] = .5f * (.5f * (at(2 * q - 1) + at(2 * q + 1)) + at(2 * q));
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
  hipLaunchKernelGGL(rn_state_gather_kernel, dim3(g->n_streams), dim3(1024), 0, st, *g, flat, newest_slot, last);
  return hipGetLastError();
}
extern "C" hipError_t rn_launch_state_scatter(const RnGroupDev *g, const float *flat, int newest_slot, int last, hipStream_t st) {
  hipLaunchKernelGGL(rn_state_scatter_kernel, dim3(g->n_streams), dim3(1024), 0, st, *g, flat, newest_slot, last);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Device -> pinned-host copy by a SMALL kernel (host-fed path, host_io.cpp: batch_process_pinned).  hipMemcpyAsync finds the
// DMA engine busy with the upload running the other way and falls back to the runtime's blit kernel -- 256 workgroups of
// 512 lanes whose PCIe-bound stores fill the memory pipeline of every CU: the analysis kernel running beside it takes
// 2.3 ms instead of 1.1 (rocprofv3 trace).  Posted writes over PCIe need few lanes to fill the link, so this copy runs on
// `blocks` workgroups (default 32) at the lowest wave priority.
// ---------------------------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(256)
rn_copy_to_host_kernel(uint4 *__restrict__ dst, const uint4 *__restrict__ src, size_t n16, uint32_t *__restrict__ dst_tail,
                       const uint32_t *__restrict__ src_tail, int n_tail) {
  __builtin_amdgcn_s_setprio(0);
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {  // four 16-byte loads in flight per lane
    const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
  }
  for (; i < n16; i += stride) dst[i] = src[i];
  if (blockIdx.x == 0 && (int)threadIdx.x < n_tail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}

// bytes: a multiple of 4; dst / src 16-byte aligned
extern "C" hipError_t rn_launch_copy_to_host(void *dst, const void *src, size_t bytes, int blocks, hipStream_t st) {
  const size_t n16 = bytes / 16;
  const int n_tail = (int)((bytes % 16) / 4);
  hipLaunchKernelGGL(rn_copy_to_host_kernel, dim3(blocks), dim3(256), 0, st, static_cast<uint4 *>(dst), static_cast<const uint4 *>(src), n16,
                     reinterpret_cast<uint32_t *>(static_cast<char *>(dst) + 16 * n16),
                     reinterpret_cast<const uint32_t *>(static_cast<const char *>(src) + 16 * n16), n_tail);
  return hipGetLastError();
}

// One 64-bit store with system-scope release: how a HIP stream releases an explicit SDMA-engine copy that lists an HSA signal as its
// dependency (host_io.cpp, copy mode "sdma": p = the value word of the signal; everything the stream's earlier kernels wrote is
// visible to the copy engine that sees the 0).
extern "C" __global__ void rn_release_store_kernel(long long *p, long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
extern "C" hipError_t rn_launch_release_store(void *p, long long v, hipStream_t st) {
  hipLaunchKernelGGL(rn_release_store_kernel, dim3(1), dim3(1), 0, st, static_cast<long long *>(p), v);
  return hipGetLastError();
}
