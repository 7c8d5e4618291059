// fft_probe.hip -- test / measurement kernels for the register-resident FFT (fft_reg.h).  Not on the product path:
// reached only through rnnoise_amd_debug_fft (tests, tools/fft_bench.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fft_reg.h"
#include "rn_dev.h"

// Each wave transforms `reps` times (output fed back as input, in registers, so that nothing can be hoisted) one 960-point
// complex input taken from in[wave][960][2] (natural order, unscaled) and writes the spectrum of the LAST pass;
// clocks[wave] = shader clocks spent in the transforms.
template <int VARIANT>
__device__ __forceinline__ void fft_probe_body(const float *__restrict__ in, float *__restrict__ out,
                                               unsigned long long *__restrict__ clocks, int reps, const RnTablesDev &tb) {
  const int lane = threadIdx.x, w = blockIdx.x;
  const float2 *x = reinterpret_cast<const float2 *>(in) + (size_t)w * 960;
  float2 *y = reinterpret_cast<float2 *>(out) + (size_t)w * 960;
  float ar[15], ai[15];
  const int i0 = 15 * fft_lam(lane);
#pragma unroll
  for (int b = 0; b < 15; b++) {
    const float2 v = x[i0 + fft_c(b)];
    ar[b] = 0.0010416667f * v.x;
    ai[b] = 0.0010416667f * v.y;
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; r++) {
    if (r) {  // timing passes: the (rescaled) spectrum is the next input, in place -- nothing but the transform is timed
#pragma unroll
      for (int b = 0; b < 15; b++) {
        ar[b] *= 0.03125f;
        ai[b] *= 0.03125f;
      }
    }
    regfft960<VARIANT>(ar, ai, lane, reinterpret_cast<const float2 *>(tb.fft_tw));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int b = 0; b < 15; b++) y[64 * b + fft_pos(lane)] = make_float2(ar[b], ai[b]);
  if (lane == 0 && clocks) clocks[w] = t1 - t0;
}

template <int VARIANT>
__global__ void __launch_bounds__(64) rn_fft_probe_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                          unsigned long long *__restrict__ clocks, int reps, RnTablesDev tb) {
  fft_probe_body<VARIANT>(in, out, clocks, reps, tb);
}
// the DPP form held to 5 / 6 / 8 waves per SIMD (<= 96 / 80 / 64 VGPRs): what the transform costs inside a kernel with that budget
#define OCC_KERNEL(W)                                                                                                   \
  __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(W, W)))                                       \
  rn_fft_probe_occ##W##_kernel(const float *__restrict__ in, float *__restrict__ out, unsigned long long *__restrict__ clocks, \
                               int reps, RnTablesDev tb) {                                                              \
    fft_probe_body<1>(in, out, clocks, reps, tb);                                                                       \
  }
OCC_KERNEL(4)
OCC_KERNEL(5)
OCC_KERNEL(6)
OCC_KERNEL(8)

// what each exchange primitive delivers: out[variant][5 masks][64] = source lane seen by every lane
__global__ void __launch_bounds__(64) rn_xlane_probe_kernel(int *__restrict__ out) {
  const int lane = threadIdx.x;
  const float v = __int_as_float(lane + 1000);
#define PROBE(VAR, IDX, MASK) out[((VAR) * 6 + (IDX)) * 64 + lane] = __float_as_int(xlane_xor<MASK, VAR>(v, lane)) - 1000
  PROBE(0, 0, 1); PROBE(0, 1, 2); PROBE(0, 2, 4); PROBE(0, 3, 8); PROBE(0, 4, 16); PROBE(0, 5, 32);
  PROBE(1, 0, 1); PROBE(1, 1, 2); PROBE(1, 2, 4); PROBE(1, 3, 8); PROBE(1, 4, 16); PROBE(1, 5, 32);
#undef PROBE
}

extern "C" hipError_t rn_launch_fft_probe_lds(const float *, float *, unsigned long long *, int, int, const RnTablesDev *, hipStream_t);
extern "C" hipError_t rn_launch_fft_probe(int variant, const float *in, float *out, unsigned long long *clocks, int n, int reps,
                                          const RnTablesDev *tb, hipStream_t st) {
  if (variant == 2) return rn_launch_fft_probe_lds(in, out, clocks, n, reps, tb, st);
  if (variant == 14) hipLaunchKernelGGL(rn_fft_probe_occ4_kernel, dim3(n), dim3(64), 0, st, in, out, clocks, reps, *tb);
  else if (variant == 15) hipLaunchKernelGGL(rn_fft_probe_occ5_kernel, dim3(n), dim3(64), 0, st, in, out, clocks, reps, *tb);
  else if (variant == 16) hipLaunchKernelGGL(rn_fft_probe_occ6_kernel, dim3(n), dim3(64), 0, st, in, out, clocks, reps, *tb);
  else if (variant == 18) hipLaunchKernelGGL(rn_fft_probe_occ8_kernel, dim3(n), dim3(64), 0, st, in, out, clocks, reps, *tb);
  else if (variant == 0) hipLaunchKernelGGL(rn_fft_probe_kernel<0>, dim3(n), dim3(64), 0, st, in, out, clocks, reps, *tb);
  else hipLaunchKernelGGL(rn_fft_probe_kernel<1>, dim3(n), dim3(64), 0, st, in, out, clocks, reps, *tb);
  return hipGetLastError();
}
extern "C" hipError_t rn_launch_xlane_probe(int *out, hipStream_t st) {
  hipLaunchKernelGGL(rn_xlane_probe_kernel, dim3(1), dim3(64), 0, st, out);
  return hipGetLastError();
}
