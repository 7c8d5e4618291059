// fft_probe.hip -- test / measurement kernels: the register-resident FFT (fft_reg.h) in isolation, the LDS work-area FFT it
// replaced, the exchange-primitive map, the device log10 sweep.  Compiled into the INSTRUMENTED library only
// (librnnoise_amd_instr.so, include/rnnoise_amd_debug.h); the product library does not contain this file.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fft_reg.h"
#include "rn_dev.h"
#include "log10_glibc.h"

#define WAVE 64
struct cpx { float r, i; };
#define RN_WSYNC()                                             \
  do {                                                         \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     \
    __builtin_amdgcn_wave_barrier();                           \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");     \
  } while (0)
__device__ __forceinline__ cpx cmul(cpx a, cpx b) {  // src/_kiss_fft_guts.h:101-103
  cpx m;
  m.r = a.r * b.r - a.i * b.i;
  m.i = a.r * b.i + a.i * b.r;
  return m;
}
__device__ __forceinline__ cpx cadd(cpx a, cpx b) { return {a.r + b.r, a.i + b.i}; }
__device__ __forceinline__ cpx csub(cpx a, cpx b) { return {a.r - b.r, a.i - b.i}; }

// ---- round 1's LDS work-area FFT: kept as the baseline the register FFT is measured against (tools/fft_bench.py, variant 2) ----
// The FFT work area is PADDED: logical element i lives at FPAD(i) = i + 2*(i/16), i.e. 16 bytes of padding
// after every 128 bytes.  Unpadded, the two middle radix-4 stages read elements 16*i+j / 64*i+j (j < 4 /
// j < 16) with the lanes of a wave spread over i, a 128-byte stride that lands on 8 (resp. 16) of the 64
// LDS banks -- 16-way / 4-way conflicts on every access, and all 16 waves of a CU run the FFT at the
// same time.  With the padding every stage spreads its 64 lanes evenly over the banks.
#define FPAD(i) ((i) + 2 * ((i) >> 4))
#define RN_FFT_PADDED (RN_WINDOW_SIZE + 2 * (RN_WINDOW_SIZE / 16))  // 1080 complex

// digit reversal for radices 5,3,4,4,4 (src/kiss_fft.c:314-346 on factors {5,192,3,64,4,16,4,4,4,1}),
// from a 960-entry u16 table built on the host that already holds the PADDED position
#define bitrev960(i) ((int)tb.bitrev[(i)])

// In-place 960-point forward FFT on LDS data already scaled by 1/960, digit-reversed and padded
// (src/kiss_fft.c:518-564 stage order 4,4,4,3,5; butterflies :101-306).  Butterflies of a
// stage are independent, so lanes take them round-robin; each butterfly is the reference's
// exact expression tree.  A butterfly's elements are m apart: within one 16-group for m = 1, 4 and
// whole groups apart otherwise, so the padded distance is a constant (18 per 16, 72 per 64, 216 per 192).
__device__ void fft960_lds(cpx *F, const cpx *__restrict__ tw, int lane) {
  RN_WSYNC();
  for (int b = lane; b < 240; b += WAVE) {  // radix-4, m=1, twiddle-free (:112-131)
    cpx *p = F + 4 * b + 2 * (b >> 2);
    cpx a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3];
    cpx s0 = csub(a0, a2), f0 = cadd(a0, a2), s1 = cadd(a1, a3);
    cpx f2 = csub(f0, s1);
    f0 = cadd(f0, s1);
    cpx d = csub(a1, a3);
    p[0] = f0;
    p[2] = f2;
    p[1] = {s0.r + d.i, s0.i - d.r};
    p[3] = {s0.r - d.i, s0.i + d.r};
  }
  RN_WSYNC();
#pragma unroll
  for (int stage = 0; stage < 2; stage++) {  // radix-4: (m=4, fstride 60), (m=16, fstride 15) (:141-165)
    const int m = stage ? 16 : 4, fs = stage ? 15 : 60;
    const int pm = stage ? 18 : 4;    // padded distance of m elements
    const int pmm = stage ? 72 : 18;  // padded distance of 4*m elements
    for (int b = lane; b < 240; b += WAVE) {
      int i = b / m, j = b % m;
      cpx *p = F + pmm * i + j;
      cpx s0 = cmul(p[pm], tw[fs * j]);
      cpx s1 = cmul(p[2 * pm], tw[2 * fs * j]);
      cpx s2 = cmul(p[3 * pm], tw[3 * fs * j]);
      cpx s5 = csub(p[0], s1), f0 = cadd(p[0], s1);
      cpx s3 = cadd(s0, s2), s4 = csub(s0, s2);
      p[2 * pm] = csub(f0, s3);
      p[0] = cadd(f0, s3);
      p[pm] = {s5.r + s4.i, s5.i - s4.r};
      p[3 * pm] = {s5.r - s4.i, s5.i + s4.r};
    }
    RN_WSYNC();
  }
  {  // radix-3, m=64, fstride 5 (:201-225)
    const float epi3i = tw[5 * 64].i;
    for (int b = lane; b < 320; b += WAVE) {
      int i = b >> 6, j = b & 63;
      cpx *p = F + 216 * i + FPAD(j);
      cpx s1 = cmul(p[72], tw[5 * j]);
      cpx s2 = cmul(p[144], tw[10 * j]);
      cpx s3 = cadd(s1, s2), s0 = csub(s1, s2);
      cpx f0 = p[0];
      cpx fm = {f0.r - s3.r * .5f, f0.i - s3.i * .5f};
      s0.r *= epi3i;
      s0.i *= epi3i;
      p[0] = cadd(f0, s3);
      p[144] = {fm.r + s0.i, fm.i - s0.r};
      p[72] = {fm.r - s0.i, fm.i + s0.r};
    }
    RN_WSYNC();
  }
  {  // radix-5, m=192, fstride 1 (:269-302)
    const cpx ya = tw[192], yb = tw[384];
    for (int j = lane; j < 192; j += WAVE) {
      cpx *p = F + FPAD(j);
      cpx s0 = p[0];
      cpx s1 = cmul(p[216], tw[j]);
      cpx s2 = cmul(p[432], tw[2 * j]);
      cpx s3 = cmul(p[648], tw[3 * j]);
      cpx s4 = cmul(p[864], tw[4 * j]);
      cpx s7 = cadd(s1, s4), s10 = csub(s1, s4), s8 = cadd(s2, s3), s9 = csub(s2, s3);
      cpx s5, s6, s11, s12;
      p[0] = {s0.r + (s7.r + s8.r), s0.i + (s7.i + s8.i)};
      s5.r = s0.r + (s7.r * ya.r + s8.r * yb.r);
      s5.i = s0.i + (s7.i * ya.r + s8.i * yb.r);
      s6.r = s10.i * ya.i + s9.i * yb.i;
      s6.i = -(s10.r * ya.i + s9.r * yb.i);
      p[216] = csub(s5, s6);
      p[864] = cadd(s5, s6);
      s11.r = s0.r + (s7.r * yb.r + s8.r * ya.r);
      s11.i = s0.i + (s7.i * yb.r + s8.i * ya.r);
      s12.r = s9.i * ya.i - s10.i * yb.i;
      s12.i = s10.r * yb.i - s9.r * ya.i;
      p[432] = cadd(s11, s12);
      p[648] = csub(s11, s12);
    }
    RN_WSYNC();
  }
}


// measurement twin of fft_probe.hip for the LDS work-area FFT above (same feedback protocol, 10 KB of LDS per wave like K1)
extern "C" __global__ void __launch_bounds__(WAVE)
rn_fft_probe_lds_kernel(const float *__restrict__ in, float *__restrict__ out, unsigned long long *__restrict__ clocks, int reps,
                        RnTablesDev tb) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  cpx *F = reinterpret_cast<cpx *>(smem_raw);
  const int lane = threadIdx.x, w = blockIdx.x;
  const float2 *x = reinterpret_cast<const float2 *>(in) + (size_t)w * 960;
  float2 *y = reinterpret_cast<float2 *>(out) + (size_t)w * 960;
  const cpx *tw = reinterpret_cast<const cpx *>(tb.twiddles);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = lane; i < 960; i += WAVE) {
    const float2 v = x[i];
    F[bitrev960(i)] = {0.0010416667f * v.x, 0.0010416667f * v.y};
  }
  for (int r = 0; r < reps; r++) {
    if (r) {  // timing passes: the (rescaled) spectrum is the next input, in place
      RN_WSYNC();
      for (int i = lane; i < 960; i += WAVE) {
        cpx v = F[FPAD(i)];
        F[FPAD(i)] = {0.03125f * v.r, 0.03125f * v.i};
      }
    }
    fft960_lds(F, tw, lane);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  for (int i = lane; i < 960; i += WAVE) y[i] = make_float2(F[FPAD(i)].r, F[FPAD(i)].i);
  if (lane == 0 && clocks) clocks[w] = t1 - t0;
}
extern "C" hipError_t rn_launch_fft_probe_lds(const float *in, float *out, unsigned long long *clocks, int n, int reps,
                                              const RnTablesDev *tb, hipStream_t st) {
  hipLaunchKernelGGL(rn_fft_probe_lds_kernel, dim3(n), dim3(WAVE), 10240, st, in, out, clocks, reps, *tb);
  return hipGetLastError();
}

// test tap: the log-energy expression of the feature stage (src/denoise.c:383) on arbitrary inputs -- ex[i], or, with ex == null,
// the float whose bit pattern is first_bits + i (exhaustive sweeps without an input array) -- through the feature stage's own
// function: tab = RnTablesDev::log_tab (the host libm's algorithm, log10_glibc.h) or null (the device library's log10)
extern "C" __global__ void rn_log_energy_kernel(const float *__restrict__ ex, unsigned first_bits, float *__restrict__ out, unsigned n,
                                                const double *tab) {
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    out[i] = rn_log_energy(ex ? ex[i] : __uint_as_float(first_bits + i), tab);
}
extern "C" hipError_t rn_launch_log_energy(const float *ex, unsigned first_bits, float *out, unsigned n, const double *tab, hipStream_t st) {
  hipLaunchKernelGGL(rn_log_energy_kernel, dim3(4096), dim3(256), 0, st, ex, first_bits, out, n, tab);
  return hipGetLastError();
}


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
    fft_probe_body<RN_FFT_XLANE>(in, out, clocks, reps, tb);                                                            \
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
  else if (variant == 3) hipLaunchKernelGGL(rn_fft_probe_kernel<2>, dim3(n), dim3(64), 0, st, in, out, clocks, reps, *tb);  // permlane32_swap form
  else if (variant == 0) hipLaunchKernelGGL(rn_fft_probe_kernel<0>, dim3(n), dim3(64), 0, st, in, out, clocks, reps, *tb);
  else hipLaunchKernelGGL(rn_fft_probe_kernel<1>, dim3(n), dim3(64), 0, st, in, out, clocks, reps, *tb);
  return hipGetLastError();
}
extern "C" hipError_t rn_launch_xlane_probe(int *out, hipStream_t st) {
  hipLaunchKernelGGL(rn_xlane_probe_kernel, dim3(1), dim3(64), 0, st, out);
  return hipGetLastError();
}
