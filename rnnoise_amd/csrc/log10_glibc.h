// log10_glibc.h -- (double) log10 exactly as the reference's host computes it: GNU libc >= 2.28 on an x86-64 with FMA.
//
// The reference takes log10 of its band energies in double and rounds to float (src/denoise.c:383).  log10 is not a correctly
// rounded function: which double comes out is a property of the libm in use, and about 2^-27 of all arguments put the float
// rounding at the mercy of that double's last bits.  The device's own log10 (ocml) is a different algorithm; 0 differences in
// 1.2e7 inputs (round 4) bounds the rate, it does not make it zero.  So the product evaluates THE HOST'S algorithm, operation
// for operation -- every operation below is an IEEE double add, multiply or fused multiply-add, which gfx950 and x86 round
// identically:
//
//   log10(x)  = e_log10.c (__ieee754_log10): x = 2^k * m, m in [0.5, 1) for k < 0 else [1, 2);  y = k (+1 for k < 0);
//               z = y * log10_2lo + ivln10 * log(m);  return z + y * log10_2hi       -- three products, two sums, none fused
//   log(m)    = e_log.c (__log, the table method of ARM's optimized routines, N = 128) in the build every AVX2 host selects
//               (sysdeps/x86_64/fpu/multiarch/e_log-fma.c: compiled with -mfma -mavx2, so the compiler fused a*b+c where the
//               expression tree allows).  WHICH sums are fused is not in the C source; it is taken from the machine code of
//               libm.so.6 (GNU libc 2.35, Ubuntu 22.04: the image of both the build container and the MI355X boxes), instruction
//               by instruction -- the comments give the x86 instruction each line restates.
//
// Pinned three ways: tests/test_log10_cpu.py compiles this header for the host and sweeps it against the running libm (every
// path: the near-1 branch, both signs of k, subnormals, specials); rn_log10_selfcheck() does a short sweep when the library
// loads and falls back to the device's own log10 -- loudly -- on a host whose libm is a different one ($RNNOISE_AMD_LOG10);
// tests/test_gpu_at_size.py sweeps the device code against the host over 10^9 arguments.
#pragma once
#include <stdint.h>
#include <string.h>
#include "log10_glibc_data.h"

#ifdef __HIPCC__
#define RN_HD __host__ __device__ __forceinline__
#else
#define RN_HD static inline
#endif

RN_HD uint64_t rn_log_bits(double x) {
  uint64_t u;
  memcpy(&u, &x, 8);
  return u;
}
RN_HD double rn_log_dbl(uint64_t u) {
  double x;
  memcpy(&x, &u, 8);
  return x;
}

// __log of e_log.c, FMA build, for a positive NORMAL finite x (its zero / subnormal / negative / Inf / NaN paths are not restated:
// log10 below hands it a mantissa in [0.5, 2)).  tab = {invc, logc} x 128 (RN_LOG_TAB_VALUES)
RN_HD double rn_log_glibc_fma(double x, const double *tab) {
  const uint64_t ix = rn_log_bits(x);
  // |x - 1| small: x in [1 - 2^-4, 1 + 0x1.09p-4)
  if (ix - 0x3fee000000000000ull < 0x0003090000000000ull) {
    if (ix == 0x3ff0000000000000ull) return 0.0;
    const double r = x - 1.0;                                                         // vsubsd
    const double p12 = __builtin_fma(r, RN_LOG_B2, RN_LOG_B1);                        // vfmadd213sd  B1 + r B2
    const double p45 = __builtin_fma(r, RN_LOG_B5, RN_LOG_B4);                        // vfmadd213sd  B4 + r B5
    const double r2 = r * r;                                                          // vmulsd
    const double p78 = __builtin_fma(r, RN_LOG_B8, RN_LOG_B7);                        // vfmadd213sd  B7 + r B8
    const double p123 = __builtin_fma(r2, RN_LOG_B3, p12);                            // vfmadd231sd  ... + r2 B3
    const double p456 = __builtin_fma(r2, RN_LOG_B6, p45);                            // vfmadd231sd  ... + r2 B6
    const double r3 = r * r2;                                                         // vmulsd
    double p = __builtin_fma(r2, RN_LOG_B9, p78);                                     // vfmadd132sd  ... + r2 B9
    p = __builtin_fma(r3, RN_LOG_B10, p);                                             // vfmadd231sd  ... + r3 B10
    p = __builtin_fma(p, r3, p456);                                                   // vfmadd132sd
    p = __builtin_fma(p, r3, p123);                                                   // vfmadd132sd
    // hi + lo = r + r^2 B0 nearly exactly (w = r 2^27; rhi = r + w - w; the source's sums arrive fused)
    const double rw = __builtin_fma(r, 0x1p27, r);                                    // vfmadd132sd  r + w
    const double rhi = __builtin_fma(-0x1p27, r, rw);                                 // vfnmadd132sd (r + w) - w
    const double rhi2 = rhi * rhi;                                                    // vmulsd
    const double rlo = r - rhi;                                                       // vsubsd
    const double hi = __builtin_fma(rhi2, RN_LOG_B0, r);                              // vfmadd132sd  r + rhi rhi B0
    const double d = r - hi;                                                          // vsubsd
    const double rs = r + rhi;                                                        // vaddsd
    double lo = __builtin_fma(rhi2, RN_LOG_B0, d);                                    // vfmadd132sd  r - hi + w
    const double t = RN_LOG_B0 * rlo;                                                 // vmulsd
    lo = __builtin_fma(t, rs, lo);                                                    // vfmadd132sd  lo += B0 rlo (rhi + r)
    const double y = __builtin_fma(p, r3, lo);                                        // vfmadd132sd  r3 p + lo
    return hi + y;                                                                    // vaddsd
  }
  const uint64_t tmp = ix - 0x3fe6000000000000ull;
  const int i = (int)((tmp >> 45) & 127);
  const int k = (int)((int64_t)tmp >> 52);
  const uint64_t iz = ix - (tmp & 0xfff0000000000000ull);
  const double invc = tab[2 * i], logc = tab[2 * i + 1];
  const double z = rn_log_dbl(iz);
  const double kd = (double)k;
  const double r = __builtin_fma(z, invc, -1.0);                                      // vfmadd132sd  z/c - 1
  const double w = __builtin_fma(kd, RN_LOG_LN2HI, logc);                             // vfmadd213sd  k ln2hi + log c
  const double q12 = __builtin_fma(r, RN_LOG_A2, RN_LOG_A1);                          // vfmadd213sd  A1 + r A2
  const double hi = r + w;                                                            // vaddsd
  const double r2 = r * r;                                                            // vmulsd
  double lo = w - hi;                                                                 // vsubsd
  lo = lo + r;                                                                        // vaddsd
  lo = __builtin_fma(kd, RN_LOG_LN2LO, lo);                                           // vfmadd231sd
  const double rr2 = r * r2;                                                          // vmulsd
  const double q34 = __builtin_fma(r, RN_LOG_A4, RN_LOG_A3);                          // vfmadd132sd  A3 + r A4
  lo = __builtin_fma(r2, RN_LOG_A0, lo);                                              // vfmadd231sd
  const double q = __builtin_fma(q34, r2, q12);                                       // vfmadd132sd
  const double y = __builtin_fma(rr2, q, lo);                                         // vfmadd132sd
  return y + hi;                                                                      // vaddsd
}

// __ieee754_log10 of e_log10.c around it (the wrapper's domain handling folded in)
RN_HD double rn_log10_glibc_fma(double x, const double *tab) {
  uint64_t ix = rn_log_bits(x);
  int k = -1023;
  if ((int64_t)ix < (int64_t)0x0010000000000000ull) {  // x < 2^-1022 (or negative)
    if ((ix << 1) == 0) return -1.0 / 0.0;             // -1 / |x|
    if ((int64_t)ix < 0) return (x - x) / (x - x);     // NaN
    x = x * 0x1p54;
    k = -1023 - 54;
    ix = rn_log_bits(x);
  }
  if (ix > 0x7fefffffffffffffull) return x + x;  // Inf, NaN
  k += (int)(ix >> 52);
  const int64_t i = (int64_t)((uint64_t)(int64_t)k >> 63);
  const double y = (double)((int64_t)k + i);
  const double m = rn_log_dbl((ix & 0x000fffffffffffffull) | ((uint64_t)(0x3ff - i) << 52));
  const double l = rn_log_glibc_fma(m, tab);
  const double a = l * RN_LOG10_IVLN10;  // mulsd
  const double b = RN_LOG10_2LO * y;     // mulsd
  const double c = y * RN_LOG10_2HI;     // mulsd
  const double zz = b + a;               // addsd
  return zz + c;                         // addsd
}

#ifdef __HIPCC__
// The feature stage's expression (src/denoise.c:383): tab = RnTablesDev::log_tab, or null for the device library's own log10
__device__ __forceinline__ float rn_log_energy(float ex, const double *tab) {
  const double x = 1e-2 + (double)ex;
  return (float)(tab ? rn_log10_glibc_fma(x, tab) : log10(x));
}
#endif
