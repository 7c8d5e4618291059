// fft_reg.h -- the 960-point kiss_fft of the reference (src/kiss_fft.c:518-586, factors 4.4.4.3.5) held entirely in
// registers: no LDS work area, no barriers, 15 independent butterflies in flight per lane.
//
// Layout.  Lane l of the wavefront holds 15 complex values a[blk], blk = 0..14: the work-area positions
// p = 64*blk + P(l).  The three radix-4 stages (m = 1, 4, 16) combine positions that differ in one base-4 digit of
// p mod 64, i.e. they are butterflies ACROSS LANES of the same register (digit = bits [1:0], [3:2], [5:4] of the lane
// number); the radix-3 (m = 64) and radix-5 (m = 192) stages combine blk = 3u+{0,1,2} and blk = t+{0,3,6,9,12}, i.e.
// they are entirely inside a lane.  Every output is the reference's exact expression tree (kiss_fft.c:101-306,
// _kiss_fft_guts.h:101-103; unfused multiplies and adds, -ffp-contract=off), so the bits are the oracle's:
// tools/proto/regfft_emul.py is the lane-level numpy model of this file, checked against oracle/rn_oracle.c:fft960.
//
// A cross-lane radix-4 butterfly on inputs v0..v3 (v1..v3 already multiplied by their twiddles by the lanes that own
// them), role k = the lane's digit:
//   level 1, partner k^2:  t = partner + s1*own        -> E+ = v0+v2 (k=0), O+ = v1+v3 (k=1), E- = v0-v2 (k=2), O- = v1-v3 (k=3)
//   role 3 rotates:        w = (k==3) ? (t.i, -t.r) : t
//   level 2, partner k^1:  out = partner_w + s2*w      -> out0 = E+ + O+ (k=0), out2 = E+ - O+ (k=1),
//                                                         out1 = E- + rot(O-) (k=2), out3 = E- - rot(O-) (k=3)
// with s1 = -1 for k >= 2, s2 = -1 for odd k (sign flips are exact: a - b == a + (-b)).  Each lane computes one
// output from two adds per component, the same operation count as the serial butterfly.  Roles 1 and 2 come out
// swapped, so after a stage the lane's digit of P(l) is bit-reversed; the later stages' twiddles are tabulated per lane
// with the true position (RnTablesDev::fft_tw), and at the end lane l holds bins 64*blk + fft_pos(l).
//
// Input.  Position p holds natural sample i with digitrev(i) = p (kiss_fft.c:314-346); for p = 64*blk + l that is
// i = 15*fft_lam(l) + fft_c(blk): every lane starts from 15 CONSECUTIVE samples (fft_c = {0,5,10,1,6,11,...}).
#pragma once
#include <hip/hip_runtime.h>

#ifndef RN_FFT_XLANE
#define RN_FFT_XLANE 2  // 0: every exchange through ds_bpermute (reference implementation); 1: DPP / swizzle forms, lane ^ 32
                        // through ds_bpermute; 2: as 1 with the lane ^ 32 level on v_permlane32_swap (no LDS)
#endif

struct rcpx { float r, i; };

// position (mod 64) of the bins a lane holds after the transform: each base-4 digit of the lane number bit-reversed
__device__ __forceinline__ int fft_pos(int lane) { return ((lane & 0x15) << 1) | ((lane & 0x2a) >> 1); }
// first natural sample index / 15 of the lane's input run: base-4 digit reversal of the lane number
__device__ __forceinline__ int fft_lam(int lane) { return (lane >> 4) + 4 * ((lane >> 2) & 3) + 16 * (lane & 3); }
// offset of block blk inside the lane's 15-sample input run
__host__ __device__ constexpr int fft_c(int blk) { return blk / 3 + 5 * (blk % 3); }

// value of `v` in lane (lane ^ MASK), MASK a power of two
template <int MASK, int VARIANT>
__device__ __forceinline__ float xlane_xor(float v, int lane) {
  const int x = __float_as_int(v);
  int r;
  if (VARIANT == 0) {
    r = __builtin_amdgcn_ds_bpermute((lane ^ MASK) << 2, x);
  } else if (MASK == 1) {
    r = __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false);  // quad_perm:[1,0,3,2]
  } else if (MASK == 2) {
    r = __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, false);  // quad_perm:[2,3,0,1]
  } else if (MASK == 4) {
    // lane ^ 4 = (lane ^ 3) ^ 7: a quad reversal, then the mirror of each half row of 8 -- both full-wave DPP patterns, so the second
    // one folds into the consuming add as its DPP operand (round 6; before: row_shl:4 into banks 0 and 2, row_shr:4 into banks 1 and
    // 3, on top of an initialised register -- three moves and a separate add per value, 30 values per transform)
    r = __builtin_amdgcn_update_dpp(0, x, 0x1B, 0xF, 0xF, true);    // quad_perm:[3,2,1,0] (bound_ctrl: every lane is written, no `old` to set up)
    r = __builtin_amdgcn_update_dpp(0, r, 0x141, 0xF, 0xF, false);  // row_half_mirror
  } else if (MASK == 8) {
    r = __builtin_amdgcn_update_dpp(0, x, 0x128, 0xF, 0xF, false);  // row_ror:8
  } else if (MASK == 16) {
    r = __builtin_amdgcn_ds_swizzle(x, 0x401F);  // bit mode, xor_mask 16 (inside each half of the wave)
  } else {  // MASK == 32 (variant 2 takes this level through v_permlane32_swap in fft_radix4_xlane and never gets here)
    r = __builtin_amdgcn_ds_bpermute((lane ^ MASK) << 2, x);
  }
  return __int_as_float(r);
}

__device__ __forceinline__ float fneg_if(float v, unsigned mask) { return __uint_as_float(__float_as_uint(v) ^ mask); }

// one cross-lane radix-4 stage on the digit at bit SHIFT of the lane number; tw = this lane's twiddle for the stage
// (TWIDDLED stages: m = 4, 16; the m = 1 stage has none, kiss_fft.c:112-131)
template <int SHIFT, bool TWIDDLED, int VARIANT>
__device__ __forceinline__ void fft_radix4_xlane(float (&ar)[15], float (&ai)[15], int lane, rcpx tw) {
  const int k = (lane >> SHIFT) & 3;
  const unsigned s1 = (k >= 2) ? 0x80000000u : 0u, s2 = (k & 1) ? 0x80000000u : 0u;
  const bool is0 = k == 0, is3 = k == 3;
#pragma unroll
  for (int b = 0; b < 15; b++) {
    float xr = ar[b], xi = ai[b];
    if (TWIDDLED) {  // role 0 is not multiplied in the reference: keep its bits (signed zeros included)
      const float mr = xr * tw.r - xi * tw.i, mi = xr * tw.i + xi * tw.r;
      xr = is0 ? xr : mr;
      xi = is0 ? xi : mi;
    }
    float tr, ti;
    if (VARIANT == 2 && SHIFT == 4) {
      // level 1 of the last cross-lane stage, partner = lane ^ 32, without the LDS crossbar (ds_bpermute: 6 LDS cycles per
      // value, profiles/r3_valu_issue.txt).  v_permlane32_swap exchanges the upper half of one register with the lower half
      // of another: after swap(xr, xi) lanes 0..31 hold {xr[l], xr[l + 32]} and lanes 32..63 {xi[l - 32], xi[l]}, so every
      // lane forms BOTH results of its pair -- partner + own for the lower lane (roles 0, 1), partner + (-own) = a - b for the
      // upper one (roles 2, 3; same operands, same rounding) -- and a second swap puts them where the butterfly wants them.
      typedef unsigned v2u_ __attribute__((ext_vector_type(2)));
      const v2u_ sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(xr), __float_as_uint(xi), false, false);
      const float lo = __uint_as_float(sw.x), hi = __uint_as_float(sw.y);   // lo = the pair's lower-lane value, hi = its upper-lane value
      const float sum = hi + lo, dif = lo - hi;
      const v2u_ sw2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(sum), __float_as_uint(dif), false, false);
      tr = __uint_as_float(sw2.x);
      ti = __uint_as_float(sw2.y);
    } else {
      tr = xlane_xor<(2 << SHIFT), VARIANT>(xr, lane) + fneg_if(xr, s1);
      ti = xlane_xor<(2 << SHIFT), VARIANT>(xi, lane) + fneg_if(xi, s1);
    }
    const float wr = is3 ? ti : tr;
    const float wi = is3 ? fneg_if(tr, 0x80000000u) : ti;
    ar[b] = xlane_xor<(1 << SHIFT), VARIANT>(wr, lane) + fneg_if(wr, s2);
    ai[b] = xlane_xor<(1 << SHIFT), VARIANT>(wi, lane) + fneg_if(wi, s2);
  }
}

// rows of RnTablesDev::fft_tw ([16][64] complex, one entry per lane)
#define RN_FTW_S2 0    // m = 4 stage:  tw[60 * (p mod 4)  * role]
#define RN_FTW_S3 1    // m = 16 stage: tw[15 * (p mod 16) * role]
#define RN_FTW_R3 2    // radix 3: tw[5q], tw[10q], q = fft_pos(lane)
#define RN_FTW_R5 4    // radix 5: 4 + 4t + (mult-1): tw[mult * (64t + q)], t = 0..2, mult = 1..4
#define RN_FTW_ROWS 16

// In: a[blk] = scaled sample 15*fft_lam(lane) + fft_c(blk).  Out: a[blk] = bin 64*blk + fft_pos(lane).
// ftw: RnTablesDev::fft_tw.  epi3i = twiddles[320].i, ya = twiddles[192], yb = twiddles[384] (kiss_fft.c:187,246-247).
template <int VARIANT>
__device__ __forceinline__ void regfft960(float (&ar)[15], float (&ai)[15], int lane, const float2 *__restrict__ ftw_) {
  // the table is global memory whatever the compiler knows about the pointer's origin (a caller that hides it behind an empty
  // asm would otherwise get flat_load, which counts on the LDS counter as well: every wait for an LDS result then also waits for
  // the twiddles)
  typedef float f2raw __attribute__((ext_vector_type(2)));
  const __attribute__((address_space(1))) f2raw *ftw_g = (const __attribute__((address_space(1))) f2raw *)ftw_;
  auto ftw = [&](int i) {
    const f2raw v = ftw_g[i];
    return make_float2(v.x, v.y);
  };
  const float2 t2 = ftw(RN_FTW_S2 * 64 + lane), t3 = ftw(RN_FTW_S3 * 64 + lane);
  fft_radix4_xlane<0, false, VARIANT>(ar, ai, lane, rcpx{1.f, 0.f});
  fft_radix4_xlane<2, true, VARIANT>(ar, ai, lane, rcpx{t2.x, t2.y});
  fft_radix4_xlane<4, true, VARIANT>(ar, ai, lane, rcpx{t3.x, t3.y});
  {  // radix 3, m = 64 (kiss_fft.c:201-225)
    const float2 w1 = ftw((RN_FTW_R3 + 0) * 64 + lane), w2 = ftw((RN_FTW_R3 + 1) * 64 + lane);
    const float epi3i = -0.86602540378443864676f;  // (float)sin(-2 pi / 3) = twiddles[5 * 64].i
#pragma unroll
    for (int u = 0; u < 5; u++) {
      const float f0r = ar[3 * u], f0i = ai[3 * u];
      const float x1r = ar[3 * u + 1], x1i = ai[3 * u + 1], x2r = ar[3 * u + 2], x2i = ai[3 * u + 2];
      const float s1r = x1r * w1.x - x1i * w1.y, s1i = x1r * w1.y + x1i * w1.x;
      const float s2r = x2r * w2.x - x2i * w2.y, s2i = x2r * w2.y + x2i * w2.x;
      const float s3r = s1r + s2r, s3i = s1i + s2i;
      float s0r = s1r - s2r, s0i = s1i - s2i;
      const float fmr = f0r - s3r * .5f, fmi = f0i - s3i * .5f;
      s0r *= epi3i;
      s0i *= epi3i;
      ar[3 * u] = f0r + s3r;
      ai[3 * u] = f0i + s3i;
      ar[3 * u + 2] = fmr + s0i;
      ai[3 * u + 2] = fmi - s0r;
      ar[3 * u + 1] = fmr - s0i;
      ai[3 * u + 1] = fmi + s0r;
    }
  }
  {  // radix 5, m = 192 (kiss_fft.c:269-302)
    const float yar = 0.30901699437494742410f, yai = -0.95105651629515357212f;   // twiddles[192] = exp(-2 pi i / 5)
    const float ybr = -0.80901699437494742410f, ybi = -0.58778525229247312917f;  // twiddles[384] = exp(-4 pi i / 5)
#pragma unroll
    for (int t = 0; t < 3; t++) {
      const float2 w1 = ftw((RN_FTW_R5 + 4 * t + 0) * 64 + lane), w2 = ftw((RN_FTW_R5 + 4 * t + 1) * 64 + lane);
      const float2 w3 = ftw((RN_FTW_R5 + 4 * t + 2) * 64 + lane), w4 = ftw((RN_FTW_R5 + 4 * t + 3) * 64 + lane);
      const float s0r = ar[t], s0i = ai[t];
      const float x1r = ar[t + 3], x1i = ai[t + 3], x2r = ar[t + 6], x2i = ai[t + 6];
      const float x3r = ar[t + 9], x3i = ai[t + 9], x4r = ar[t + 12], x4i = ai[t + 12];
      const float s1r = x1r * w1.x - x1i * w1.y, s1i = x1r * w1.y + x1i * w1.x;
      const float s2r = x2r * w2.x - x2i * w2.y, s2i = x2r * w2.y + x2i * w2.x;
      const float s3r = x3r * w3.x - x3i * w3.y, s3i = x3r * w3.y + x3i * w3.x;
      const float s4r = x4r * w4.x - x4i * w4.y, s4i = x4r * w4.y + x4i * w4.x;
      const float s7r = s1r + s4r, s7i = s1i + s4i, s10r = s1r - s4r, s10i = s1i - s4i;
      const float s8r = s2r + s3r, s8i = s2i + s3i, s9r = s2r - s3r, s9i = s2i - s3i;
      ar[t] = s0r + (s7r + s8r);
      ai[t] = s0i + (s7i + s8i);
      const float s5r = s0r + (s7r * yar + s8r * ybr), s5i = s0i + (s7i * yar + s8i * ybr);
      const float s6r = s10i * yai + s9i * ybi, s6i = -(s10r * yai + s9r * ybi);
      ar[t + 3] = s5r - s6r;
      ai[t + 3] = s5i - s6i;
      ar[t + 12] = s5r + s6r;
      ai[t + 12] = s5i + s6i;
      const float s11r = s0r + (s7r * ybr + s8r * yar), s11i = s0i + (s7i * ybr + s8i * yar);
      const float s12r = s9i * yai - s10i * ybi, s12i = s10r * ybi - s9r * yai;
      ar[t + 6] = s11r + s12r;
      ai[t + 6] = s11i + s12i;
      ar[t + 9] = s11r - s12r;
      ai[t + 9] = s11i - s12i;
    }
  }
}
