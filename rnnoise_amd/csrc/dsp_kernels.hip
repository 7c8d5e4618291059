// dsp_kernels.hip -- analysis (K1) and synthesis (K3) kernels: one wavefront owns one
// 480-sample frame of one stream; thousands of independent streams fill the grid.
//
// Numerics contract (DESIGN.md "parity"): every float reduction that feeds a quantiser
// or a discrete decision keeps the reference's summation order -- lane = accumulator /
// lag / butterfly, serial chain over the reduction index -- and the file is compiled with
// -ffp-contract=off so that a*b+c is fused only where the reference's AVX2 intrinsics
// fuse.  The arithmetic follows oracle/rn_oracle.c operation by operation; reference
// citations (paths relative to /root/reference) are repeated at each stage.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "rn_dev.h"

#define WAVE 64
#define RN_K1_LEAN_MIN_STREAMS 3072  // see rn_analysis_lean_kernel
#define RN_K1_LEAN_MAX_STREAMS 24576

struct cpx { float r, i; };

__device__ __constant__ int c_eband[RN_NB_BANDS + 2] = {
    0,  2,  4,  6,  8,  10, 12, 15, 18,  21,  24,  28,  32,  36,  41,  47,  53,
    60, 68, 77, 87, 98, 110, 124, 140, 157, 176, 198, 223, 251, 282, 317, 356, 400};  // src/denoise.c:63-65

__device__ __constant__ int c_second_check[16] = {0, 0, 3, 2, 3, 2, 5, 2, 3, 2, 3, 2, 5, 2, 3, 2};  // src/pitch.c:422

__device__ __forceinline__ cpx cmul(cpx a, cpx b) {  // src/_kiss_fft_guts.h:101-103
  cpx m;
  m.r = a.r * b.r - a.i * b.i;
  m.i = a.r * b.i + a.i * b.r;
  return m;
}
__device__ __forceinline__ cpx cadd(cpx a, cpx b) { return {a.r + b.r, a.i + b.i}; }
__device__ __forceinline__ cpx csub(cpx a, cpx b) { return {a.r - b.r, a.i - b.i}; }

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
  __syncthreads();
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
  __syncthreads();
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
    __syncthreads();
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
    __syncthreads();
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
    __syncthreads();
  }
}

// Band energy / correlation (src/denoise.c:90-138).  The reference's interleaved loop adds, for
// every bin of band b, (1-frac)*tmp to sum[b] and frac*tmp to sum[b+1]; so accumulator k receives
// band k-1's `frac` parts in bin order, then band k's `1-frac` parts.  Here all 64 lanes first form
// the 800 products (each rounded exactly as in the reference) and lay them out so that accumulator
// k's sequence is contiguous (hi part of bin -> Q[eband[b+1]+bin], lo part -> Q[eband[b]+bin]); then
// lane k < 34 adds its sequence in order.  Padding steps add +0.0f, which changes no bit.
// Q: LDS scratch of >= 864 floats; sums: LDS scratch [34].
// XPAD / PPAD: the operand is an FFT work area (padded layout) rather than a plain array
template <bool XPAD, bool PPAD>
__device__ void band_accumulate(float *bandE, const cpx *X, const cpx *P, float *Q, float *sums,
                                const RnTablesDev &tb, int lane) {
#pragma unroll
  for (int t = 0; t < 7; t++) {  // 400 = 6.25 x 64; constant trip count so that the table / HBM loads overlap
    const int bin0 = lane + WAVE * t, bin = bin0 < 400 ? bin0 : 399;
    const int b = tb.band_of_bin[bin];
    const float frac = tb.band_frac[bin];
    const cpx x = X[XPAD ? FPAD(bin) : bin], y = P[PPAD ? FPAD(bin) : bin];
    float tmp = x.r * y.r;
    tmp += x.i * y.i;
    Q[c_eband[b + 1] + bin] = frac * tmp;  // lanes past the end redo bin 399 with the same values: no branch
    Q[c_eband[b] + bin] = (1 - frac) * tmp;
  }
  __syncthreads();
  {
    const int k = lane < RN_NB_BANDS + 2 ? lane : 0;
    const int lo = k ? c_eband[k - 1] : 0;
    const int start = lo + c_eband[k];
    const int len = lane < RN_NB_BANDS + 2 ? (k <= RN_NB_BANDS ? c_eband[k + 1] : 400) - lo : 0;
    const float *q = Q + start;
    float s = 0;
#pragma unroll 4
    for (int t = 0; t < 84; t++) {  // longest accumulator: 39 + 44 = 83 terms
      const float v = q[t];         // start + 83 <= 839 < 864
      s += (t < len) ? v : 0.f;
    }
    if (lane < RN_NB_BANDS + 2) sums[lane] = s;
  }
  __syncthreads();
  if (lane < RN_NB_BANDS) {
    float v = sums[lane + 1];
    if (lane == 0) v = (sums[0] + sums[1]) * 2 / 3;
    if (lane == RN_NB_BANDS - 1) v = (sums[RN_NB_BANDS] + sums[RN_NB_BANDS + 1]) * 2 / 3;
    bandE[lane] = v;
  }
  __syncthreads();
}

// src/denoise.c:160-170, lane i < 32 produces out[i]; c[j] = rnn_dct_table[j*32 + i], fetched by the
// caller well before use (the 32 loads are independent of the sum chain)
__device__ __forceinline__ float dct_lane(const float *in, const float *c, const RnTablesDev &tb) {
  float sum = 0;
#pragma unroll
  for (int j = 0; j < RN_NB_BANDS; j++) sum += in[j] * c[j];
  return (float)(sum * tb.dct_scale);
}

__device__ __forceinline__ float lane_bcast(float v, int l) {  // l must be wave-uniform
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// Best-two selection of find_best_pitch (src/pitch.c:62-92) given syy[i] = Syy before lag i.
// The reference walks the lags in order and tests each against the CURRENT second-best; the state only
// changes at a lag that passes that test.  So: all 64 lanes test their lag against the current state at
// once, the first passing lag is processed exactly like the reference does (second test, shift/replace),
// and the lanes after it are re-tested against the new state.  Lags between two passing ones saw the
// same state in the reference, hence identical decisions; a typical frame needs < 10 rounds for 147 lags.
__device__ __forceinline__ void best_pitch_select(const float *xcorr, const float *syy, int max_pitch, int &bp0,
                                                  int &bp1, int lane) {
  float bn0 = -1, bn1 = -1, bd0 = 0, bd1 = 0;
  int p0 = 0, p1 = 1;  // locals (not the reference parameters): keeps the selection in registers
  for (int base = 0; base < max_pitch; base += WAVE) {
    const int i = base + lane;
    const float xc = (i < max_pitch) ? xcorr[i] : 0.f;
    const float S = (i < max_pitch) ? syy[i] : 1.f;
    const float x16 = xc * 1e-12f;
    const float num = x16 * x16;
    unsigned long long live = __ballot(xc > 0);
    while (live) {
      const unsigned long long m = __ballot(num * bd1 > bn1 * S) & live;
      if (!m) break;
      const int bit = __ffsll((long long)m) - 1;
      const int idx = base + bit;
      const float n_ = lane_bcast(num, bit), S_ = lane_bcast(S, bit);
      // reference order: [1] was just tested, now [0]; a hit on [0] shifts the old best down (pitch.c:69-87)
      const bool top = n_ * bd0 > bn0 * S_;
      bn1 = top ? bn0 : n_;
      bd1 = top ? bd0 : S_;
      p1 = top ? p0 : idx;
      bn0 = top ? n_ : bn0;
      bd0 = top ? S_ : bd0;
      p0 = top ? idx : p0;
      live &= ~((2ull << bit) - 1ull);
    }
  }
  bp0 = p0;
  bp1 = p1;
}

// src/pitch.c:44-102 (float build), restructured so that only the genuinely serial part stays
// serial:  (1) all lanes form d[i] = y[i+len]^2 - y[i]^2 (each product rounded once, as in the
// reference);  (2) the running energy Syy -- a float recurrence with a clamp, hence order-bound -- is
// swept once from the start value Syy0 = 1 + sum_{j<len} y[j]^2 (computed by the caller inside a
// dot-product pass), 4 steps per LDS transaction, leaving Syy-before-step-i in syy[i];
// (3) best_pitch_select.  Used for the coarse (4x decimated) search; the fine search shares its sweep
// with yy_lookup (energy_sweeps below).  syy: scratch >= max_pitch rounded up to 4.
__device__ void find_best_pitch(const float *xcorr, const float *y, int len, int max_pitch, float Syy0, float *syy,
                                int &bp0, int &bp1, int lane) {
  const int mp4 = (max_pitch + 3) & ~3;
  for (int i = lane; i < mp4; i += WAVE) {
    const float a = (i + len < len + max_pitch) ? y[i + len] : 0.f, b = y[i];
    syy[i] = a * a - b * b;
  }
  float Syy = Syy0;
  __syncthreads();
  for (int i = 0; i < mp4; i += 4) {
    const float4 d = *reinterpret_cast<const float4 *>(syy + i);
    float4 o;
    o.x = Syy; Syy = fmaxf(1.f, Syy + d.x);  // MAX32(1, Syy): same value for every non-NaN Syy
    o.y = Syy; Syy = fmaxf(1.f, Syy + d.y);
    o.z = Syy; Syy = fmaxf(1.f, Syy + d.z);
    o.w = Syy; Syy = fmaxf(1.f, Syy + d.w);
    if (lane == 0) *reinterpret_cast<float4 *>(syy + i) = o;
  }
  __syncthreads();
  best_pitch_select(xcorr, syy, max_pitch, bp0, bp1, lane);
  __syncthreads();
}

// The two long running-energy recurrences of the pitch stage, swept TOGETHER (lane 0 / lane 1 of the
// same instructions; a one-lane VALU instruction costs as much issue time as a 64-lane one):
//   lane 0: Syy of the fine find_best_pitch (src/pitch.c:56-61,93-94; y = x_lp, len 480, 294 lags)
//           from syy0 = 1 + sum_{j<480} y[j]^2:  Syy = max(1, Syy + (y[i+480]^2 - y[i]^2))
//   lane 1: yy_lookup of remove_doubling (src/pitch.c:441-456; x = x_lp+384, N 480, 384 periods)
//           from xx = sum_{j<480} x[j]^2:  yy = (yy + x[-i]^2) - x[N-i]^2 (clamped copy stored)
// (both start values come out of the fine cross-correlation pass, as two extra chains).
// One step is  s = max(lo, (s + A) - B)  with (A, B, lo) = (d[i], 0, 1) for lane 0 and
// (x[-i]^2, x[N-i]^2, -inf) for lane 1; x - 0 and max(-inf, x) change no bit.
// rsq[k] = x_lp[863-k]^2 (864 floats): reversed, so that both of lane 1's operands walk UP it:
// A_i = rsq[479+i], B_i = rsq[i-1].  Results overwrite the A operand just consumed: afterwards
// Syy-before-lag-i = D[i-1] (D[-1] = syy0)  and  yy_lookup[i] = rsq[479+i] (rsq[479] = xx), clamped
// at 0 by a parallel pass.  D: 16-byte aligned, D[-1..295]; zero4: 4 floats.
__device__ __forceinline__ void energy_sweeps(const float *xlp, float *rsq, float *D, float *zero4, float syy0,
                                              float xx, int lane) {
  for (int k = lane; k < 864; k += WAVE) {
    const float v = xlp[863 - k];
    rsq[k] = v * v;
  }
  for (int i = lane; i < 296; i += WAVE) {
    const float a = xlp[i + 480], b = xlp[i];
    D[i] = a * a - b * b;
  }
  if (lane < 4) zero4[lane] = 0.f;
  __syncthreads();
  if (lane == 0) D[-1] = syy0;
  if (lane == 1) rsq[479] = xx;
  {
    float s = (lane == 0) ? syy0 : xx;
    float *pa = (lane == 0) ? D : rsq + 480;
    const float *pb = (lane == 0) ? zero4 : rsq;
    const int sb = (lane == 0) ? 0 : 4;
    const float lo = (lane == 0) ? 1.f : -__builtin_inff();
    float4 a = *reinterpret_cast<const float4 *>(pa), b = *reinterpret_cast<const float4 *>(pb);
    for (int j = 0; j < 384; j += 4) {
      pb += sb;
      const float4 an = *reinterpret_cast<const float4 *>(pa + j + 4);  // last one reads past the operands; unused
      const float4 bn = *reinterpret_cast<const float4 *>(pb);
      float4 o;
      s = fmaxf(lo, (s + a.x) - b.x); o.x = s;
      s = fmaxf(lo, (s + a.y) - b.y); o.y = s;
      s = fmaxf(lo, (s + a.z) - b.z); o.z = s;
      s = fmaxf(lo, (s + a.w) - b.w); o.w = s;
      if (lane == 1 || (lane == 0 && j < 296)) *reinterpret_cast<float4 *>(pa + j) = o;
      a = an;
      b = bn;
    }
  }
  __syncthreads();
  for (int i = 1 + lane; i <= 384; i += WAVE) rsq[479 + i] = fmaxf(0.f, rsq[479 + i]);  // MAX32(0, yy)
  __syncthreads();
}

__device__ __forceinline__ float pitch_gain(float xy, float xx, float yy) {  // src/pitch.c:416-419
  return (float)(xy / sqrt((double)(1 + xx * yy)));
}

// profiling taps (only when the debug record is armed): shader-clock delta since the previous tap
#define CLK_TAP(idx)                                                         \
  do {                                                                       \
    if (dbg) {                                                               \
      unsigned long long now_ = __builtin_amdgcn_s_memtime();                \
      if (lane == 0) dbg[RN_DBG_CLK + (idx)] = (float)(now_ - clk_prev);     \
      clk_prev = now_;                                                       \
    }                                                                        \
  } while (0)

// ---------------------------------------------------------------------------------------------
// K0: rnn_biquad (src/denoise.c:409-419, coefficients :469-470), transposed: lane = stream.
// The recurrence is strictly serial per stream (every step rounds its state to float), so the
// wave-per-frame kernel would idle 63 of 64 lanes for 480 steps; here 64 streams advance in
// lock-step instead.  Output goes straight into the stream's pitch ring (slot `slot`).
// a0*yi and a1*yi are products of two 24-bit significands, exact in double, so
// fma(-a, yi, b*xi) rounds once exactly like the reference's (b*xi - a*yi).
// ---------------------------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(WAVE)
rn_hp_kernel(RnGroupDev g, const float *__restrict__ in, int slot, int apply_hp) {
  const int s = blockIdx.x * WAVE + threadIdx.x;
  if (s >= g.n_streams) return;
  const float a0 = -1.99599f, a1 = 0.99600f, b0 = -2.f;
  const double na0 = -(double)a0, na1 = -(double)a1, b0d = (double)b0;
  float m0 = g.mem_hp[2 * s], m1 = g.mem_hp[2 * s + 1];
  const float4 *x = reinterpret_cast<const float4 *>(in + (size_t)s * RN_FRAME_SIZE);
  float4 *y = reinterpret_cast<float4 *>(g.pitch_ring + (size_t)s * RN_RING_SIZE + slot * RN_FRAME_SIZE);
  // 32 samples (one 128-byte line per stream) per block, the next block's 8 loads in flight while this one is
  // filtered: with one wave per SIMD nothing else hides the HBM round trip
  constexpr int BLK = 8;  // float4 per block
  float4 cur[BLK], nxt[BLK];
#pragma unroll
  for (int j = 0; j < BLK; j++) nxt[j] = x[j];
  for (int blk = 0; blk < RN_FRAME_SIZE / 4 / BLK; blk++) {
#pragma unroll
    for (int j = 0; j < BLK; j++) cur[j] = nxt[j];
    if (blk + 1 < RN_FRAME_SIZE / 4 / BLK) {
#pragma unroll
      for (int j = 0; j < BLK; j++) nxt[j] = x[(blk + 1) * BLK + j];
    }
#define HP_STEP(xi, yo)                                              \
    {                                                                \
      const float yi = (xi) + m0;                                    \
      const double xd = (double)(xi), yd = (double)yi;               \
      m0 = (float)((double)m1 + fma(na0, yd, b0d * xd));             \
      m1 = (float)fma(na1, yd, xd);                                  \
      (yo) = yi;                                                     \
    }
#pragma unroll
    for (int j = 0; j < BLK; j++) {
      const float4 v = cur[j];
      float4 o;
      if (apply_hp) {
        HP_STEP(v.x, o.x) HP_STEP(v.y, o.y) HP_STEP(v.z, o.z) HP_STEP(v.w, o.w)
      } else {
        o = v;  // training frames arrive already filtered by the caller's mixer (src/dump_features.c)
      }
      y[blk * BLK + j] = o;
    }
#undef HP_STEP
  }
  if (apply_hp) {
    g.mem_hp[2 * s] = m0;
    g.mem_hp[2 * s + 1] = m1;
  }

  // ---- rnn_pitch_downsample's serial half (src/pitch.c:146-214): 2x decimation, 5-lag autocorrelation
  // (src/celt_lpc.c:92-174), lag window, order-4 Levinson (src/celt_lpc.c:38-89) -> the 5 FIR taps.
  // In the wave-per-frame kernel these 5 chains of 864 steps used 5 lanes of 64; here every lane
  // streams its own pitch_buf once, keeping the last 4 decimated samples in registers.  For sample t
  // and lag k the product xlp[t-k]*xlp[t] is term i = t-k of the reference's sum for lag k: terms
  // i < 860 go to the main chain (rnn_pitch_xcorr over fastN), later ones to the tail chain `d`.
  {
    const float *ring = g.pitch_ring + (size_t)s * RN_RING_SIZE;
    const int ring0 = RN_RING0(slot);
    // pitch_buf in blocks of 32 floats (8 float4); ring0 and the ring size are multiples of 32, so a block never
    // straddles the wrap; the next block is requested before this one is consumed
    auto block = [&](int b, float4 (&dst)[BLK]) {
      int p = ring0 + 32 * b;
      p = (p >= RN_RING_SIZE) ? p - RN_RING_SIZE : p;
      const float4 *src = reinterpret_cast<const float4 *>(ring + p);
#pragma unroll
      for (int j = 0; j < BLK; j++) dst[j] = src[j];
    };
    float ac[5] = {0, 0, 0, 0, 0}, d[5] = {0, 0, 0, 0, 0};
    float w1 = 0, w2 = 0, w3 = 0, w4 = 0;  // xlp[t-1..t-4]; zeros before the start add exact +0 products
    float prev = 0;                          // pitch_buf[4c-1]
    block(0, nxt);
    for (int b = 0; b < RN_PITCH_BUF_SIZE / 32; b++) {
#pragma unroll
      for (int j = 0; j < BLK; j++) cur[j] = nxt[j];
      if (b + 1 < RN_PITCH_BUF_SIZE / 32) block(b + 1, nxt);
#pragma unroll
      for (int j = 0; j < BLK; j++) {
        const int c = b * BLK + j;
        const float4 v = cur[j];
        float xl[2];
        xl[0] = (c == 0) ? .5f * (.5f * (v.y) + v.x) : .5f * (.5f * (prev + v.y) + v.x);  // t = 2c
        xl[1] = .5f * (.5f * (v.y + v.w) + v.z);                                        // t = 2c+1
        prev = v.w;
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const int t = 2 * c + h;
          const float x0 = xl[h];
          if (t < 860) {
            ac[0] = ac[0] + x0 * x0;
            ac[1] = ac[1] + w1 * x0;
            ac[2] = ac[2] + w2 * x0;
            ac[3] = ac[3] + w3 * x0;
            ac[4] = ac[4] + w4 * x0;
          } else {  // t = 860..863: term i = t-k is < 860 for k > t-860, else it belongs to the tail
            const int e = t - 860;
            d[0] = d[0] + x0 * x0;
            if (e >= 1) d[1] = d[1] + x0 * w1; else ac[1] = ac[1] + w1 * x0;
            if (e >= 2) d[2] = d[2] + x0 * w2; else ac[2] = ac[2] + w2 * x0;
            if (e >= 3) d[3] = d[3] + x0 * w3; else ac[3] = ac[3] + w3 * x0;
            ac[4] = ac[4] + w4 * x0;
          }
          w4 = w3; w3 = w2; w2 = w1; w1 = x0;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 5; k++) ac[k] = ac[k] + d[k];
    ac[0] *= 1.0001f;
#pragma unroll
    for (int i = 1; i <= 4; i++) ac[i] -= ac[i] * (.008f * i) * (.008f * i);
    float lpc[4] = {0, 0, 0, 0};
    if (ac[0] != 0) {
      float error = ac[0];
      bool done = false;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        if (!done) {
          float rr = 0;
#pragma unroll
          for (int j = 0; j < i; j++) rr += lpc[j] * ac[i - j];
          rr += ac[i + 1];
          const float r = -rr / error;
          lpc[i] = r;
#pragma unroll
          for (int j = 0; j < (i + 1) >> 1; j++) {
            const float t1 = lpc[j], t2 = lpc[i - 1 - j];
            lpc[j] = t1 + r * t2;
            lpc[i - 1 - j] = t2 + r * t1;
          }
          error = error - (r * r) * error;
          if (error < .001f * ac[0]) done = true;  // `break` (celt_lpc.c:81-82)
        }
      }
    }
    float tmp = 1.f;
    const float c1 = .8f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      tmp = .9f * tmp;
      lpc[i] = lpc[i] * tmp;
    }
    float *o = g.lpc2 + ((size_t)slot * g.n_streams + s) * 8;  // one copy per ring slot: K0 runs up to 2 frames ahead of K1
    o[0] = lpc[0] + .8f;
    o[1] = lpc[1] + c1 * lpc[0];
    o[2] = lpc[2] + c1 * lpc[1];
    o[3] = lpc[3] + c1 * lpc[2];
    o[4] = c1 * lpc[3];
    if (g.debug) {
#pragma unroll
      for (int k = 0; k < 5; k++) g.debug[(size_t)s * RN_DBG_FLOATS + RN_DBG_AC + k] = ac[k];
    }
  }
}

// dot-product chain (src/pitch.h:51-142: one serial `sum = sum + x*y` per lag), n a multiple of 8,
// x 16-byte aligned, y arbitrary; both may differ per lane.  The next 8 operand pairs are fetched
// from LDS while the current 8 are being added, so the LDS round trip is off the chain.
// s0 is the chain's start value: a lane with x == y and s0 = 1 computes a find_best_pitch start
// energy 1 + sum y[j]^2 (src/pitch.c:56-61) in the same instructions as the real dot products.
#define OPAQUE(v) asm("" : "+v"(v))
__device__ __forceinline__ float chain_dot8(const float *x, const float *y, int n, float s0 = 0.f) {
  float s = s0;
  float4 xa = *reinterpret_cast<const float4 *>(x), xb = *reinterpret_cast<const float4 *>(x + 4);
  float ya[8];
#pragma unroll
  for (int k = 0; k < 8; k++) ya[k] = y[k];
  for (int i = 0; i < n; i += 8) {
    const int nx = (i + 8 < n) ? i + 8 : i;
    const float4 xc = *reinterpret_cast<const float4 *>(x + nx), xd = *reinterpret_cast<const float4 *>(x + nx + 4);
    float yn[8];
#pragma unroll
    for (int k = 0; k < 8; k++) yn[k] = y[nx + k];
    // each product passes through an empty asm: the SLP vectoriser would otherwise pair the multiplies
    // into v_pk_mul_f32 and pay ~1.3 register shuffles per step to feed them
    float p0 = xa.x * ya[0], p1 = xa.y * ya[1], p2 = xa.z * ya[2], p3 = xa.w * ya[3];
    float p4 = xb.x * ya[4], p5 = xb.y * ya[5], p6 = xb.z * ya[6], p7 = xb.w * ya[7];
    OPAQUE(p0); OPAQUE(p1); OPAQUE(p2); OPAQUE(p3); OPAQUE(p4); OPAQUE(p5); OPAQUE(p6); OPAQUE(p7);
    s = s + p0;
    s = s + p1;
    s = s + p2;
    s = s + p3;
    s = s + p4;
    s = s + p5;
    s = s + p6;
    s = s + p7;
    xa = xc;
    xb = xd;
#pragma unroll
    for (int k = 0; k < 8; k++) ya[k] = yn[k];
  }
  return s;
}

// One 10 KB LDS arena per wave (16 waves = one full round per CU at 4096 streams), time-shared
// (float offsets, the SCR_* constants below):
//   FFT phases   : F = [0,2160) (960 complex, padded layout); the band products Q live in [1084,1948),
//                  above the 481 bins that matter; small per-frame vectors in [2392,2560)
//   coarse search: xlp [0,864) | y4 [1296,1728) | interleaved pairs Z [1728,2334)
//                  during the 147 chains, then running energies [1728,1876) and xcorr [2028,2175)
//   fine search  : xlp | reversed squares -> yy_lookup [864,1728) | energy increments -> Syy [1731,2028)
//                  | xcorr [2028,2324) | 4 zeros [2324,2328); the 64 doubling dots reuse [2120,2184)
// two independent dot-product chains per lane (lags l and l+64 against the same x): the pair is
// written as 2-wide vector arithmetic so that it compiles to v_pk_mul_f32 / v_pk_add_f32 -- half the
// VALU instructions of two scalar chains; each component is still mul-then-add in the reference order.
// z[k] = {y[k], y[k+64]} comes from an interleaved copy, so a pair is one 8-byte LDS read that lands
// in an aligned register pair (built from two arrays the pairs cost ~2.5 moves per step).
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f chain_dot8_x2(const float *x, const v2f *z, int n) {
  v2f s = {0.f, 0.f};
  float4 xa = *reinterpret_cast<const float4 *>(x), xb = *reinterpret_cast<const float4 *>(x + 4);
  v2f y[8];
#pragma unroll
  for (int k = 0; k < 8; k++) y[k] = z[k];
#pragma unroll 2
  for (int i = 0; i < n; i += 8) {
    const int nx = (i + 8 < n) ? i + 8 : i;
    const float4 xc = *reinterpret_cast<const float4 *>(x + nx), xd = *reinterpret_cast<const float4 *>(x + nx + 4);
    v2f yn[8];
#pragma unroll
    for (int k = 0; k < 8; k++) yn[k] = z[nx + k];
    s = s + v2f{xa.x, xa.x} * y[0];
    s = s + v2f{xa.y, xa.y} * y[1];
    s = s + v2f{xa.z, xa.z} * y[2];
    s = s + v2f{xa.w, xa.w} * y[3];
    s = s + v2f{xb.x, xb.x} * y[4];
    s = s + v2f{xb.y, xb.y} * y[5];
    s = s + v2f{xb.z, xb.z} * y[6];
    s = s + v2f{xb.w, xb.w} * y[7];
    xa = xc;
    xb = xd;
#pragma unroll
    for (int k = 0; k < 8; k++) y[k] = yn[k];
  }
  return s;
}

struct AnalysisLds {
  float a[2560];
};
#define SCR_XLP 0
#define SCR_SQ 864    // [864]  fine search: reversed squares of xlp, later yy_lookup
#define SCR_Y4 1296   // [432]  4x-decimated signal (coarse search only)
#define SCR_Z 1728    // [606]  {y4[j], y4[j+64]} pairs for the packed coarse chains
#define SCR_SYY 1728  // [148]  running energies of the coarse find_best_pitch
#define SCR_D 1732    // [-1..295] fine search: Syy increments, then Syy itself (16-byte aligned)
#define SCR_XC 2028   // [296]  xcorr[] of pitch_search
#define SCR_ZERO 2324 // [4]
#define SCR_DOTS 2120 // [64]
#define SCR_Q 1084    // [864]  band products (above the padded bins 0..480 = floats [0,1082))
#define SCR_MISC 2392 // sums[40] | Ex[32] | Ep[32] | Exp[32] | Ly[32]

// ---------------------------------------------------------------------------------------------
// K1: rnn_compute_frame_features (src/denoise.c:347-398) on the high-passed frame that K0 put
// into the pitch ring.  grid = n_streams blocks of one wavefront; ~12 KB of LDS per wave.
// `ring0` = physical ring position of pitch_buf[0] (src/denoise.c:359-360 shift = ring rotation).
// ---------------------------------------------------------------------------------------------
template <bool TRAIN>
__device__ __forceinline__ void analysis_body(const RnGroupDev &g, const RnTablesDev &tb, int slot, int parity,
                                              const RnTrainArgs &tr) {
  const int ring0 = RN_RING0(slot);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  AnalysisLds &L = *reinterpret_cast<AnalysisLds *>(smem_raw);
  const int s = blockIdx.x, lane = threadIdx.x;
  const cpx *tw = reinterpret_cast<const cpx *>(tb.twiddles);
  float *scr = L.a;
  cpx *F = reinterpret_cast<cpx *>(L.a);
  float *xlp = scr + SCR_XLP, *Qs = scr + SCR_Q;
  float *sums = scr + SCR_MISC, *Ex = sums + 40, *Ep = Ex + 32, *Exp = Ep + 32, *Ly = Exp + 32;
  float *dbg = g.debug ? g.debug + (size_t)s * RN_DBG_FLOATS : nullptr;
  unsigned long long clk_prev = dbg ? __builtin_amdgcn_s_memtime() : 0;
  const float *ring = g.pitch_ring + (size_t)s * RN_RING_SIZE;
  auto pb_at = [&](int i) {  // pitch_buf[i], i in [0, 1728): ring0 + i < 2 * RN_RING_SIZE, one conditional wrap
    int p = ring0 + i;
    p = (p >= RN_RING_SIZE) ? p - RN_RING_SIZE : p;
    return ring[p];
  };
#define PB(i) pb_at(i)

  CLK_TAP(0);
  CLK_TAP(1);
  // ---- rnn_frame_analysis (src/denoise.c:332-345): window [prev | cur], FFT, Ex ----
#pragma unroll 5
  for (int t = 0; t < RN_WINDOW_SIZE / WAVE; t++) {  // constant trip count: HBM/L2 round trips overlap 5 x 3 at a time
    const int i = lane + WAVE * t;
    float w = tb.half_window[i < RN_FRAME_SIZE ? i : RN_WINDOW_SIZE - 1 - i];
    float v = PB(RN_PITCH_BUF_SIZE - RN_WINDOW_SIZE + i) * w;
    F[bitrev960(i)] = {0.0010416667f * v, 0.0010416667f * 0.f};
  }
  fft960_lds(F, tw, lane);
  if (TRAIN) {  // band limit of the TRAINING build (src/denoise.c:340-343)
    for (int i = tr.lowpass[s] + lane; i < RN_FREQ_SIZE; i += WAVE) F[FPAD(i)] = {0.f, 0.f};
    __syncthreads();
  }
  float *gX = g.spec_X[parity] + (size_t)s * RN_SPEC_STRIDE;
  for (int i = lane; i < RN_FREQ_SIZE; i += WAVE) {
    cpx v = F[FPAD(i)];
    gX[2 * i] = v.r;
    gX[2 * i + 1] = v.i;
  }
  band_accumulate<true, true>(Ex, F, F, Qs, sums, tb, lane);

  CLK_TAP(2);  // window + FFT(X) + Ex
  // ---- rnn_pitch_downsample (src/pitch.c:146-214) ----
#pragma unroll 7
  for (int t = 0; t < 14; t++) {  // 864 = 13.5 x 64; constant trip count so that the loads overlap (7 x 3 at a time)
    const int i0 = lane + WAVE * t, i = i0 < 864 ? i0 : 863;  // clamp, not a branch
    const float a = PB(i ? 2 * i - 1 : 0), b = PB(2 * i), c = PB(2 * i + 1);
    float v = .5f * (.5f * (a + c) + b);
    if (t == 0) v = (i == 0) ? .5f * (.5f * c + b) : v;  // the first output has no left neighbour (src/pitch.c:166)
    xlp[i] = v;  // lanes past the end recompute and rewrite element 863 with the same value: no branch
  }
  __syncthreads();
  // the 5 FIR taps (autocorrelation + Levinson) were computed by the lane-per-stream kernel K0
  float lpc2[5];
#pragma unroll
  for (int k = 0; k < 5; k++) lpc2[k] = g.lpc2[((size_t)slot * g.n_streams + s) * 8 + k];
  if (dbg && lane < 5) dbg[RN_DBG_LPC + lane] = lpc2[lane];
  {  // celt_fir5 in place (src/pitch.c:104-143): outputs are independent given the OLD samples
    float r[14];
#pragma unroll
    for (int t = 0; t < 14; t++) {
      int i = lane + WAVE * t;
      float sum = 0;
      if (i < 864) {
        sum = xlp[i];
        sum = sum + lpc2[0] * (i >= 1 ? xlp[i - 1] : 0.f);
        sum = sum + lpc2[1] * (i >= 2 ? xlp[i - 2] : 0.f);
        sum = sum + lpc2[2] * (i >= 3 ? xlp[i - 3] : 0.f);
        sum = sum + lpc2[3] * (i >= 4 ? xlp[i - 4] : 0.f);
        sum = sum + lpc2[4] * (i >= 5 ? xlp[i - 5] : 0.f);
      }
      r[t] = sum;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 14; t++) {
      int i = lane + WAVE * t;
      if (i < 864) xlp[i] = r[t];
    }
  }
  __syncthreads();

  CLK_TAP(3);  // downsample + autocorr + LPC + FIR
  if (dbg) for (int i = lane; i < 864; i += WAVE) dbg[RN_DBG_XLP + i] = xlp[i];
  // ---- rnn_pitch_search (src/pitch.c:281-385), len 960, max_pitch 588 ----
  float *y4 = scr + SCR_Y4, *xc = scr + SCR_XC;
  // 4x decimated lp[2j], j<432: y_lp4 = y4[0..386], x_lp4 = y4[192..431] (src/pitch.c:309-312)
  for (int j = lane; j < 432; j += WAVE) y4[j] = xlp[2 * j];
  float syy0_coarse;
  {
    v2f *Z = reinterpret_cast<v2f *>(scr + SCR_Z);
    for (int j = lane; j < 303; j += WAVE) Z[j] = v2f{xlp[2 * j], xlp[2 * j + 128]};
    __syncthreads();
    // 147 lags: lanes take lags (l, l+64) as a packed pair, then the 19 lags 128..146
    const v2f p = chain_dot8_x2(y4 + 192, Z + lane, 240);
    // lanes 0..18: lags 128..146; lane 19: the start energy 1 + sum y4[j]^2 of the coarse find_best_pitch
    float q = 0;
    if (lane < 20) q = chain_dot8(lane < 19 ? y4 + 192 : y4, lane < 19 ? y4 + lane + 128 : y4, 240, lane < 19 ? 0.f : 1.f);
    __syncthreads();  // Z is dead; xcorr goes into its area
    xc[lane] = p.x;
    xc[lane + 64] = p.y;
    if (lane < 147 - 128) xc[lane + 128] = q;
    syy0_coarse = lane_bcast(q, 19);
  }
  __syncthreads();
  int bp0, bp1;
  CLK_TAP(4);  // coarse xcorr
  find_best_pitch(xc, y4, 240, 147, syy0_coarse, scr + SCR_SYY, bp0, bp1, lane);
  CLK_TAP(5);  // coarse best-pitch scan
  if (dbg) {
    for (int i = lane; i < 147; i += WAVE) dbg[RN_DBG_XC_COARSE + i] = xc[i];
    if (lane == 0) { dbg[RN_DBG_BEST] = bp0; dbg[RN_DBG_BEST + 1] = bp1; }
  }
  __syncthreads();
  for (int i = lane; i < 294; i += WAVE) xc[i] = 0;
  __syncthreads();
  float xx, syy0_fine;
  {  // lanes 0..9: the fine lags; lane 10: xx = <x, x> of remove_doubling; lane 11: 1 + sum x_lp[j]^2 (fine Syy start)
    int c = (lane < 5) ? (2 * bp0 - 2 + lane) : (2 * bp1 - 2 + (lane - 5));
    const bool lag = lane < 10 && c >= 0 && c < 294;
    float sum = 0;
    if (lag || lane == 10 || lane == 11)
      sum = chain_dot8(lane == 11 ? xlp : xlp + 384, lag ? xlp + c : (lane == 11 ? xlp : xlp + 384), 480,
                       lane == 11 ? 1.f : 0.f);
    if (lag) xc[c] = (-1 > sum) ? -1 : sum;
    xx = lane_bcast(sum, 10);
    syy0_fine = lane_bcast(sum, 11);
  }
  __syncthreads();
  CLK_TAP(6);  // fine xcorr (+ the two start energies)
  // running energies of the fine search and of remove_doubling, one shared sweep (y4 is dead)
  float *rsq = scr + SCR_SQ, *Dsyy = scr + SCR_D;
  energy_sweeps(xlp, rsq, Dsyy, scr + SCR_ZERO, syy0_fine, xx, lane);
  const float *yyl = rsq + 479;  // yy_lookup[i], i = 0..384
  CLK_TAP(9);  // fine-search Syy + yy_lookup sweeps
  best_pitch_select(xc, Dsyy - 1, 294, bp0, bp1, lane);
  CLK_TAP(7);  // fine best-pitch selection
  int offset = 0;
  if (bp0 > 0 && bp0 < 293) {
    float a = xc[bp0 - 1], b = xc[bp0], c = xc[bp0 + 1];
    if ((c - a) > .7f * (b - a)) offset = 1;
    else if ((a - c) > .7f * (b - c)) offset = -1;
  }
  int pitch_index = RN_PITCH_MAX_PERIOD - (2 * bp0 - offset);
  if (dbg) {
    for (int i = lane; i < 294; i += WAVE) dbg[RN_DBG_XC_FINE + i] = xc[i];
    if (lane == 0) { dbg[RN_DBG_BEST + 2] = bp0; dbg[RN_DBG_BEST + 3] = bp1; dbg[RN_DBG_BEST + 4] = offset; dbg[RN_DBG_BEST + 5] = pitch_index; }
  }

  // ---- rnn_remove_doubling (src/pitch.c:423-528): maxperiod 384, minperiod 30, N 480 ----
  float pgain;
  {
    const int maxperiod = 384, minperiod = 30, N = 480, minperiod0 = RN_PITCH_MIN_PERIOD;
    const int *sc = c_second_check;
    const float *x = xlp + maxperiod;
    float *dots = scr + SCR_DOTS;
    int T0 = pitch_index / 2;
    const int prev_period = g.last_period[s] / 2;
    const float prev_gain = g.last_gain[s];
    if (T0 >= maxperiod) T0 = maxperiod - 1;
    int T = T0;
    __syncthreads();  // the fine xcorr is dead from here on; its area becomes the dots
    // every dot product the routine can ask for, in ONE pass of 480-step chains (each chain is an
    // independent serial sum, so computing it speculatively changes no bit):
    //   lane 1: xy(T0);  lanes 2..29: (k, T1 / T1b), k = 2..15 (pitch.c:462-483);
    //   lanes 32..61: the +-1 neighbours of every period the decision loop can end on
    //   (T0 and T1(k)), needed by the final 3-point refinement (pitch.c:511-512).
    //   (xx, the chain at offset 0, came out of energy_sweeps)
    {
      int off = -1;
      if (lane == 1) off = T0;
      else if (lane >= 2 && lane < 30) {
        int k = 2 + ((lane - 2) >> 1);
        int T1 = (2 * T0 + k) / (2 * k), T1b;
        if (k == 2) T1b = (T1 + T0 > maxperiod) ? T0 : T0 + T1;
        else T1b = (2 * sc[k] * T0 + k) / (2 * k);
        off = ((lane - 2) & 1) ? T1b : T1;
      } else if (lane >= 32 && lane < 62) {
        const int c = (lane - 32) >> 1;  // candidate 0: T0; candidate c >= 1: T1(k = c + 1)
        const int Tc = c ? (2 * T0 + (c + 1)) / (2 * (c + 1)) : T0;
        off = Tc + (((lane - 32) & 1) ? 1 : -1);
        if (off < 0) off = 0;  // only for candidates the decision loop never selects (T1 < minperiod)
      }
      if (off >= 0) dots[lane] = chain_dot8(x, x - off, N);
    }
    __syncthreads();
    float xy = dots[1];
    CLK_TAP(8);  // 59 candidate dot products of remove_doubling
    float yy = yyl[T0];
    float best_xy = xy, best_yy = yy;
    if (dbg && lane == 0) { dbg[RN_DBG_DOTS] = xx; dbg[RN_DBG_DOTS + 1] = xy; dbg[RN_DBG_DOTS + 2] = yy; }
    const float g0 = pitch_gain(xy, xx, yy);
    float gg = g0;
    int cand = 0;  // which candidate won: 0 = T0, c = k-1 for T1(k)
    {
      // The reference loop (pitch.c:462-500) runs k = 2..15, stops at the first T1 < minperiod and keeps the
      // LAST k whose gain beats its threshold.  Nothing in an iteration depends on an earlier one, and T1 =
      // floor(T0/k + 1/2) does not increase with k, so: lane k evaluates iteration k, and the winner is the
      // highest lane that is both before the stop and over its threshold.
      const int k = lane < 2 ? 2 : (lane > 15 ? 15 : lane);
      const int T1 = (2 * T0 + k) / (2 * k);
      int T1b;
      if (k == 2) T1b = (T1 + T0 > maxperiod) ? T0 : T0 + T1;
      else T1b = (2 * sc[k] * T0 + k) / (2 * k);
      float xy1 = dots[2 + 2 * (k - 2)], xy2 = dots[3 + 2 * (k - 2)];
      xy1 = .5f * (xy1 + xy2);
      const float yy1 = .5f * (yyl[T1] + yyl[T1b]);
      const float g1 = pitch_gain(xy1, xx, yy1);
      float cont;
      int dT = T1 - prev_period;
      dT = dT < 0 ? -dT : dT;
      if (dT <= 1) cont = prev_gain;
      else if (dT <= 2 && 5 * k * k < T0) cont = .5f * prev_gain;
      else cont = 0;
      float thresh = (.3f > .7f * g0 - cont) ? .3f : .7f * g0 - cont;
      if (T1 < 3 * minperiod) thresh = (.4f > .85f * g0 - cont) ? .4f : .85f * g0 - cont;
      else if (T1 < 2 * minperiod) thresh = (.5f > .9f * g0 - cont) ? .5f : .9f * g0 - cont;
      const bool hit = lane >= 2 && lane <= 15 && T1 >= minperiod && g1 > thresh;
      const unsigned long long m = __ballot(hit);
      if (m) {
        const int kb = 63 - __clzll((long long)m);
        best_xy = lane_bcast(xy1, kb);
        best_yy = lane_bcast(yy1, kb);
        gg = lane_bcast(g1, kb);
        T = __builtin_amdgcn_readlane(T1, kb);
        cand = kb - 1;
      }
    }
    best_xy = (0 > best_xy) ? 0 : best_xy;
    float pg;
    if (best_yy <= best_xy) pg = 1.f;
    else pg = best_xy / (best_yy + 1);
    // 3-point refinement around the selected period: xcorr[k] = <x, x-(T+k-1)> (pitch.c:511-512)
    float xc1 = cand ? dots[2 + 2 * (cand - 1)] : dots[1];
    float xc0 = dots[32 + 2 * cand], xc2 = dots[33 + 2 * cand];
    if (dbg && lane == 0) { dbg[RN_DBG_DOTS + 3] = T; dbg[RN_DBG_DOTS + 4] = xc0; dbg[RN_DBG_DOTS + 5] = xc1; dbg[RN_DBG_DOTS + 6] = xc2; }
    int off2 = 0;
    if ((xc2 - xc0) > .7f * (xc1 - xc0)) off2 = 1;
    else if ((xc0 - xc2) > .7f * (xc1 - xc2)) off2 = -1;
    if (pg > gg) pg = gg;
    pitch_index = 2 * T + off2;
    if (pitch_index < minperiod0) pitch_index = minperiod0;
    pgain = pg;
    __syncthreads();
  }
  if (lane == 0) {
    g.last_period[s] = pitch_index;
    g.last_gain[s] = pgain;
    g.pitch[s] = pitch_index;
  }

  CLK_TAP(10);  // doubling decisions + 3 final dots
  // ---- pitch-aligned frame -> P, Ep, Exp (src/denoise.c:371-377) ----
#pragma unroll 5
  for (int t = 0; t < RN_WINDOW_SIZE / WAVE; t++) {  // constant trip count: HBM/L2 round trips overlap 5 x 3 at a time
    const int i = lane + WAVE * t;
    float w = tb.half_window[i < RN_FRAME_SIZE ? i : RN_WINDOW_SIZE - 1 - i];
    float v = PB(RN_PITCH_BUF_SIZE - RN_WINDOW_SIZE - pitch_index + i) * w;
    F[bitrev960(i)] = {0.0010416667f * v, 0.0010416667f * 0.f};
  }
  {
    // a second, opaque copy of the pointer: otherwise the per-lane twiddles of the first FFT are kept alive
    // (and spilled to scratch) across the whole pitch stage instead of being re-read from L1/L2
    const cpx *tw2 = tw;
    asm volatile("" : "+s"(tw2));
    fft960_lds(F, tw2, lane);
  }
  float *gP = g.spec_P[parity] + (size_t)s * RN_SPEC_STRIDE;
  for (int i = lane; i < RN_FREQ_SIZE; i += WAVE) {
    cpx v = F[FPAD(i)];
    gP[2 * i] = v.r;
    gP[2 * i + 1] = v.i;
  }
  float dctc[RN_NB_BANDS];  // this lane's DCT column, requested now, consumed after the band energies
#pragma unroll
  for (int j = 0; j < RN_NB_BANDS; j++) dctc[j] = tb.dct[j * RN_NB_BANDS + (lane & 31)];
  band_accumulate<true, true>(Ep, F, F, Qs, sums, tb, lane);
  // X is read back from HBM/L2 (this block wrote it; the barriers since then make it visible)
  band_accumulate<false, true>(Exp, reinterpret_cast<const cpx *>(gX), F, Qs, sums, tb, lane);
  float *gE = g.spec_E[parity] + (size_t)s * 96;
  if (lane < RN_NB_BANDS) {
    Exp[lane] = (float)((double)Exp[lane] / sqrt(.001 + (double)(Ex[lane] * Ep[lane])));
    gE[lane] = Ex[lane];
    gE[32 + lane] = Ep[lane];
    gE[64 + lane] = Exp[lane];
  }
  __syncthreads();

  // ---- features (src/denoise.c:378-397) ----
  float *feat = g.features + (size_t)s * 68;
  float f_hi = 0;
  if (lane < RN_NB_BANDS) {
    f_hi = dct_lane(Exp, dctc, tb);
    Ly[lane] = (float)log10(1e-2 + (double)Ex[lane]);
  }
  __syncthreads();
  // log-energy follower + total energy: 32 serial steps, evaluated uniformly.  The reference forms
  // follow-1.5 in double and rounds the selected maximum to float (src/denoise.c:381-386); follow-1.5 is
  // exact in double, rounding is monotonic and the other operands are floats, so
  // (float)max(follow-1.5, b) == max(follow-1.5f, b): the whole recurrence stays in float, same bits.
  float E = 0;
  {
    float logMax = -2, follow = -2;
    for (int i = 0; i < RN_NB_BANDS; i++) {
      float ly = Ly[i];
      const float fd = follow - 1.5f;
      const float t = (fd > ly) ? fd : ly;
      const float lm7 = logMax - 7;
      ly = (lm7 > t) ? lm7 : t;
      logMax = (logMax > ly) ? logMax : ly;
      follow = (fd > ly) ? fd : ly;
      E += Ex[i];
      if (lane == 0) Ly[i] = ly;
    }
  }
  __syncthreads();
  // inference: silent frames zero the features and skip the network (src/denoise.c:389-393);
  // TRAINING build: features are always produced and "silence" means E < 0.1 (:389,:397)
  const int silence = TRAIN ? (((double)E < 0.1) ? 1 : 0) : (((double)E < 0.04) ? 1 : 0);
  const bool zero = !TRAIN && silence;
  if (lane < RN_NB_BANDS) {
    float f_lo = dct_lane(Ly, dctc, tb);
    if (lane == 0) f_lo -= 12;
    if (lane == 1) f_lo -= 4;
    feat[lane] = zero ? 0.f : f_lo;
    feat[RN_NB_BANDS + lane] = zero ? 0.f : f_hi;
    if (TRAIN) {
      tr.rec[(size_t)s * 98 + lane] = f_lo;
      tr.rec[(size_t)s * 98 + RN_NB_BANDS + lane] = f_hi;
    }
  }
  if (lane == 0) {
    const float fp = (float)(.01 * (double)(pitch_index - 300));
    feat[2 * RN_NB_BANDS] = zero ? 0.f : fp;
    if (TRAIN) tr.rec[(size_t)s * 98 + 2 * RN_NB_BANDS] = fp;
    g.silence[s] = silence;
  }
  if (TRAIN) {
    // rnn_frame_analysis of the CLEAN frame (src/dump_features.c:468) and the band-gain targets (:472-478)
    __syncthreads();
    float *cm = tr.clean_mem + (size_t)s * RN_FRAME_SIZE;
    const float *cx = tr.clean + (size_t)s * RN_FRAME_SIZE;
    for (int i = lane; i < RN_WINDOW_SIZE; i += WAVE) {
      float w = tb.half_window[i < RN_FRAME_SIZE ? i : RN_WINDOW_SIZE - 1 - i];
      float v = (i < RN_FRAME_SIZE ? cm[i] : cx[i - RN_FRAME_SIZE]) * w;
      F[bitrev960(i)] = {0.0010416667f * v, 0.0010416667f * 0.f};
    }
    fft960_lds(F, tw, lane);
    for (int i = lane; i < RN_FRAME_SIZE; i += WAVE) cm[i] = cx[i];
    for (int i = tr.lowpass[s] + lane; i < RN_FREQ_SIZE; i += WAVE) F[FPAD(i)] = {0.f, 0.f};
    __syncthreads();
    float *Ey = Ep;  // Ep already went to HBM
    band_accumulate<true, true>(Ey, F, F, Qs, sums, tb, lane);
    if (lane < RN_NB_BANDS) {
      float gt = (float)sqrt(((double)Ey[lane] + 1e-3) / ((double)Ex[lane] + 1e-3));
      if (gt > 1) gt = 1;
      if (silence || lane > tr.band_lp[s]) gt = -1;
      if ((double)Ey[lane] < 5e-2 && (double)Ex[lane] < 5e-2) gt = -1;
      const float vt = tr.vad[s];
      if (vt == 0 && tr.noise_free[s]) gt = -1;
      tr.rec[(size_t)s * 98 + RN_NB_FEATURES + lane] = gt;
      if (lane == 0) tr.rec[(size_t)s * 98 + 97] = vt;
    }
  }
  CLK_TAP(11);  // window + FFT(P) + Ep + Exp + features
#undef PB
}

extern "C" __global__ void __launch_bounds__(WAVE)
rn_analysis_kernel(RnGroupDev g, RnTablesDev tb, int slot, int parity) {
  analysis_body<false>(g, tb, slot, parity, RnTrainArgs{});
}
// The same kernel held to 80 VGPRs (a few registers spilled, ~1 % slower by itself).  While a 16-stream tile of
// the network kernel is resident on a CU (2 waves x 120 VGPRs per SIMD), only 272 VGPRs per SIMD are left: two
// waves of the 104-register build, three of this one -- 12 analysis waves beside the tile instead of 8, which is
// also what the LDS allows.  Measured (same box): +7.7 % at 4096 streams, +2.3 % at 8192, 0 at 6144, +1.5 % at 16,384,
// -2.5 % at <= 2048 (too few waves for it to matter), -0.6 % at 32,768 and 65,536 (no overlap left)
// => used from 3072 to 24575 streams.
extern "C" __global__ void __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(6, 6)))
rn_analysis_lean_kernel(RnGroupDev g, RnTablesDev tb, int slot, int parity) {
  analysis_body<false>(g, tb, slot, parity, RnTrainArgs{});
}

// TRAINING-mode variant: the inner loop of src/dump_features.c:466-491
extern "C" __global__ void __launch_bounds__(WAVE)
rn_train_features_kernel(RnGroupDev g, RnTablesDev tb, int slot, int parity, RnTrainArgs tr) {
  analysis_body<true>(g, tb, slot, parity, tr);
}


struct SynthLds {
  cpx F[RN_FFT_PADDED];  // inverse-FFT work area (padded layout); band products before that
  float misc[192];
};

// the second half of band_accumulate for callers that formed the 800 products themselves
__device__ void band_chain_finish(float *bandE, const float *Q, float *sums, int lane) {
  __syncthreads();
  {
    const int k = lane < RN_NB_BANDS + 2 ? lane : 0;
    const int lo = k ? c_eband[k - 1] : 0;
    const int len = lane < RN_NB_BANDS + 2 ? (k <= RN_NB_BANDS ? c_eband[k + 1] : 400) - lo : 0;
    const float *q = Q + lo + c_eband[k];
    float s = 0;
#pragma unroll 4
    for (int t = 0; t < 84; t++) {
      const float v = q[t];
      s += (t < len) ? v : 0.f;
    }
    if (lane < RN_NB_BANDS + 2) sums[lane] = s;
  }
  __syncthreads();
  if (lane < RN_NB_BANDS) {
    float v = sums[lane + 1];
    if (lane == 0) v = (sums[0] + sums[1]) * 2 / 3;
    if (lane == RN_NB_BANDS - 1) v = (sums[RN_NB_BANDS] + sums[RN_NB_BANDS + 1]) * 2 / 3;
    bandE[lane] = v;
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// K3: rnn_pitch_filter + gain smoothing/interpolation + frame_synthesis
// (src/denoise.c:474-496, 421-455, 140-154, 400-407, 200-217)
// Lane l owns bins l, l+64, ..., l+448 (and bin 480 for lane 32) in registers for the whole
// kernel; every HBM operand is requested before the first dependent instruction, so the wave
// pays one memory round trip instead of one per stage.  8.4 KB of LDS per wave.
// ---------------------------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(WAVE)
rn_synthesis_kernel(RnGroupDev g, RnTablesDev tb, float *__restrict__ out, int parity, int prev) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  SynthLds &L = *reinterpret_cast<SynthLds *>(smem_raw);
  const int s = blockIdx.x, lane = threadIdx.x;
  const cpx *tw = reinterpret_cast<const cpx *>(tb.twiddles);
  const float2 *dX = reinterpret_cast<const float2 *>(g.spec_X[prev] + (size_t)s * RN_SPEC_STRIDE);
  const float2 *dP = reinterpret_cast<const float2 *>(g.spec_P[prev] + (size_t)s * RN_SPEC_STRIDE);
  const float *dE = g.spec_E[prev] + (size_t)s * 96;
  const float *cE = g.spec_E[parity] + (size_t)s * 96;
  float *r = L.misc + 0, *gsm = L.misc + 32, *newE = L.misc + 64, *norm = L.misc + 96, *sums = L.misc + 128;
  float *Q = reinterpret_cast<float *>(L.F);
  const int silence = g.silence[s];
  constexpr int NBIN = 8;  // bins lane + 64*j; the 481st bin (480) is lane 32's j = 7

  // ---- every HBM operand up front ----
  float2 X[NBIN], P[NBIN];
  float frac[NBIN];
  int band[NBIN];
#pragma unroll
  for (int j = 0; j < NBIN; j++) {
    const int bin = lane + WAVE * j;
    const bool ok = bin < RN_FREQ_SIZE;
    X[j] = ok ? dX[bin] : make_float2(0.f, 0.f);
    P[j] = (ok && !silence) ? dP[bin] : make_float2(0.f, 0.f);
    band[j] = (bin < 400) ? tb.band_of_bin[bin] : 0;
    frac[j] = (bin < 400) ? tb.band_frac[bin] : 0.f;
  }
  float e_ex = 0, e_ep = 0, e_exp = 0, c_ex = 0, gi = 0, lastg = 0;
  if (lane < RN_NB_BANDS && !silence) {
    e_ex = dE[lane]; e_ep = dE[32 + lane]; e_exp = dE[64 + lane]; c_ex = cE[lane];
    gi = g.gains[(size_t)s * RN_NB_BANDS + lane];
    lastg = g.lastg[(size_t)s * RN_NB_BANDS + lane];
  }
  float *sm = g.synth_mem + (size_t)s * RN_FRAME_SIZE;
  float smv[8], wlo[8], whi[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int i = lane + WAVE * j;
    const bool ok = i < RN_FRAME_SIZE;
    smv[j] = ok ? sm[i] : 0.f;
    wlo[j] = ok ? tb.half_window[i] : 0.f;
    whi[j] = ok ? tb.half_window[RN_FRAME_SIZE - 1 - i] : 0.f;
  }

// src/denoise.c:140-154 per bin (bins >= 400 -> 0), from a 32-entry band vector in LDS
#define INTERP(vec, j)                                                                                      \
  ((lane + WAVE * (j)) >= 400 ? 0.f                                                                         \
   : band[j] == 0 ? (vec)[0]                                                                                \
   : band[j] == RN_NB_BANDS ? (vec)[RN_NB_BANDS - 1]                                                        \
                            : (1 - frac[j]) * (vec)[band[j] - 1] + frac[j] * (vec)[band[j]])

  if (!silence) {
    if (lane < RN_NB_BANDS) {  // src/denoise.c:429-440
      float rv;
      if (e_exp > gi) rv = 1;
      else rv = (float)((double)((e_exp * e_exp) * (1 - (gi * gi))) / (.001 + (double)((gi * gi) * (1 - (e_exp * e_exp)))));
      float t = (0 > rv) ? 0 : rv;
      t = (1 < t) ? 1 : t;
      rv = (float)sqrt((double)t);
      rv = (float)((double)rv * sqrt((double)e_ex / (1e-8 + (double)e_ep)));
      r[lane] = rv;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NBIN; j++) {  // :441-445, then the products of compute_band_energy (:446)
      const int bin = lane + WAVE * j;
      const float rf = INTERP(r, j);
      X[j].x += rf * P[j].x;
      X[j].y += rf * P[j].y;
      if (bin < 400) {
        float tmp = X[j].x * X[j].x;
        tmp += X[j].y * X[j].y;
        Q[c_eband[band[j] + 1] + bin] = frac[j] * tmp;
        Q[c_eband[band[j]] + bin] = (1 - frac[j]) * tmp;
      }
    }
    band_chain_finish(newE, Q, sums, lane);
    if (lane < RN_NB_BANDS) {
      norm[lane] = (float)sqrt((double)e_ex / (1e-8 + (double)newE[lane]));  // :447-449
      const float alpha = .6f;  // gain smoothing (src/denoise.c:479-487)
      gi = (gi > alpha * lastg) ? gi : alpha * lastg;
      double q = (double)gi * ((double)e_ex + 1e-3) / ((double)c_ex + 1e-3);
      g.lastg[(size_t)s * RN_NB_BANDS + lane] = (float)((1.f < q) ? 1.f : q);
      gsm[lane] = gi;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NBIN; j++) {  // :450-454 then :488-493
      const float nf = INTERP(norm, j), gf = INTERP(gsm, j);
      X[j].x *= nf;
      X[j].y *= nf;
      X[j].x *= gf;
      X[j].y *= gf;
    }
    __syncthreads();
  }
#undef INTERP
  // inverse_transform (src/denoise.c:200-217): Hermitian extension through the FORWARD FFT
#pragma unroll
  for (int j = 0; j < NBIN; j++) {
    const int bin = lane + WAVE * j;
    if (bin < RN_FREQ_SIZE) {
      L.F[bitrev960(bin)] = {0.0010416667f * X[j].x, 0.0010416667f * X[j].y};
      if (bin > 0 && bin < RN_FREQ_SIZE - 1)
        L.F[bitrev960(RN_WINDOW_SIZE - bin)] = {0.0010416667f * X[j].x, 0.0010416667f * (-X[j].y)};
    }
  }
  fft960_lds(L.F, tw, lane);
  // window + overlap-add (src/denoise.c:400-407)
  float *o = out + (size_t)s * RN_FRAME_SIZE;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int i = lane + WAVE * j;
    if (i < RN_FRAME_SIZE) {
      const int ilo = (RN_WINDOW_SIZE - i) % RN_WINDOW_SIZE, ihi = RN_FRAME_SIZE - i;
      float lo = (float)RN_WINDOW_SIZE * L.F[FPAD(ilo)].r;  // x[i]
      float hi = (float)RN_WINDOW_SIZE * L.F[FPAD(ihi)].r;  // x[480+i], window index 479-i
      lo *= wlo[j];
      hi *= whi[j];
      o[i] = lo + smv[j];
      sm[i] = hi;
    }
  }
}

// test tap: the log-energy expression of the feature stage (src/denoise.c:383) on arbitrary inputs, so that a sweep can
// measure how often ocml's log10 and the host libm's round a float differently (DESIGN.md section 2 "known residuals")
extern "C" __global__ void rn_log_energy_kernel(const float *__restrict__ ex, float *__restrict__ out, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    out[i] = (float)log10(1e-2 + (double)ex[i]);
}
extern "C" hipError_t rn_launch_log_energy(const float *ex, float *out, int n, hipStream_t st) {
  hipLaunchKernelGGL(rn_log_energy_kernel, dim3(1024), dim3(256), 0, st, ex, out, n);
  return hipGetLastError();
}

// host-visible launch helpers -----------------------------------------------------------------
// K0 and K1 are launched separately so that the host may put K0 of the next frame on a side stream
extern "C" hipError_t rn_launch_hp(const RnGroupDev *g, const float *in, int slot, hipStream_t st, hipEvent_t e0, hipEvent_t done) {
  RN_LAUNCH(rn_hp_kernel, dim3((g->n_streams + WAVE - 1) / WAVE), dim3(WAVE), 0, st, e0, done, *g, in, slot, 1);
  return hipGetLastError();
}
extern "C" hipError_t rn_launch_analysis(const RnGroupDev *g, const RnTablesDev *tb, int slot, int parity, hipStream_t st,
                                         hipEvent_t e0, hipEvent_t e1) {
  static const int force = [] { const char *e = getenv("RNNOISE_AMD_K1_LEAN"); return e ? atoi(e) : -1; }();  // 0 / 1: A/B runs
  const bool lean = force >= 0 ? force != 0 : (g->n_streams >= RN_K1_LEAN_MIN_STREAMS && g->n_streams < RN_K1_LEAN_MAX_STREAMS);
  if (lean)
    RN_LAUNCH(rn_analysis_lean_kernel, dim3(g->n_streams), dim3(WAVE), sizeof(AnalysisLds), st, e0, e1, *g, *tb, slot, parity);
  else
    RN_LAUNCH(rn_analysis_kernel, dim3(g->n_streams), dim3(WAVE), sizeof(AnalysisLds), st, e0, e1, *g, *tb, slot, parity);
  return hipGetLastError();
}
extern "C" hipError_t rn_launch_train_features(const RnGroupDev *g, const RnTablesDev *tb, const float *noisy, int slot,
                                               int parity, const RnTrainArgs *tr, hipStream_t st) {
  hipLaunchKernelGGL(rn_hp_kernel, dim3((g->n_streams + WAVE - 1) / WAVE), dim3(WAVE), 0, st, *g, noisy, slot, 0);
  hipLaunchKernelGGL(rn_train_features_kernel, dim3(g->n_streams), dim3(WAVE), sizeof(AnalysisLds), st, *g, *tb, slot,
                     parity, *tr);
  return hipGetLastError();
}
extern "C" hipError_t rn_launch_synthesis(const RnGroupDev *g, const RnTablesDev *tb, float *out, int cur, int prev,
                                          hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  RN_LAUNCH(rn_synthesis_kernel, dim3(g->n_streams), dim3(WAVE), sizeof(SynthLds), st, e0, e1, *g, *tb, out, cur, prev);
  return hipGetLastError();
}
