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
#include "fft_reg.h"
#include "log10_glibc.h"

#define WAVE 64
// from here on K1_SPW streams share a workgroup (see rn_analysis_single_kernel).  6,144 until round 6's last day; since the narrow
// phases and the follower are shared by the four streams of a workgroup (round 6) that form is ahead from 3,072 streams -- 23.3
// against 20.8 M frames/s there, 26.4 against 24.5 at 4,096 (one frame per call 0.201 against 0.231 ms), 27.8 against 26.0 at
// 5,120 -- and level at 2,048 (profiles/r6_late_ab.txt)
#define RN_K1_MULTI_MIN_STREAMS 2560

// Every LDS arena below belongs to ONE wavefront, and a wavefront's LDS instructions execute in issue order, so the
// hand-offs between lanes of a wave need no s_barrier: a wavefront-scope fence pins the compiler's ordering and nothing
// else.  (In a one-wave workgroup __syncthreads() compiles to exactly this; the analysis kernel runs several waves per
// workgroup and must not make them march in lock-step through every stage.)  Workgroup barriers are spelled
// __syncthreads() and appear only where one wave works on data of the others (analysis_body: "narrow phases").
#define RN_WSYNC()                                             \
  do {                                                         \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     \
    __builtin_amdgcn_wave_barrier();                           \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");     \
  } while (0)

__device__ __constant__ int c_eband[RN_NB_BANDS + 2] = {
    0,  2,  4,  6,  8,  10, 12, 15, 18,  21,  24,  28,  32,  36,  41,  47,  53,
    60, 68, 77, 87, 98, 110, 124, 140, 157, 176, 198, 223, 251, 282, 317, 356, 400};  // src/denoise.c:63-65

__device__ __constant__ int c_second_check[16] = {0, 0, 3, 2, 3, 2, 5, 2, 3, 2, 3, 2, 5, 2, 3, 2};  // src/pitch.c:422

// ---- LDS pointers with their address space spelled out.  The chains below take their operands through pointers picked per
// lane (this stream's arena or another's, the signal or its shifted copy); when such a pointer reaches a load as a generic
// one the compiler emits flat_load, which the LDS serves at a fraction of a ds_read's rate (round 2's doubling dots and
// fine-search chains ran on flat loads: 9-12 LDS cycles per instruction, profiles/r3_k1_sections_before.txt).
#define LDS_AS __attribute__((address_space(3)))
// ... and global tables.  A pointer that went through an empty asm (the "opaque copies" below) is a generic one to the compiler:
// loads through it become flat_load, which counts on lgkmcnt as well as vmcnt, so every later wait for an LDS result also
// waits for the table values.  The tables are global memory; saying so keeps them global_load.
#define GLOBAL_AS __attribute__((address_space(1)))
typedef const LDS_AS float *ldsf;
typedef float v4f_ __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
#define OPAQUE(v) asm("" : "+v"(v))
__device__ __forceinline__ ldsf to_lds(const float *p) { return (ldsf)p; }
__device__ __forceinline__ v4f_ lds_read16(ldsf p) { return *(const LDS_AS v4f_ *)p; }
// An 8-byte LDS read that stays ONE ds_read_b64 (2 LDS cycles per wave, 256 B/clk): left alone, the load-store optimiser
// fuses neighbouring ones into ds_read2_b64, which the LDS serves at half that rate (8 cycles per instruction,
// MI355X_MICROARCH.md section LDS) -- in the chains below that doubled the cycles of the operand stream.
__device__ __forceinline__ v2f lds_read8(ldsf p) { return *(const volatile LDS_AS v2f *)p; }
// lane K of each 16-lane row to every lane of the row: as an operand of an arithmetic instruction it folds into the instruction
// (v_..._dpp row_newbcast:K); the SOURCE lane must be active
template <int K>
__device__ __forceinline__ float row_bcast(float v) {  // lane K of each 16-lane row to every lane of the row
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + K, 0xF, 0xF, true));
}

// Band energy / correlation (src/denoise.c:90-138).  The reference's interleaved loop adds, for
// every bin of band b, (1-frac)*tmp to sum[b] and frac*tmp to sum[b+1]; so accumulator k receives
// band k-1's `frac` parts in bin order, then band k's `1-frac` parts.  Here all 64 lanes first form
// the 800 products (each rounded exactly as in the reference) and lay them out so that accumulator
// k's sequence is contiguous; then lane k < 34 adds its sequence in order.  Padding steps add +0.0f,
// which changes no bit (a sum that starts at +0 never becomes -0).
// ---- band energy / correlation from REGISTER-resident spectra (the transform of fft_reg.h leaves lane l with the bins
// 64*j + fft_pos(l)).  Same arithmetic and per-accumulator order as band_accumulate above; the products go to LDS in the
// layout of RnTablesDev::band_q -- accumulator k's terms contiguous from a 16-byte aligned slot -- so that the serial sums
// read four terms per LDS instruction.  NX / NY: array lengths (only bins < 400, j = 0..6, are used).
// NARR: number of product arrays formed side by side (1: <x, y>;  2: <y, y> into the first and <x, y> into the second, the Ep /
// Exp pair of src/denoise.c:373-375, which share every table load and the y operand).
template <int NARR, int NX, int NY>
__device__ __forceinline__ void band_products(float *Q, const float (&xr)[NX], const float (&xi)[NX], const float (&yr)[NY],
                                              const float (&yi)[NY], const RnTablesDev &tb, int pos, int lane) {
  // opaque copies: re-read from L1 at every call rather than kept across the pitch stage
  const GLOBAL_AS uint32_t *band_q = (const GLOBAL_AS uint32_t *)tb.band_q;
  const GLOBAL_AS float *band_frac = (const GLOBAL_AS float *)tb.band_frac;
  const GLOBAL_AS uint16_t *band_pad = (const GLOBAL_AS uint16_t *)tb.band_pad;
  asm volatile("" : "+s"(band_q), "+s"(band_frac), "+s"(band_pad));
  {  // the pad floats of every accumulator (RnTablesDev::band_pad): +0.0f
    const int ps = band_pad[lane];
    Q[ps] = 0.f;
    if (NARR == 2) Q[RN_BAND_QSTRIDE + ps] = 0.f;
  }
#pragma unroll
  for (int j = 0; j < 7; j++) {
    const int bin = WAVE * j + pos, bc = bin < 400 ? bin : 399;
    const uint32_t q = band_q[bc];
    const float frac = band_frac[bc];
    const float omf = 1 - frac;
    float tmp = (NARR == 2 ? yr[j] : xr[j]) * yr[j];
    tmp += (NARR == 2 ? yi[j] : xi[j]) * yi[j];
    float tmp2 = 0;
    if (NARR == 2) {
      tmp2 = xr[j] * yr[j];
      tmp2 += xi[j] * yi[j];
    }
    if (bin < 400) {
      Q[(q >> 11) & 0x7ff] = frac * tmp;
      Q[q & 0x7ff] = omf * tmp;
      if (NARR == 2) {
        Q[RN_BAND_QSTRIDE + ((q >> 11) & 0x7ff)] = frac * tmp2;
        Q[RN_BAND_QSTRIDE + (q & 0x7ff)] = omf * tmp2;
      }
    }
  }
}
// The 34 serial sums (src/denoise.c:104-112) of NARR product arrays as written by band_products, in ONE pass: lane k < 34 adds
// accumulator k of the first array; with two arrays, lane k + 30 (k = 4..33) adds accumulator k of the second at the same
// time and lanes 0..3, whose own chains are one slot long, take its first four accumulators afterwards.  A lane reads its
// chain 16 bytes at a time and leaves the loop when the chain ends (its length is rounded up to whole slots, the pad floats
// are +0.0f), so a term costs one add -- the chain of the widest accumulator (83 terms) sets the pass's length, the 33 shorter
// ones ride along.  sums: [40] per array.
template <int NARR>
__device__ __forceinline__ void band_sums(float *sums, const float *Q, const RnTablesDev &tb, int lane) {
  const bool second = NARR == 2 && lane >= RN_NB_BANDS + 2;
  const int k = second ? lane - 30 : (lane < RN_NB_BANDS + 2 ? lane : 0);
  const uint32_t ch = tb.band_chain[k];
  const int nt = (NARR == 1 && lane >= RN_NB_BANDS + 2) ? 0 : (int)(((ch >> 16) + 3) >> 2);
  ldsf q = to_lds(Q) + (ch & 0xffff) + (second ? RN_BAND_QSTRIDE : 0);
  RN_WSYNC();
  float s = 0;
#pragma unroll
  for (int t = 0; t < 21; t++) {  // longest accumulator: 39 + 44 = 83 terms = 21 slots
    if (t >= nt) break;
    const v4f_ v = lds_read16(q + 4 * t);
    s += v.x;
    s += v.y;
    s += v.z;
    s += v.w;
  }
  if (nt) sums[second ? 40 + k : k] = s;
  if (NARR == 2 && lane < 4) {  // accumulators 0..3 of the second array: 2, 4, 4, 4 terms (one slot each)
    const v4f_ v = lds_read16(q + RN_BAND_QSTRIDE);
    float s2 = 0;
    s2 += v.x;
    s2 += v.y;
    s2 += v.z;
    s2 += v.w;
    sums[40 + lane] = s2;
  }
  RN_WSYNC();
}
// band vector from the 34 sums (the first and the last band take two accumulators each, src/denoise.c:109-112)
__device__ __forceinline__ float band_of_sums(const float *sums, int lane) {
  float v = sums[lane + 1];
  if (lane == 0) v = (sums[0] + sums[1]) * 2 / 3;
  if (lane == RN_NB_BANDS - 1) v = (sums[RN_NB_BANDS] + sums[RN_NB_BANDS + 1]) * 2 / 3;
  return v;
}
// one band vector: sums + band_of_sums (Q as written by band_products<1>)
__device__ __forceinline__ void band_chain(float *bandE, const float *Q, float *sums, const RnTablesDev &tb, int lane) {
  band_sums<1>(sums, Q, tb, lane);
  if (lane < RN_NB_BANDS) bandE[lane] = band_of_sums(sums, lane);
  RN_WSYNC();
}

// src/denoise.c:160-170, lane i < 32 produces out[i]; c[j] = rnn_dct_table[j*32 + i], fetched by the
// caller well before use (the 32 loads are independent of the sum chain)
__device__ __forceinline__ float dct_lane(const float *in, const float *c, const RnTablesDev &tb) {
  float sum = 0;
#pragma unroll
  for (int j = 0; j < RN_NB_BANDS; j++) sum += in[j] * c[j];
  return (float)(sum * tb.dct_scale);
}

// max of two floats as ONE v_max_f32.  fmaxf() is llvm.maxnum, whose operands the compiler first canonicalises (v_max_f32 v, v, v)
// unless it can prove them quiet -- it cannot for values read from LDS -- although the instruction quiets a signalling NaN itself:
// three instructions where one gives the same bits for every operand pair.  Used on the follower's 32-step dependent chain.
__device__ __forceinline__ float max1(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// The log-energy follower of rnn_compute_frame_features (src/denoise.c:378-388) over ly[32] (log10 of the band energies, replaced in
// place by the followed values); returns the frame's energy sum E.  The reference forms follow-1.5 in double and rounds the selected
// maximum to float (src/denoise.c:381-386); follow-1.5 is exact in double, rounding is monotonic and the other operands are floats, so
// (float)max(follow-1.5, b) == max(follow-1.5f, b): the whole recurrence stays in float, same bits.  (The reference's MAX16 / MIN16
// ternaries as v_max_f32: the same value for every non-NaN operand pair -- at most the sign of a zero differs, when +0 meets -0, which
// neither log10 nor these differences produce -- and one instruction instead of a compare, a select and the VCC wait between them.)
__device__ __forceinline__ float log_follower(float *ly, const float *ex, bool store) {
  float logMax = -2, follow = -2, E = 0;
  for (int i = 0; i < RN_NB_BANDS; i++) {
    const float fd = follow - 1.5f;
    const float v = max1(logMax - 7, max1(fd, ly[i]));
    logMax = max1(logMax, v);
    follow = max1(fd, v);
    E += ex[i];
    if (store) ly[i] = v;
  }
  return E;
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
// reference) -- fbp_increments;  (2) the running energy Syy -- a float recurrence with a clamp, hence
// order-bound -- is swept once from the start value Syy0 = 1 + sum_{j<len} y[j]^2 (computed by the
// caller inside a dot-product pass), 4 steps per LDS transaction, leaving Syy-before-step-i in
// syy[i] -- fbp_sweep, ONE LANE PER STREAM (see the narrow phases of analysis_body);
// (3) best_pitch_select.  Used for the coarse (4x decimated) search; the fine search shares its sweep
// with yy_lookup (energy_sweeps below).  syy: scratch >= max_pitch rounded up to 4.
__device__ __forceinline__ void fbp_increments(const float *y, int len, int max_pitch, float *syy, int lane) {
  const int mp4 = (max_pitch + 3) & ~3;
  for (int i = lane; i < mp4; i += WAVE) {
    const float a = (i + len < len + max_pitch) ? y[i + len] : 0.f, b = y[i];
    syy[i] = a * a - b * b;
  }
}
// lane-private: every participating lane sweeps the syy[] of ITS stream (pointer and start value differ per lane)
__device__ __forceinline__ void fbp_sweep(float *syy, int max_pitch, float Syy0, bool store) {
  const int mp4 = (max_pitch + 3) & ~3;
  float Syy = Syy0;
  float4 d = *reinterpret_cast<const float4 *>(syy);
  for (int i = 0; i < mp4; i += 4) {
    // the next four increments are requested before this block's dependent adds: the LDS round trip is off the chain
    const float4 dn = *reinterpret_cast<const float4 *>(syy + (i + 4 < mp4 ? i + 4 : i));
    float4 o;
    o.x = Syy; Syy = fmaxf(1.f, Syy + d.x);  // MAX32(1, Syy): same value for every non-NaN Syy
    o.y = Syy; Syy = fmaxf(1.f, Syy + d.y);
    o.z = Syy; Syy = fmaxf(1.f, Syy + d.z);
    o.w = Syy; Syy = fmaxf(1.f, Syy + d.w);
    if (store) *reinterpret_cast<float4 *>(syy + i) = o;
    d = dn;
  }
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
__device__ __forceinline__ void energy_sweeps_prepare(const float *xlp, float *rsq, float *D, float *zero4, int lane) {
  for (int k = lane; k < 864; k += WAVE) {
    const float v = xlp[863 - k];
    rsq[k] = v * v;
  }
  for (int i = lane; i < 296; i += WAVE) {
    const float a = xlp[i + 480], b = xlp[i];
    D[i] = a * a - b * b;
  }
  if (lane < 4) zero4[lane] = 0.f;
}
// One lane per recurrence: role 0 = Syy of the fine search, role 1 = yy_lookup; the arrays are those of the lane's OWN
// stream (several streams' recurrences advance in one wave, see the narrow phases of analysis_body).
__device__ __forceinline__ void energy_sweeps_run(float *rsq, float *D, const float *zero4, float syy0, float xx, int role,
                                                  bool on) {
  if (on && role == 0) D[-1] = syy0;
  if (on && role == 1) rsq[479] = xx;
  float s = (role == 0) ? syy0 : xx;
  float *pa = (role == 0) ? D : rsq + 480;
  const float *pb = (role == 0) ? zero4 : rsq;
  const int sb = (role == 0) ? 0 : 4;
  const float lo = (role == 0) ? 1.f : -__builtin_inff();
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
    if (on && (role == 1 || j < 296)) *reinterpret_cast<float4 *>(pa + j) = o;
    a = an;
    b = bn;
  }
}
// The same two recurrences as functions of their own for workgroups whose narrow phases are spread over the waves
// (analysis_body): the fine Syy sweep and yy_lookup then advance at the same time on two SIMDs, and each sheds the operation
// the shared form carried for the other's sake -- (s + a) - 0 == s + a and max(-inf, x) == x, so the stored bits are those of
// energy_sweeps_run for every non-NaN operand.  One ROW of 16 lanes per stream: the row fetches 64 steps' operands in one
// 16-byte read per lane (lane k: steps 4k .. 4k+3) a whole block ahead, and every lane of the row runs the chain, taking step
// operands from lane k's registers as DPP operands (row_newbcast:k) -- the chain wave shares its CU's LDS pipe with fifteen
// waves in their wide phases, and with one small read per four steps the recurrence spent more time waiting for LDS round
// trips than adding (11 k cycles for 768 dependent instructions; profiles/r4_k1_narrow.txt).  All lanes of a row hold the same
// value and store it to the same address every four steps.
typedef LDS_AS float *ldsfw;
template <int K>
struct SyyRowSteps {
  static __device__ __forceinline__ void run(float &s, const v4f_ &a, ldsfw out, bool store) {
    v4f_ o;
    s = fmaxf(1.f, s + row_bcast<K>(a.x)); o.x = s;
    s = fmaxf(1.f, s + row_bcast<K>(a.y)); o.y = s;
    s = fmaxf(1.f, s + row_bcast<K>(a.z)); o.z = s;
    s = fmaxf(1.f, s + row_bcast<K>(a.w)); o.w = s;
    if (store) *(LDS_AS v4f_ *)(out + 4 * K) = o;
    SyyRowSteps<K + 1>::run(s, a, out, store);
  }
};
template <>
struct SyyRowSteps<16> {
  static __device__ __forceinline__ void run(float &, const v4f_ &, ldsfw, bool) {}
};
template <int K>
struct YyRowSteps {
  static __device__ __forceinline__ void run(float &s, const v4f_ &a, const v4f_ &b, ldsfw out, bool store) {
    v4f_ o;
    s = (s + row_bcast<K>(a.x)) - row_bcast<K>(b.x); o.x = s;
    s = (s + row_bcast<K>(a.y)) - row_bcast<K>(b.y); o.y = s;
    s = (s + row_bcast<K>(a.z)) - row_bcast<K>(b.z); o.z = s;
    s = (s + row_bcast<K>(a.w)) - row_bcast<K>(b.w); o.w = s;
    if (store) *(LDS_AS v4f_ *)(out + 4 * K) = o;
    YyRowSteps<K + 1>::run(s, a, b, out, store);
  }
};
template <>
struct YyRowSteps<16> {
  static __device__ __forceinline__ void run(float &, const v4f_ &, const v4f_ &, ldsfw, bool) {}
};
// the coarse running energy (fbp_sweep) the same way: syy[i] holds the increment of step i going in and Syy BEFORE step i coming
// out, i < 148; three blocks of 64 (the 44 steps past the end run on whatever the dead area behind holds and land there)
// (all 16 lanes of a row store the same values to the same address; storing from lane 0 only is not faster -- 0.830 against
//  0.827 ms, profiles/r4_k1_phases.txt -- same-address writes of an access group cost nothing extra)
template <int K>
struct SyyBeforeRowSteps {
  static __device__ __forceinline__ void run(float &s, const v4f_ &a, ldsfw out, bool store) {
    v4f_ o;
    o.x = s; s = fmaxf(1.f, s + row_bcast<K>(a.x));
    o.y = s; s = fmaxf(1.f, s + row_bcast<K>(a.y));
    o.z = s; s = fmaxf(1.f, s + row_bcast<K>(a.z));
    o.w = s; s = fmaxf(1.f, s + row_bcast<K>(a.w));
    if (store) *(LDS_AS v4f_ *)(out + 4 * K) = o;
    SyyBeforeRowSteps<K + 1>::run(s, a, out, store);
  }
};
template <>
struct SyyBeforeRowSteps<16> {
  static __device__ __forceinline__ void run(float &, const v4f_ &, ldsfw, bool) {}
};
__device__ __forceinline__ void fbp_sweep_row(float *syy, float Syy0, int l16) {
  float s = Syy0;
  ldsfw d = (ldsfw)syy;
  v4f_ a = *(const LDS_AS v4f_ *)(d + 4 * l16);
#pragma unroll 1
  for (int j = 0; j < 192; j += 64) {
    const v4f_ an = *(const LDS_AS v4f_ *)(d + (j + 64 < 192 ? j + 64 : j) + 4 * l16);
    SyyBeforeRowSteps<0>::run(s, a, d + j, true);
    a = an;
  }
}
// The two sweeps WITHOUT prepared operand arrays (workgroups of several streams): the operands of a block of 64 steps are
// squares of x_lp samples that the row's lanes hold anyway after one 16-byte read each, so lane k squares the samples of ITS
// four steps (each product rounded once, the difference of two products rounded once: energy_sweeps_prepare's values) and the
// chain takes them as DPP operands as above.  Nothing is staged in LDS: the pass that wrote 864 squares and 296 increments per
// stream is gone, x_lp's shifted copy survives from the fine search to the doubling dots, and each sweep can run beside
// another phase's chains instead of behind them (analysis_body).
__device__ __forceinline__ v4f_ sq_diff4(const v4f_ hi, const v4f_ lo) {
  v4f_ r;
  r.x = hi.x * hi.x - lo.x * lo.x;
  r.y = hi.y * hi.y - lo.y * lo.y;
  r.z = hi.z * hi.z - lo.z * lo.z;
  r.w = hi.w * hi.w - lo.w * lo.w;
  return r;
}
__device__ __forceinline__ v4f_ sq_rev4(const v4f_ v) {  // squares, last component first (a sweep that walks DOWN x_lp)
  v4f_ r;
  r.x = v.w * v.w;
  r.y = v.z * v.z;
  r.z = v.y * v.y;
  r.w = v.x * v.x;
  return r;
}
// Syy of the fine find_best_pitch: step i adds x_lp[i+480]^2 - x_lp[i]^2; D[-1..295] receives Syy after each step (Syy before lag i = D[i - 1])
__device__ __forceinline__ void sweep_syy_fine_row_x(const float *xlp, float *D, float syy0, int l16) {
  D[-1] = syy0;
  float s = syy0;
  ldsfw d = (ldsfw)D;
  ldsf x = to_lds(xlp) + 4 * l16;
  v4f_ a = sq_diff4(lds_read16(x + 480), lds_read16(x));
  v4f_ hi = lds_read16(x + 64 + 480), lo = lds_read16(x + 64);
#pragma unroll 1
  for (int j = 0; j < 320; j += 64) {  // 296 steps: four whole blocks and 40 steps of a fifth
    if (j < 256) {
      SyyRowSteps<0>::run(s, a, d + j, true);
    } else {  // steps 256 .. 295: ten of the sixteen groups
      v4f_ o;
#define SYY_G(K)                                              \
      s = fmaxf(1.f, s + row_bcast<K>(a.x)); o.x = s;         \
      s = fmaxf(1.f, s + row_bcast<K>(a.y)); o.y = s;         \
      s = fmaxf(1.f, s + row_bcast<K>(a.z)); o.z = s;         \
      s = fmaxf(1.f, s + row_bcast<K>(a.w)); o.w = s;         \
      *(LDS_AS v4f_ *)(d + j + 4 * K) = o;
      SYY_G(0) SYY_G(1) SYY_G(2) SYY_G(3) SYY_G(4) SYY_G(5) SYY_G(6) SYY_G(7) SYY_G(8) SYY_G(9)
#undef SYY_G
    }
    a = sq_diff4(hi, lo);  // the next block's increments from the samples requested a block ago; then the block after that
    const int jn = j + 128 < 320 ? j + 128 : 256;
    hi = lds_read16(x + jn + 480);
    lo = lds_read16(x + jn);
  }
}
// yy_lookup of remove_doubling: step m = 0..383 is yy = (yy + x_lp[383-m]^2) - x_lp[863-m]^2, the value after it is yy_lookup[m+1];
// yyl1 = &yy_lookup[1] (16-byte aligned), yy_lookup[0] = xx.  The values are stored as they are: the reference's MAX32(0, yy)
// (src/pitch.c:452-455) is applied by the reader.
__device__ __forceinline__ void sweep_yy_lookup_row_x(const float *xlp, float *yyl1, float xx, int l16) {
  yyl1[-1] = xx;
  float s = xx;
  ldsfw out = (ldsfw)yyl1;
  ldsf xa = to_lds(xlp) + 380 - 4 * l16, xb = to_lds(xlp) + 860 - 4 * l16;  // the samples of steps 4k .. 4k+3, last step first
  v4f_ a = sq_rev4(lds_read16(xa)), b = sq_rev4(lds_read16(xb));
  v4f_ ra = lds_read16(xa - 64), rb = lds_read16(xb - 64);
#pragma unroll 1
  for (int j = 0; j < 384; j += 64) {
    YyRowSteps<0>::run(s, a, b, out + j, true);
    a = sq_rev4(ra);
    b = sq_rev4(rb);
    const int jn = j + 128 < 384 ? j + 128 : 320;
    ra = lds_read16(xa - jn);
    rb = lds_read16(xb - jn);
  }
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

// dot-product chain (src/pitch.h:51-142: one serial `sum = sum + x*y` per lag), n a multiple of 8,
// x 16-byte aligned, y arbitrary; both may differ per lane.  The next 8 operand pairs are fetched
// from LDS while the current 8 are being added, so the LDS round trip is off the chain.
// s0 is the chain's start value: a lane with x == y and s0 = 1 computes a find_best_pitch start
// energy 1 + sum y[j]^2 (src/pitch.c:56-61) in the same instructions as the real dot products.
__device__ __forceinline__ float chain_dot8(ldsf x, ldsf y, int n, float s0 = 0.f) {
  float s = s0;
  v4f_ xa = lds_read16(x), xb = lds_read16(x + 4);
  float ya[8];
#pragma unroll
  for (int k = 0; k < 8; k++) ya[k] = y[k];
#pragma unroll 2
  for (int i = 0; i < n; i += 8) {
    const int nx = (i + 8 < n) ? i + 8 : i;
    const v4f_ xc = lds_read16(x + nx), xd = lds_read16(x + nx + 4);
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

// chain_dot8 with x == y (a start energy s0 + sum y[j]^2, src/pitch.c:56-61): one operand stream instead of two, y 16-byte aligned
__device__ __forceinline__ float chain_sq8(ldsf y, int n, float s0) {
  float s = s0;
  v4f_ a = lds_read16(y), b = lds_read16(y + 4);
#pragma unroll 2
  for (int i = 0; i < n; i += 8) {
    const int nx = (i + 8 < n) ? i + 8 : i;
    const v4f_ an = lds_read16(y + nx), bn = lds_read16(y + nx + 4);
    float p0 = a.x * a.x, p1 = a.y * a.y, p2 = a.z * a.z, p3 = a.w * a.w;
    float p4 = b.x * b.x, p5 = b.y * b.y, p6 = b.z * b.z, p7 = b.w * b.w;
    OPAQUE(p0); OPAQUE(p1); OPAQUE(p2); OPAQUE(p3); OPAQUE(p4); OPAQUE(p5); OPAQUE(p6); OPAQUE(p7);
    s = s + p0;
    s = s + p1;
    s = s + p2;
    s = s + p3;
    s = s + p4;
    s = s + p5;
    s = s + p6;
    s = s + p7;
    a = an;
    b = bn;
  }
  return s;
}

// chain_sq8 for a pass in which every lane of the wave takes part, one ROW of 16 lanes per chain: lane k of the row squares
// y[i + k] (one 4-byte read and one multiply per 16 steps) and step k adds lane k's square as a DPP operand
// (v_add_f32_dpp row_newbcast:k) -- 18 instructions per 16 steps instead of 36, on a wave that runs alone between two
// barriers and issues one instruction per ~5 cycles whatever it is.  Same squares, same order of adds; every lane of the
// row ends with the chain's value.  n a multiple of 16.
template <int K>
struct SqRowSteps {
  static __device__ __forceinline__ void run(float &s, float q) {
    s = s + row_bcast<K>(q);
    SqRowSteps<K + 1>::run(s, q);
  }
};
template <>
struct SqRowSteps<16> {
  static __device__ __forceinline__ void run(float &, float) {}
};
__device__ __forceinline__ float chain_sq_row(ldsf y, int n, float s0, int l16) {
  float s = s0;
  ldsf yl = y + l16;
  float v = yl[0], vn = yl[16];
#pragma unroll 1
  for (int i = 0; i < n; i += 16) {
    float q = v * v;
    OPAQUE(q);
    v = vn;
    vn = yl[i + 32 < n ? i + 32 : i];  // two blocks ahead (the block is 16 dependent adds long)
    SqRowSteps<0>::run(s, q);
  }
  return s;
}

// One lag of the 5-lag autocorrelation of rnn_pitch_downsample (src/celt_lpc.c:92-174) as a row chain: 860 terms
// x[i] * x[i + lag] in order (rnn_pitch_xcorr over fastN), then the tail chain of the terms i = 860 .. 863 - lag, then their
// sum -- what lane `lag` of rn_hp_one_kernel computes, 19 instructions per 16 terms instead of 64.  Excluded tail terms enter
// as +0.0f (a sum that starts at +0 never becomes -0: adding +0 changes no bit).  Every lane of the row returns the value.
template <int K, int N>
struct AcRowSteps {
  static __device__ __forceinline__ void run(float &s, float p) {
    s = s + row_bcast<K>(p);
    AcRowSteps<K + 1, N>::run(s, p);
  }
};
template <int N>
struct AcRowSteps<N, N> {
  static __device__ __forceinline__ void run(float &, float) {}
};
__device__ __forceinline__ float autocorr_row(ldsf x, int lag, int l16) {
  float s = 0.f;
  ldsf xa = x + l16, xb = x + lag + l16;
  float a = xa[0], b = xb[0], an = xa[16], bn = xb[16];
#pragma unroll 1
  for (int i = 0; i < 848; i += 16) {  // 53 whole blocks
    float p = a * b;
    OPAQUE(p);
    a = an;
    b = bn;
    an = xa[i + 32];  // (two blocks ahead; past the signal's end the values are read and never added)
    bn = xb[i + 32];
    AcRowSteps<0, 16>::run(s, p);
  }
  {  // terms 848 .. 859
    float p = a * b;
    OPAQUE(p);
    AcRowSteps<0, 12>::run(s, p);
  }
  float d = 0.f;
  {
    const int i = 860 + (l16 & 3);
    float p = (l16 < 4 && i + lag <= 863) ? x[i + lag] * x[i] : 0.f;
    OPAQUE(p);
    AcRowSteps<0, 4>::run(d, p);
  }
  return s + d;
}

// chain_dot8 with the y operand fetched two steps per LDS instruction: y2 = 8-byte aligned address of {y[0], y[1]}.
// For an arbitrary (odd) start the caller points y2 into a copy of the signal shifted by one sample (see the doubling
// dots): half the LDS instructions, and the per-lane-offset reads collide on 32 eight-byte slots instead of 32 banks.
__device__ __forceinline__ float chain_dot8_y2(ldsf x, ldsf y2, int n, float s0 = 0.f) {
  float s = s0;
  v4f_ xa = lds_read16(x), xb = lds_read16(x + 4);
  v2f ya[4];
#pragma unroll
  for (int k = 0; k < 4; k++) ya[k] = lds_read8(y2 + 2 * k);
#pragma unroll 2
  for (int i = 0; i < n; i += 8) {
    const int nx = (i + 8 < n) ? i + 8 : i;
    const v4f_ xc = lds_read16(x + nx), xd = lds_read16(x + nx + 4);
    v2f yn[4];
#pragma unroll
    for (int k = 0; k < 4; k++) yn[k] = lds_read8(y2 + nx + 2 * k);
    float p0 = xa.x * ya[0].x, p1 = xa.y * ya[0].y, p2 = xa.z * ya[1].x, p3 = xa.w * ya[1].y;
    float p4 = xb.x * ya[2].x, p5 = xb.y * ya[2].y, p6 = xb.z * ya[3].x, p7 = xb.w * ya[3].y;
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
    for (int k = 0; k < 4; k++) ya[k] = yn[k];
  }
  return s;
}

// chain_dot8_y2 for a pass in which EVERY lane of the wave takes part and all chains share x (the doubling dots): the x
// operand then does not come from LDS once per lane -- a ds_read_b128 of the same 16 bytes by 64 lanes costs 4 LDS cycles, as
// much as the whole y operand, and these passes are bound by the LDS pipe -- but once per wave: each row of 16 lanes holds
// x[i .. i+15] in one register (lane l: x[i + (l & 15)], one 4-byte read per 16 steps) and step k takes lane k of the row as a
// DPP operand of the multiply (v_mul_f32_dpp row_newbcast:k).  Same products, same order of adds: the bits are chain_dot8_y2's.
// A DPP operand reads the register of another LANE, which must be active: callers run this with all 64 lanes (lanes without a
// chain of their own compute a dummy one).  n a multiple of 16.
template <int K>
struct ChainSteps {
  static __device__ __forceinline__ void run(float &s, float xr, const v2f (&y)[8]) {
    float p = row_bcast<K>(xr) * ((K & 1) ? y[K >> 1].y : y[K >> 1].x);
    OPAQUE(p);
    s = s + p;
    ChainSteps<K + 1>::run(s, xr, y);
  }
};
template <>
struct ChainSteps<16> {
  static __device__ __forceinline__ void run(float &, float, const v2f (&)[8]) {}
};
// DEEP: operands TWO blocks of 16 steps ahead instead of one (16 more registers).  The narrow phase's chain wave shares its CU's
// LDS pipe with fifteen waves in their wide phases: one block ahead (~170 cycles of chain) does not cover an LDS round trip
// there, and the chain waited for operands at every block (9 k of its 11.5 k cycles).
template <bool DEEP>
__device__ __forceinline__ float chain_dot16_xrow(ldsf x, ldsf y2, int n, int lane) {
  float s = 0.f;
  ldsf xl = x + (lane & 15);
  if (!DEEP) {
    float xr = *xl;
    v2f ya[8];
#pragma unroll
    for (int k = 0; k < 8; k++) ya[k] = lds_read8(y2 + 2 * k);
    for (int i = 0; i < n; i += 16) {
      const int nx = (i + 16 < n) ? i + 16 : i;
      const float xn = xl[nx];
      v2f yn[8];
#pragma unroll
      for (int k = 0; k < 8; k++) yn[k] = lds_read8(y2 + nx + 2 * k);
      ChainSteps<0>::run(s, xr, ya);
      xr = xn;
#pragma unroll
      for (int k = 0; k < 8; k++) ya[k] = yn[k];
    }
    return s;
  }
  // two blocks ahead: three operand sets in rotation, the loop body spelled out three blocks at a time so that no set is
  // ever copied into another (n a multiple of 48)
  float x0 = xl[0], x1 = xl[16], x2;
  v2f y0[8], y1[8], y2v[8];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    y0[k] = lds_read8(y2 + 2 * k);
    y1[k] = lds_read8(y2 + 16 + 2 * k);
  }
  for (int i = 0; i < n; i += 48) {
    const int a = (i + 32 < n) ? i + 32 : i, b = (i + 48 < n) ? i + 48 : i, c = (i + 64 < n) ? i + 64 : i;
    x2 = xl[a];
#pragma unroll
    for (int k = 0; k < 8; k++) y2v[k] = lds_read8(y2 + a + 2 * k);
    ChainSteps<0>::run(s, x0, y0);
    x0 = xl[b];
#pragma unroll
    for (int k = 0; k < 8; k++) y0[k] = lds_read8(y2 + b + 2 * k);
    ChainSteps<0>::run(s, x1, y1);
    x1 = xl[c];
#pragma unroll
    for (int k = 0; k < 8; k++) y1[k] = lds_read8(y2 + c + 2 * k);
    ChainSteps<0>::run(s, x2, y2v);
  }
  return s;
}

// One 10 KB LDS arena per wave (16 waves = one full round per CU at 4096 streams), time-shared
// (float offsets, the SCR_* constants below):
//   FFT phases   : F = [0,2160) (960 complex, padded layout); the band products Q live in [1084,1948),
//                  above the 481 bins that matter; small per-frame vectors in [2392,2560)
//   coarse search: xlp [0,864) | y4 [864,1296) | y4 shifted by one [1344,1730) | interleaved pairs Z [1732,2310)
//                  during the chains, then running energies [1728,1876) and xcorr [2028,2175)
//   fine search  : xlp | x_lp shifted by one sample [864,1727) -- it stays until the doubling dots are through -- | Syy
//                  [1731,2028) | xcorr [2028,2324); then yy_lookup [1731,2116) over the dead Syy / xcorr, doubling dots [2120,2184)
//                  (one-stream workgroups, which run their chains themselves: reversed squares -> yy_lookup [864,1728) and
//                  energy increments -> Syy [1731,2028) staged first, 4 zeros [2324,2328), the shifted copy made afterwards)
// Three chains per lane against the same x: two lags as a 2-wide vector chain -- it compiles to v_pk_mul_f32 /
// v_pk_add_f32, each component still mul-then-add in the reference order; z[k] = {y[k], y[k+49]} comes from an interleaved
// copy, so a pair is one 8-byte LDS read that lands in an aligned register pair -- plus a scalar chain whose y operand comes
// two steps per LDS instruction from y2 (8-byte aligned; the caller points odd offsets into a copy shifted by one sample).
// One pass of this over 49 lanes x 3 lags (+ one lane for the start energy) replaces the 64 x 2 pass and the 20-lane
// second pass of the coarse search.
struct XC3 { v2f p; float q; };
__device__ __forceinline__ XC3 chain_dot8_x3(ldsf x, ldsf z, ldsf y2, int n, v2f s0) {
  v2f s = s0;
  float q = 0.f;
  v4f_ xa = lds_read16(x), xb = lds_read16(x + 4);
  v2f y[8], w[4];
#pragma unroll
  for (int k = 0; k < 8; k++) y[k] = lds_read8(z + 2 * k);
#pragma unroll
  for (int k = 0; k < 4; k++) w[k] = lds_read8(y2 + 2 * k);
#pragma unroll 2
  for (int i = 0; i < n; i += 8) {
    const int nx = (i + 8 < n) ? i + 8 : i;
    const v4f_ xc = lds_read16(x + nx), xd = lds_read16(x + nx + 4);
    v2f yn[8], wn[4];
#pragma unroll
    for (int k = 0; k < 8; k++) yn[k] = lds_read8(z + 2 * (nx + k));
#pragma unroll
    for (int k = 0; k < 4; k++) wn[k] = lds_read8(y2 + nx + 2 * k);
    float p0 = xa.x * w[0].x, p1 = xa.y * w[0].y, p2 = xa.z * w[1].x, p3 = xa.w * w[1].y;
    float p4 = xb.x * w[2].x, p5 = xb.y * w[2].y, p6 = xb.z * w[3].x, p7 = xb.w * w[3].y;
    OPAQUE(p0); OPAQUE(p1); OPAQUE(p2); OPAQUE(p3); OPAQUE(p4); OPAQUE(p5); OPAQUE(p6); OPAQUE(p7);
    // (x[1], x[3], ... sit in the HIGH register of their pair after the 16-byte read: broadcast into a packed multiply they
    //  would be an op_sel operand, and packed-FP32 instructions with an op_sel bit are not allowed in this library --
    //  profiles/r5_gru_race.txt, tests/test_kernel_budgets_cpu.py -- so their two products are scalar multiplies)
#define X3_LO(xv, yv) s = s + v2f{xv, xv} * (yv)
#define X3_HI(xv, yv) do { float a_ = (xv) * (yv).x, b_ = (xv) * (yv).y; OPAQUE(a_); OPAQUE(b_); s = s + v2f{a_, b_}; } while (0)
    X3_LO(xa.x, y[0]);
    q = q + p0;
    X3_HI(xa.y, y[1]);
    q = q + p1;
    X3_LO(xa.z, y[2]);
    q = q + p2;
    X3_HI(xa.w, y[3]);
    q = q + p3;
    X3_LO(xb.x, y[4]);
    q = q + p4;
    X3_HI(xb.y, y[5]);
    q = q + p5;
    X3_LO(xb.z, y[6]);
    q = q + p6;
    X3_HI(xb.w, y[7]);
    q = q + p7;
#undef X3_LO
#undef X3_HI
    xa = xc;
    xb = xd;
#pragma unroll
    for (int k = 0; k < 8; k++) y[k] = yn[k];
#pragma unroll
    for (int k = 0; k < 4; k++) w[k] = wn[k];
  }
  return XC3{s, q};
}

struct AnalysisLds {
  // 9,504 B: three of these workgroups (4 arenas each) leave 49 KB of a CU's LDS to a network-kernel tile.  2376 = 8 (mod 32):
  // the arenas of a workgroup start 8 banks apart, so when wave 0 walks the same index of all four streams in a narrow phase
  // the four accesses fall on different banks (with a multiple of 32 they were 4-way conflicts)
  float a[2376];
};
#define SCR_XLP 0
#define SCR_SQ 864    // [864]  one-stream workgroups: reversed squares of xlp, later yy_lookup
#define SCR_Y4 864    // [432]  4x-decimated signal (coarse search only; over the not yet written squares)
#define SCR_Y4S 1344  // [386]  the same shifted by one sample (8-byte reads at odd offsets); 480 floats after y4: the two copies'
                      //        8-byte slots interleave, so even and odd lanes of one read do not collide
#define SCR_Z 1732    // [578]  {y4[k], y4[k+49]} pairs for the packed coarse chains
#define SCR_SYY 1728  // [148]  running energies of the coarse find_best_pitch
#define SCR_D 1732    // [-1..295] fine search: Syy increments, then Syy itself (16-byte aligned)
#define SCR_XC 2028   // [296]  xcorr[] of pitch_search
#define SCR_ZERO 2324 // [4]
#define SCR_DOTS 2120 // [64]  doubling dots (behind yy_lookup, over the dead fine xcorr)
#define SCR_YYL 1732  // &yy_lookup[1] (16-byte aligned; yy_lookup[0] at 1731), [385] over the dead Syy / fine xcorr areas
#define SCR_XS 864    // [863] x_lp shifted by one sample, for the 8-byte reads of the fine-search chains and the doubling dots
#define SCR_Q 0       // [2 x RN_BAND_QSTRIDE] band products in the layout of RnTablesDev::band_q: over the staged window, which
                      //        is dead once the transform has its inputs; two arrays for the Ep / Exp pair
#define SCR_EX 2336   // [32]  band energies of X: the one vector that lives from the first transform to the features
#define SCR_MISC 2092 // sums[2][40] | Ep[32] | Exp[32] | Ly[32]: transform phases only (behind the band products)

// ---------------------------------------------------------------------------------------------
// K1: rnn_compute_frame_features (src/denoise.c:347-398) on the high-passed frame that K0 put
// into the pitch ring.  One wavefront owns one stream-frame and a 10 KB LDS arena; SPW of them form
// a workgroup (grid = ceil(n_streams / SPW)).
// `ring0` = physical ring position of pitch_buf[0] (src/denoise.c:359-360 shift = ring rotation).
//
// Narrow phases.  Stretches of the pitch analysis are serial chains that occupy 1, 12, 2 and 31 lanes of a
// wave for 148, 480, 384 and 480 steps (coarse running energy; fine cross-correlations + start
// energies; fine running energy + yy_lookup; the candidate dots of remove_doubling) -- a third of the
// kernel's instructions for a few percent of its arithmetic.  A one-lane instruction costs the issue slot
// of a 64-lane one, so the workgroup's waves meet at a barrier and ONE wave runs such a chain for all SPW
// streams side by side (disjoint lane groups, each pointing into its own stream's arena; two waves with two
// streams each for the 31-lane pass) while the others wait without issuing anything.  Every chain is still
// one lane's serial sum in the reference order.  Between two barriers several of these phases run on
// different waves at once, whenever their operands allow it:
//   barrier | coarse running energy (nw1)                                                | barrier
//   barrier | fine cross-correlations + <x, x> (nw2)  ||  start energy, then fine Syy (nw3a) | barrier
//   barrier | candidate dots (nw1: streams 0, 1; nw3b: streams 2, 3)  ||  yy_lookup (nw2)    | barrier
// (round 4, first form: the running energies ran in a phase of their own behind the fine chains, from squares and
//  increments that every wave staged in LDS first, and each wave ran its own 31 candidate dots: 0.900 -> 0.854 ms at
//  65,536 streams, profiles/r4_k1_phases.txt)
// ---------------------------------------------------------------------------------------------
#define K1_SPW 4       // streams (= waves) per workgroup of the inference kernels: 12 fine-search lanes x 4 <= 64
#define SCR_MAIL 2328  // [8] mailbox of the stream inside its own arena: values handed between a stream's wave and wave 0
#define MAIL_SYY0C 0   //   start energy of the coarse find_best_pitch
#define MAIL_BP0 1     //   coarse best lags (int bits)
#define MAIL_BP1 2
#define MAIL_XX 3      //   <x, x> of remove_doubling
#define MAIL_SYY0F 4   //   start energy of the fine find_best_pitch
#define MAIL_T0 5      //   remove_doubling's T0 (int bits), for the wave that runs this stream's candidate dots
#define MAIL_E 6       //   the frame's band-energy sum (behind `silence`), from the wave that ran the workgroup's log-energy followers
// K1_STOP(k): instrumented build only -- the whole workgroup leaves the kernel at stop point k (tools/k1_prefix.sh runs the
// kernel once per stop point under the PMC counters: the differences are each section's LDS cycles, bank conflicts, VALU
// instructions and time).  Stop points sit where all waves of a workgroup pass together.
#if RN_INSTRUMENT
#define K1_STOP(k) do { if (k1_stop == (k)) return; } while (0)
#elif defined(RN_K1_MARKS)  // tools/asm_sections.py: section boundaries as comments in the assembly (analysis builds only)
#define K1_STOP(k) asm volatile("; K1MARK %0" ::"n"(k))
#else
#define K1_STOP(k) do { } while (0)
#endif
template <bool TRAIN, int SPW>
__device__ __forceinline__ void analysis_body(const RnGroupDev &g, const RnTablesDev &tb, int slot_arg, int parity,
                                              const RnTrainArgs &tr, int listed_row = -1) {
  const int slot = slot_arg & 255;
  const int k1_stop = RN_INSTRUMENT ? ((slot_arg >> 16) & 31) : 0;
  (void)k1_stop;
  // Which wave of the workgroup runs which narrow phase (see below).  Bit 9 of the slot argument: the four phases go to four
  // DIFFERENT waves, rotated from workgroup to workgroup by a hash of the block number -- the extra work is then spread over
  // the waves (and so over the SIMDs: a workgroup's waves sit on different SIMDs) instead of making wave 0 the straggler of
  // every workgroup, and the two running-energy sweeps of phase 3 advance side by side.  Clear: everything on wave 0 (round 3).
  const bool spread = SPW > 1;                            // (round 4 measured it against everything-on-wave-0 in one kernel: 4 %; that
                                                          //  form now lives only in the one-stream workgroups, where wave 0 is the stream)
  // (A/B switches of the instrumented build, rn_launch_analysis: compile-time constants in the product)
  const bool xrow = !(RN_INSTRUMENT && (slot_arg & 1024));              // bit 10: the doubling dots read x per lane from LDS (chain_dot8_y2)
  const int narrow_prio = (RN_INSTRUMENT && (slot_arg & 4096)) ? 1 : 3; // bit 12: the narrow-phase waves keep the kernel's priority
  const bool fine_deep = !(RN_INSTRUMENT && (slot_arg & 16384));        // bit 14: the fine-search chains fetch one block ahead, not two
  // bit 13: TIMING ONLY, WRONG RESULTS -- the 31 candidate chains of remove_doubling read at offsets 2 l (one bank pair per lane: no bank
  // conflict is possible): what the kernel would take if the conflicts of that pass were gone (profiles/r6_k1_dots_conflicts.txt)
  const bool dots_noconf = RN_INSTRUMENT && (slot_arg & 8192);
  const int nw0 = spread ? (int)((blockIdx.x * 0x9E3779B1u) >> 30) : 0;
  const int nw1 = nw0, nw2 = spread ? (nw0 + 1) & 3 : 0, nw3a = spread ? (nw0 + 2) & 3 : 0, nw3b = spread ? (nw0 + 3) & 3 : 0;
  const int ring0 = RN_RING0(slot);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  AnalysisLds *arenas = reinterpret_cast<AnalysisLds *>(smem_raw);
  const int wave = SPW > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) : 0, lane = threadIdx.x & (WAVE - 1);
  AnalysisLds &L = arenas[wave];
  float *mail = L.a + SCR_MAIL;
  // a tail workgroup's surplus waves redo the last stream without storing anything: they still meet every barrier
  // (a listed row: rn_dev.h RnRows.  In a workgroup of several waves ALL of them then work on that one row and only wave 0
  //  stores: the others are there for the narrow phases -- see rn_analysis_rows_kernel)
  const int s_raw = listed_row >= 0 ? listed_row : (int)blockIdx.x * SPW + wave;
  const bool in_range = s_raw < g.n_streams;
  const bool wr = in_range && (listed_row < 0 || wave == 0);
  const int s = in_range ? s_raw : g.n_streams - 1;
  // ... and only wave 0 does the frame's wide work at all: the other waves go from barrier to barrier and serve the narrow
  // phases, every row of which then points at wave 0's arena (`solo`; the rows repeat one chain on the same addresses)
  const bool solo = SPW > 1 && listed_row >= 0;
  const bool active = !solo || wave == 0;
#define ARENA(i) arenas[solo ? 0 : (i)]
// workgroup barrier between a stream's own wave and wave 0 (a wavefront fence when the workgroup is one wave)
#define WG_SYNC()                      \
  do {                                 \
    if (SPW > 1) __syncthreads();      \
    else RN_WSYNC();                   \
  } while (0)
  float *scr = L.a;
  float *xlp = scr + SCR_XLP, *Qs = scr + SCR_Q;
  float *sums = scr + SCR_MISC, *Ex = scr + SCR_EX, *Ep = sums + 80, *Exp = Ep + 32, *Ly = Exp + 32;
  static_assert(SCR_Q + 2 * RN_BAND_QSTRIDE <= SCR_MISC && SCR_MISC + 80 + 96 <= SCR_MAIL, "transform-phase LDS map");
#if RN_INSTRUMENT
  float *dbg = (g.debug && wr) ? g.debug + (size_t)s * RN_DBG_FLOATS : nullptr;
  unsigned long long clk_prev = dbg ? __builtin_amdgcn_s_memtime() : 0;
#else  // product build: every `if (dbg)` below folds away
  float *const dbg = nullptr;
  unsigned long long clk_prev = 0;
  (void)clk_prev;
#endif
  const float *ring = g.pitch_ring + (size_t)s * RN_RING_SIZE;

  CLK_TAP(0);
  CLK_TAP(1);
  const int pos = fft_pos(lane);              // this lane's bins after a transform: 64*j + pos
  const float2 *ftw = reinterpret_cast<const float2 *>(tb.fft_tw);
  float *S = scr;                             // [960] windowed frame in natural order (FFT phases only)
  // window [start, start+960) of pitch_buf -> this lane's 15 consecutive scaled samples (fft_reg.h "Input"), through LDS:
  // the global loads stay coalesced, and the 15-float runs are read back conflict-free (stride 15 is odd)
  // (`start` wave-uniform.  All 30 loads are in flight together; the ring position wraps at most once inside a window, which
  // an unsigned minimum resolves -- a - SIZE is huge when a < SIZE -- and which half of the window a sample is in is known
  // at compile time for every t but one)
  auto window_to_regs = [&](float (&ar)[15], float (&ai)[15], int start, const float *hw_) {
    const GLOBAL_AS float *hw = (const GLOBAL_AS float *)hw_;
    int p0 = ring0 + start;
    p0 = (p0 >= RN_RING_SIZE) ? p0 - RN_RING_SIZE : p0;
    // (byte offsets in 32 bits: the load then takes the wave-uniform ring base as a scalar pair + one offset register, instead of a
    //  64-bit address formed per load)
    const unsigned pl = 4u * ((unsigned)p0 + (unsigned)lane);
    float v[RN_WINDOW_SIZE / WAVE], w[RN_WINDOW_SIZE / WAVE];
#pragma unroll
    for (int t = 0; t < RN_WINDOW_SIZE / WAVE; t++) {
      const unsigned a = pl + 4u * WAVE * t;
      v[t] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(ring) + min(a, a - 4u * (unsigned)RN_RING_SIZE));
      const unsigned i = (unsigned)lane + WAVE * t;
      w[t] = hw[WAVE * t + WAVE <= RN_FRAME_SIZE ? i : (WAVE * t >= RN_FRAME_SIZE || i >= RN_FRAME_SIZE ? RN_WINDOW_SIZE - 1 - i : i)];
    }
#pragma unroll
    for (int t = 0; t < RN_WINDOW_SIZE / WAVE; t++) S[lane + WAVE * t] = v[t] * w[t];
    RN_WSYNC();
    const float *run = S + 15 * fft_lam(lane);
#pragma unroll
    for (int b = 0; b < 15; b++) {
      ar[b] = 0.0010416667f * run[fft_c(b)];  // the 1/960 input scale of kiss_fft (src/kiss_fft.c:582)
      ai[b] = 0.0010416667f * 0.f;
    }
    RN_WSYNC();
  };
  // ---- rnn_frame_analysis (src/denoise.c:332-345): window [prev | cur], FFT, Ex ----
  float *gX = g.spec_X[parity] + (size_t)s * RN_SPEC_STRIDE;
  float *y4 = scr + SCR_Y4, *xc = scr + SCR_XC, *rsq = scr + SCR_SQ, *Dsyy = scr + SCR_D;
  int bp0 = 0, bp1 = 0, pitch_index = 0;
  float xx = 0.f;
  // pitch_buf 2x decimated (src/pitch.c:155-166) into this wave's x_lp: 864 consecutive samples of the decimated ring, which the
  // high-pass kernel keeps beside the pitch ring (rn_dev.h: RN_XRING_SLOT) -- every decimated sample is a fixed function of three
  // neighbouring ring samples, formed once when its slot is written, instead of 1728 samples decimated again by every frame that
  // sees them.  Only x_lp[0], which has no left neighbour (src/pitch.c:166), is formed here.
  auto decimate_to_xlp = [&]() {
    const char *xring = reinterpret_cast<const char *>(g.xlp_ring + (size_t)s * RN_XRING_SIZE);
    // byte offsets in 32 bits (the loads take the wave-uniform base as a scalar pair); sample i = lane + 64 t sits 256 t bytes behind
    // sample `lane`; the last round (t = 13) has 32 samples: its upper lanes re-read and rewrite sample 863
    const unsigned a0 = 4u * ((unsigned)ring0 / 2u + (unsigned)lane), a13 = 4u * ((unsigned)ring0 / 2u + 832u + (unsigned)min(lane, 31));
    const float2 pb01 = *reinterpret_cast<const float2 *>(ring + ring0);  // (ring0 is even and < RN_RING_SIZE)
#pragma unroll
    for (int h = 0; h < 2; h++) {  // 864 = 13.5 x 64: two rounds of 7 loads in flight
      float v[7];
#pragma unroll
      for (int t = 0; t < 7; t++) {
        const unsigned a = 7 * h + t < 13 ? a0 + 256u * (7 * h + t) : a13;
        v[t] = *reinterpret_cast<const float *>(xring + min(a, a - 4u * (unsigned)RN_XRING_SIZE));
      }
      if (h == 0 && lane == 0) v[0] = .5f * (.5f * pb01.y + pb01.x);
#pragma unroll
      for (int t = 0; t < 7; t++) {
        if (7 * h + t < 13) xlp[lane + WAVE * (7 * h + t)] = v[t];
        else xlp[832 + min(lane, 31)] = v[t];
      }
    }
  };
  // One-row workgroups: the 5 autocorrelation lags behind the FIR taps are formed HERE, by two spare waves (lags 0..3 on the
  // four rows of wave 1, lag 4 on wave 2, each over its own decimated copy of pitch_buf in its own arena), while wave 0
  // transforms X -- the high-pass kernel of such a row stops after the ring store.  The lags meet wave 0 at a barrier in front
  // of its FIR; the lag window and the Levinson recursion (rn_dev.h) it runs itself.
  float *ac_mail = scr + SCR_EX;  // (a spare wave's Ex area is free)
  if (solo && !active) {
    if (wave == 1 || wave == 2) {
      decimate_to_xlp();
      RN_WSYNC();
      const int row = lane >> 4, lag = wave == 1 ? row : 4;
      const float acv = autocorr_row(to_lds(xlp), lag, lane & 15);
      if ((lane & 15) == 0 && (wave == 1 || row == 0)) ac_mail[lag] = acv;
    }
    __syncthreads();
  }
  // ... and the coarse search's 147 cross-correlations + start energy are split over the four waves: one chain per lane (49 lags
  // on each of waves 0..2, the energy on wave 3) instead of three per lane on wave 0 -- the same products in the same order per
  // chain (chain_dot8_y2 is the scalar chain of chain_dot8_x3), results into wave 0's arena.
  auto coarse_chains_solo = [&]() {
    float *a0 = arenas[0].a;
    if (wave < 3) {
      const int lag = 49 * wave + (lane < 49 ? lane : 48);
      const float v = chain_dot8_y2(to_lds(a0 + SCR_Y4 + 192), to_lds(a0 + ((lag & 1) ? SCR_Y4S + (lag - 1) : SCR_Y4 + lag)), 240);
      if (lane < 49) a0[SCR_XC + lag] = v;
    } else {
      const float e = chain_dot8_y2(to_lds(a0 + SCR_Y4), to_lds(a0 + SCR_Y4), 240, 1.f);
      if (lane == 0) a0[SCR_MAIL + MAIL_SYY0C] = e;
    }
  };
  if (solo && !active) {
    __syncthreads();  // (wave 0's 4x-decimated signal is in place)
    coarse_chains_solo();
  }
  if (active) {  // ======== wide stretch A: transform of X, Ex, downsampling, coarse cross-correlations
  {
    float xr[15], xi[15];
    window_to_regs(xr, xi, RN_PITCH_BUF_SIZE - RN_WINDOW_SIZE, tb.half_window);
    K1_STOP(1);
    regfft960<RN_FFT_XLANE>(xr, xi, lane, ftw);
    K1_STOP(2);
    if (TRAIN) {  // band limit of the TRAINING build (src/denoise.c:340-343)
      const int lp = tr.lowpass[s];
#pragma unroll
      for (int j = 0; j < 8; j++)
        if (WAVE * j + pos >= lp) xr[j] = xi[j] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int bin = WAVE * j + pos;
      if (bin < RN_FREQ_SIZE && wr) reinterpret_cast<float2 *>(gX)[bin] = make_float2(xr[j], xi[j]);
    }
    band_products<1>(Qs, xr, xi, xr, xi, tb, pos, lane);
  }
  band_chain(Ex, Qs, sums, tb, lane);

  CLK_TAP(2);  // window + FFT(X) + Ex
  K1_STOP(3);
  // ---- rnn_pitch_downsample (src/pitch.c:146-214) ----
  // (one-row workgroups: wave 1 has decimated pitch_buf already -- the FIR below takes its input from that wave's arena)
  if (!solo) decimate_to_xlp();
  RN_WSYNC();
  const float *xdec = solo ? arenas[1].a + SCR_XLP : xlp;
  // the 5 FIR taps (autocorrelation + Levinson) were computed by the lane-per-stream kernel K0
  float lpc2[5];
  if (solo) {
    __syncthreads();  // (the spare waves' lags, and wave 1's decimated signal, are in place)
    float acs[5];
#pragma unroll
    for (int k = 0; k < 5; k++) acs[k] = arenas[k < 4 ? 1 : 2].a[SCR_EX + k];
    rn_fir_taps_from_ac(acs, lpc2);
  } else {
#pragma unroll
    for (int k = 0; k < 5; k++) lpc2[k] = g.lpc2[((size_t)slot * g.n_stride + s) * 8 + k];
  }
  if (dbg && lane < 5) dbg[RN_DBG_LPC + lane] = lpc2[lane];
  {  // celt_fir5 in place (src/pitch.c:104-143): outputs are independent given the OLD samples
    float r[14];
#pragma unroll
    for (int t = 0; t < 14; t++) {
      int i = lane + WAVE * t;
      float sum = 0;
      if (i < 864) {
        sum = xdec[i];
        sum = sum + lpc2[0] * (i >= 1 ? xdec[i - 1] : 0.f);
        sum = sum + lpc2[1] * (i >= 2 ? xdec[i - 2] : 0.f);
        sum = sum + lpc2[2] * (i >= 3 ? xdec[i - 3] : 0.f);
        sum = sum + lpc2[3] * (i >= 4 ? xdec[i - 4] : 0.f);
        sum = sum + lpc2[4] * (i >= 5 ? xdec[i - 5] : 0.f);
      }
      r[t] = sum;
    }
    RN_WSYNC();
#pragma unroll
    for (int t = 0; t < 14; t++) {
      int i = lane + WAVE * t;
      if (i < 864) xlp[i] = r[t];
    }
  }
  RN_WSYNC();

  CLK_TAP(3);  // downsample + autocorr + LPC + FIR
  K1_STOP(4);
  if (dbg) for (int i = lane; i < 864; i += WAVE) dbg[RN_DBG_XLP + i] = xlp[i];
  // ---- rnn_pitch_search (src/pitch.c:281-385), len 960, max_pitch 588 ----
  // 4x decimated lp[2j], j<432: y_lp4 = y4[0..386], x_lp4 = y4[192..431] (src/pitch.c:309-312)
  float syy0_coarse = 0.f;
  {
    float *y4s = scr + SCR_Y4S;
    v2f *Z = reinterpret_cast<v2f *>(scr + SCR_Z);
#pragma unroll
    for (int t = 0; t < 7; t++) {  // 432 = 6.75 x 64; clamped, not branched
      const int j0 = lane + WAVE * t, j = j0 < 432 ? j0 : 431;
      const float v = xlp[2 * j];
      y4[j] = v;
      if (j >= 1) y4s[j - 1] = v;  // y4s[i] = y4[i + 1] (only i < 386 is read)
    }
    if (solo) {  // (one-row workgroups: see coarse_chains_solo)
      RN_WSYNC();
      fbp_increments(y4, 240, 147, scr + SCR_SYY, lane);
      __syncthreads();
      coarse_chains_solo();
    } else {
#pragma unroll
    for (int t = 0; t < 5; t++) {  // 289 pairs
      const int j0 = lane + WAVE * t, j = j0 < 289 ? j0 : 288;
      Z[j] = v2f{xlp[2 * j], xlp[2 * j + 98]};
    }
    RN_WSYNC();
    // ONE pass for the 147 lags and the start energy: a lane takes lags l3 and l3 + 49 as a packed pair and lag l3 + 98 as a
    // scalar chain, l3 = 0..48 on lanes 0..30 and 32..49; lane 31 runs 1 + sum y4[j]^2 (src/pitch.c:56-61) as the first
    // chain of its pair -- its x operand is y4 itself and its pairs start at Z[0] = {y4[0], ..}, the address lane 0 of the
    // same 32-lane access group reads anyway (a broadcast, not a bank conflict).  Lanes 50..63 repeat lanes 35..48 of their
    // own group, for the same reason, and are not stored.
    const bool en = lane == 31, stored = lane < 50 && !en;
    const int l3 = en ? 0 : (lane < 31 ? lane : (lane < 50 ? lane - 1 : lane - 15));
    const int a3 = l3 + 98;  // offset of the scalar chain: y4[a3 + j], fetched 8 bytes at a time from the copy that aligns it
    const XC3 r = chain_dot8_x3(to_lds(scr + (en ? SCR_Y4 : SCR_Y4 + 192)), to_lds(scr + (en ? SCR_Z : SCR_Z + 2 * l3)),
                                to_lds(scr + ((a3 & 1) ? SCR_Y4S + (a3 - 1) : SCR_Y4 + a3)), 240, en ? v2f{1.f, 0.f} : v2f{0.f, 0.f});
    K1_STOP(5);
    RN_WSYNC();  // Z is dead; xcorr goes into its area
    if (stored) {
      xc[l3] = r.p.x;
      xc[l3 + 49] = r.p.y;
      xc[l3 + 98] = r.q;
    }
    syy0_coarse = lane_bcast(r.p.x, 31);
    }
  }
  RN_WSYNC();
  CLK_TAP(4);  // coarse xcorr
  K1_STOP(6);
  if (!solo) {
    fbp_increments(y4, 240, 147, scr + SCR_SYY, lane);
    if (lane == 0) mail[MAIL_SYY0C] = syy0_coarse;
  }
  }  // ======== (A)
  WG_SYNC();
  if (wave == nw1) {  // narrow phase 1: the coarse running energy of every stream of the workgroup, one lane each
    // The other waves of the workgroup wait for this one, and its chains are dependent instructions: it takes every issue
    // slot it can use (a lone wave issues once per ~5 cycles whatever its priority; profiles/r3_valu_issue.txt) ahead of the
    // three waves of other workgroups on its SIMD, which have independent work for the remaining slots.
    if (SPW > 1) __builtin_amdgcn_s_setprio(3);
    if (spread) {  // one row of 16 lanes per stream (see sweep_syy_fine_row_x)
      float *ag = ARENA((lane >> 4) < SPW ? (lane >> 4) : 0).a;
      fbp_sweep_row(ag + SCR_SYY, ag[SCR_MAIL + MAIL_SYY0C], lane & 15);
    } else {
      const int gi = lane < SPW ? lane : 0;
      fbp_sweep(arenas[gi].a + SCR_SYY, 147, arenas[gi].a[SCR_MAIL + MAIL_SYY0C], lane < SPW);
    }
    if (SPW > 1) __builtin_amdgcn_s_setprio(1);
  }
  WG_SYNC();
  if (active) {  // ======== wide stretch B: coarse selection, the shifted copy
  best_pitch_select(xc, scr + SCR_SYY, 147, bp0, bp1, lane);
  CLK_TAP(5);  // coarse best-pitch scan
  K1_STOP(7);
  if (dbg) {
    for (int i = lane; i < 147; i += WAVE) dbg[RN_DBG_XC_COARSE + i] = xc[i];
    if (lane == 0) { dbg[RN_DBG_BEST] = bp0; dbg[RN_DBG_BEST + 1] = bp1; }
  }
  RN_WSYNC();
  for (int i = lane; i < 294; i += WAVE) xc[i] = 0;
  if (spread) {
    // x_lp shifted by one sample over the (not yet written) squares: the fine-search chains of phase 2 then fetch their y
    // operand 8 bytes at a time whatever the parity of their lag, as the doubling dots do
#pragma unroll
    for (int k = 0; k < 14; k++) {
      const int i0 = lane + WAVE * k, i = i0 < 863 ? i0 : 862;
      scr[SCR_XS + i] = xlp[i + 1];  // (lanes past the end rewrite element 862 with its own value)
    }
  } else {
    // operands of the running energies of the fine search and of remove_doubling (y4 and the coarse energies are dead)
    energy_sweeps_prepare(xlp, rsq, Dsyy, scr + SCR_ZERO, lane);
  }
  if (lane == 0) {
    mail[MAIL_BP0] = __int_as_float(bp0);
    mail[MAIL_BP1] = __int_as_float(bp1);
  }
  }  // ======== (B)
  K1_STOP(8);
  WG_SYNC();
  if (spread) {
    // narrow phase 2.  Wave nw2: one ROW of 16 lanes per stream -- lanes 0..9 of the row the fine lags, lane 10 xx =
    // <x, x> of remove_doubling, the rest idle along on <x, x> -- so that the chains of a row share x and take it from the
    // row's registers (chain_dot16_xrow), and every y operand is an aligned 8-byte read from x_lp or its shifted copy: a
    // third of the LDS cycles of the 12-lanes-per-stream form below, whose 4-byte reads at 24 unrelated offsets per access
    // group were mostly bank conflicts.  Wave nw3a, at the same time on another SIMD: the start energy 1 + sum x_lp[j]^2
    // of the fine find_best_pitch of every stream, one lane each, and from it the running energy Syy of the fine search,
    // one row per stream (sweep_syy_fine_row_x: its increments are formed on the way, nothing of it waits for nw2's chains).
    if (wave == nw2) {
      if (narrow_prio == 3) __builtin_amdgcn_s_setprio(3);
      const int gq = lane >> 4, r = lane & 15;  // (SPW == 4 rows)
      float *ag = ARENA(gq < SPW ? gq : 0).a;
      const int b0 = __float_as_int(ag[SCR_MAIL + MAIL_BP0]), b1 = __float_as_int(ag[SCR_MAIL + MAIL_BP1]);
      const int c = (r < 5) ? (2 * b0 - 2 + r) : (2 * b1 - 2 + (r - 5));
      const bool lag = r < 10 && c >= 0 && c < 294;
      const int a = lag ? c : 384;
      ldsf xg = to_lds(ag + SCR_XLP + 384), yg = to_lds(ag + ((a & 1) ? SCR_XS + (a - 1) : SCR_XLP + a));
      const float sum = fine_deep ? chain_dot16_xrow<true>(xg, yg, 480, lane) : chain_dot16_xrow<false>(xg, yg, 480, lane);
      if (lag) ag[SCR_XC + c] = (-1 > sum) ? -1 : sum;
      if (r == 10) ag[SCR_MAIL + MAIL_XX] = sum;
      __builtin_amdgcn_s_setprio(1);
    } else if (wave == nw3a) {
      if (narrow_prio == 3) __builtin_amdgcn_s_setprio(3);
      {  // (every lane takes part: the rows' DPP operands come from the other lanes' registers)
        float *a = ARENA((lane >> 4) < SPW ? (lane >> 4) : 0).a;
        const float syy0 = chain_sq_row(to_lds(a + SCR_XLP), 480, 1.f, lane & 15);
        sweep_syy_fine_row_x(a + SCR_XLP, a + SCR_D, syy0, lane & 15);
      }
      __builtin_amdgcn_s_setprio(1);
    } else if (solo && wave != 0 && wave == (nw3b != 0 ? nw3b : nw1)) {
      // One-row workgroups: a wave with nothing to do in this phase forms the frame's first 32 features -- log10 of the band
      // energies, the follower recurrence, their DCT (src/denoise.c:378-397) need Ex only, which wave 0 has had in its arena
      // since the first transform -- and stores them; wave 0's tail then keeps just the energy sum behind `silence`.
      const float *ex0 = arenas[0].a + SCR_EX;
      float dcol[RN_NB_BANDS];
#pragma unroll
      for (int j = 0; j < RN_NB_BANDS; j++) dcol[j] = tb.dct[j * RN_NB_BANDS + (lane & 31)];
      if (lane < RN_NB_BANDS) Ly[lane] = rn_log_energy(ex0[lane], tb.log_tab);
      RN_WSYNC();
      const float e_sum = log_follower(Ly, ex0, lane == 0);
      RN_WSYNC();
      if (lane < RN_NB_BANDS && in_range) {
        float f_lo = dct_lane(Ly, dcol, tb);
        if (lane == 0) f_lo -= 12;
        if (lane == 1) f_lo -= 4;
        g.features[(size_t)s * 68 + lane] = ((double)e_sum < 0.04) ? 0.f : f_lo;
      }
    }
  } else {
    // one-stream workgroups: narrow phase 2 on 12 lanes -- lanes 0..9 the fine lags, lane 10 xx = <x, x> of remove_doubling,
    // lane 11 the start energy 1 + sum x_lp[j]^2 of the fine find_best_pitch -- then phase 3 on two: the fine running
    // energy and yy_lookup
    {
      const int r = lane;
      const bool on = lane < 12;
      const int b0 = __float_as_int(mail[MAIL_BP0]), b1 = __float_as_int(mail[MAIL_BP1]);
      const int c = (r < 5) ? (2 * b0 - 2 + r) : (2 * b1 - 2 + (r - 5));
      const bool lag = on && r < 10 && c >= 0 && c < 294;
      float sum = 0;
      if (lag || (on && r >= 10))
        sum = chain_dot8(to_lds(xlp + (r == 11 ? 0 : 384)), to_lds(xlp + (lag ? c : (r == 11 ? 0 : 384))), 480,
                         r == 11 ? 1.f : 0.f);
      if (lag) xc[c] = (-1 > sum) ? -1 : sum;
      if (on && r == 10) mail[MAIL_XX] = sum;
      if (on && r == 11) mail[MAIL_SYY0F] = sum;
    }
    RN_WSYNC();
    CLK_TAP(6);  // fine xcorr (+ the two start energies)
    energy_sweeps_run(scr + SCR_SQ, scr + SCR_D, scr + SCR_ZERO, mail[MAIL_SYY0F], mail[MAIL_XX], lane & 1, lane < 2);
    CLK_TAP(9);  // fine-search Syy + yy_lookup sweeps
  }
  WG_SYNC();
  K1_STOP(9);
  if (active) {  // ======== wide stretch C: fine selection
  xx = mail[MAIL_XX];
  best_pitch_select(xc, Dsyy - 1, 294, bp0, bp1, lane);
  CLK_TAP(7);  // fine best-pitch selection
  K1_STOP(10);
  int offset = 0;
  if (bp0 > 0 && bp0 < 293) {
    float a = xc[bp0 - 1], b = xc[bp0], c = xc[bp0 + 1];
    if ((c - a) > .7f * (b - a)) offset = 1;
    else if ((a - c) > .7f * (b - c)) offset = -1;
  }
  pitch_index = RN_PITCH_MAX_PERIOD - (2 * bp0 - offset);
  if (dbg) {
    for (int i = lane; i < 294; i += WAVE) dbg[RN_DBG_XC_FINE + i] = xc[i];
    if (lane == 0) { dbg[RN_DBG_BEST + 2] = bp0; dbg[RN_DBG_BEST + 3] = bp1; dbg[RN_DBG_BEST + 4] = offset; dbg[RN_DBG_BEST + 5] = pitch_index; }
  }
  }  // ======== (C)

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
    // yy_lookup[i] lives at scr[SCR_YYL - 1 + i], over the dead Syy / fine xcorr areas, UNCLAMPED: the readers below apply the
    // reference's MAX32(0, yy) (src/pitch.c:452-455; yy_lookup[0] = xx, a sum of squares, is its own maximum with 0)
    float *yyl = scr + SCR_YYL - 1;
    if (spread) {
      // x_lp's shifted copy is still in place (nothing was staged over it), and yy_lookup is swept NOW, beside the candidate
      // dots, by a third wave -- straight into its final place, which is free once every stream of the workgroup is through
      // its fine selection: that, and T0 for the wave that runs this stream's dots, is what the barrier stands for
      if (lane == 0) mail[MAIL_T0] = __int_as_float(T0);
      __syncthreads();
    } else {
      RN_WSYNC();  // the fine xcorr and the fine Syy are dead from here on
      // one-stream workgroups: yy_lookup moves out of the squares array into the dead areas, and the squares array becomes
      // xs[i] = x_lp[i + 1]: every dot product below can then fetch its y operand 8 bytes at a time from an 8-byte aligned
      // address, whatever the parity of its offset
      float *xs = scr + SCR_XS;
      float t[7], u[14];
#pragma unroll
      for (int k = 0; k < 7; k++) {  // 385 = 6.02 x 64
        const int i0 = lane + WAVE * k, i = i0 <= 384 ? i0 : 384;
        t[k] = rsq[479 + i];
      }
#pragma unroll
      for (int k = 0; k < 14; k++) {
        const int i0 = lane + WAVE * k + 1;
        u[k] = xlp[i0 < 864 ? i0 : 863];
      }
      RN_WSYNC();
#pragma unroll
      for (int k = 0; k < 7; k++) {
        const int i0 = lane + WAVE * k;
        yyl[i0 <= 384 ? i0 : 384] = t[k];  // lanes past the end rewrite element 384 with its own value
      }
#pragma unroll
      for (int k = 0; k < 14; k++) {
        const int i0 = lane + WAVE * k;
        if (i0 < 863) xs[i0] = u[k];
      }
      RN_WSYNC();
    }
    K1_STOP(11);
    // Every dot product the decision loop can ask for, in ONE pass of 480-step chains on 31 lanes (each chain is an
    // independent serial sum, so computing it speculatively changes no bit):
    //   lane 1: xy(T0);  lanes 2..29: (k, T1 / T1b), k = 2..15 (pitch.c:462-483);  lanes 30, 31: the -1 / +1 neighbours of T0,
    //   which the final 3-point refinement (pitch.c:511-512) needs when no shorter period wins.
    //   (xx, the chain at offset 0, came out of energy_sweeps)
    // The neighbours of a shorter period T1(k) are fetched by a second, two-lane pass only when such a k wins.  (Round 2 ran
    // all 30 neighbour chains speculatively on lanes 32..61: 59 lanes with unrelated offsets collide on the LDS banks --
    // 2,860 LDS cycles per frame, half of them conflicts, a third of the whole kernel's; 31 lanes fill one 32-lane access
    // group and leave the other empty.)
    // Paired (workgroups of four streams): the pass costs the issue slots of 64 lanes whether 31 or 62 of them carry a chain,
    // so two waves of the workgroup run it for two streams each -- lanes 0..31 one stream, 32..63 the other, each half an LDS
    // access group of its own reading its own arena -- and the other two wait at the barrier without issuing anything: half
    // the instructions per frame for the kernel's largest chain pass (every lane's chain is what it was).
    auto candidate_dots = [&](float *ag, int T0g, int l) {
      int off = -1;
      if (l == 1) off = T0g;
      else if (l >= 2 && l < 30) {
        int k = 2 + ((l - 2) >> 1);
        int T1 = (2 * T0g + k) / (2 * k), T1b;
        if (k == 2) T1b = (T1 + T0g > maxperiod) ? T0g : T0g + T1;
        else T1b = (2 * sc[k] * T0g + k) / (2 * k);
        off = ((l - 2) & 1) ? T1b : T1;
      } else if (l == 30 || l == 31) {
        off = T0g + ((l & 1) ? 1 : -1);
        if (off < 0) off = 0;
      }
      if (dots_noconf && off >= 0) off = 2 * l;
      // every lane runs a chain (chain_dot16_xrow takes x from the registers of the other lanes of its row); the lanes without
      // an offset of their own run <x, x>, all of them on the same addresses (a broadcast, not a bank conflict), and drop it
      const int a = maxperiod - (off >= 0 ? off : 0);  // y = x_lp + a
      ldsf xg = to_lds(ag + SCR_XLP + maxperiod), ya = to_lds(ag + ((a & 1) ? SCR_XS + (a - 1) : SCR_XLP + a));
      const float d = xrow ? chain_dot16_xrow<false>(xg, ya, N, lane) : (off >= 0 ? chain_dot8_y2(xg, ya, N) : 0.f);
      if (off >= 0) ag[SCR_DOTS + l] = d;
    };
    if (spread) {
      if (wave == nw1 || wave == nw3b) {
        if (narrow_prio == 3) __builtin_amdgcn_s_setprio(3);
        float *ag = ARENA((wave == nw1 ? 0 : 2) + (lane >> 5)).a;
        const int T0g = __float_as_int(ag[SCR_MAIL + MAIL_T0]);
        candidate_dots(ag, T0g, lane & 31);
        __builtin_amdgcn_s_setprio(1);
      } else if (wave == nw2) {  // yy_lookup of every stream, one row each (every lane takes part: DPP operands)
        if (narrow_prio == 3) __builtin_amdgcn_s_setprio(3);
        float *a = ARENA((lane >> 4) < SPW ? (lane >> 4) : 0).a;
        sweep_yy_lookup_row_x(a + SCR_XLP, a + SCR_YYL, a[SCR_MAIL + MAIL_XX], lane & 15);
        __builtin_amdgcn_s_setprio(1);
      } else if (!solo) {
        // nw3a, the one wave with nothing to do in this phase: log10 of the band energies and the log-energy follower (src/denoise.c:
        // 378-388) of EVERY stream of the workgroup -- they need Ex only, in its arena since the first transform.  The follower is a
        // 32-step dependent chain of ~10 instructions a step whatever the number of lanes: run by each wave for its own stream it cost
        // four times the issue slots (~450 of a wave's 7,400 VALU instructions, profiles/r6_k1_follower.txt).  One row of 16 lanes per
        // stream: two of the stream's 32 logarithms per lane, then the chain on the row's first lane.  Ly's place (over the dead fine
        // cross-correlations) is written by nothing else until the features read it.
        const int gq = lane >> 4, r = lane & 15;  // (SPW == 4 rows)
        float *ag = ARENA(gq).a;
        float *ly = ag + (SCR_MISC + 80 + 64);
        const float *ex = ag + SCR_EX;
        const float e0 = ex[r], e1 = ex[16 + r];
        ly[r] = rn_log_energy(e0, tb.log_tab);
        ly[16 + r] = rn_log_energy(e1, tb.log_tab);
        RN_WSYNC();
        if (r == 0) ag[SCR_MAIL + MAIL_E] = log_follower(ly, ex, true);
      }
      __syncthreads();
      if (!active) return;  // (that was the last barrier: the surplus waves of a one-row workgroup are done)
    } else {
      candidate_dots(scr, T0, lane);
    }
    RN_WSYNC();
    float xy = dots[1];
    CLK_TAP(8);  // 59 candidate dot products of remove_doubling
    K1_STOP(12);
    float yy = fmaxf(0.f, yyl[T0]);
    float best_xy = xy, best_yy = yy;
    if (dbg && lane == 0) { dbg[RN_DBG_DOTS] = xx; dbg[RN_DBG_DOTS + 1] = xy; dbg[RN_DBG_DOTS + 2] = yy; }
    float g0, gg;
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
      const float yy1 = .5f * (fmaxf(0.f, yyl[T1]) + fmaxf(0.f, yyl[T1b]));
      // (one evaluation of compute_pitch_gain -- a double-precision square root and a division, ~60 instructions -- for both uses:
      //  lane 0 forms g0 of T0, which every threshold below needs, lanes 2..15 their own g1)
      const float g1 = pitch_gain(lane == 0 ? xy : xy1, xx, lane == 0 ? yy : yy1);
      g0 = lane_bcast(g1, 0);
      gg = g0;
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
    float xc0 = dots[30], xc2 = dots[31];
    if (cand) {  // (wave-uniform) a shorter period won: its two neighbours, T >= minperiod = 30 so both offsets are valid
      RN_WSYNC();
      {
        const int a = maxperiod - (lane < 2 ? T + (lane ? 1 : -1) : 0);
        const float d = chain_dot16_xrow<false>(to_lds(x), to_lds(scr + ((a & 1) ? SCR_XS + (a - 1) : SCR_XLP + a)), N, lane);
        if (lane < 2) dots[32 + lane] = d;
      }
      RN_WSYNC();
      xc0 = dots[32];
      xc2 = dots[33];
    }
    if (dbg && lane == 0) { dbg[RN_DBG_DOTS + 3] = T; dbg[RN_DBG_DOTS + 4] = xc0; dbg[RN_DBG_DOTS + 5] = xc1; dbg[RN_DBG_DOTS + 6] = xc2; }
    int off2 = 0;
    if ((xc2 - xc0) > .7f * (xc1 - xc0)) off2 = 1;
    else if ((xc0 - xc2) > .7f * (xc1 - xc2)) off2 = -1;
    if (pg > gg) pg = gg;
    pitch_index = 2 * T + off2;
    if (pitch_index < minperiod0) pitch_index = minperiod0;
    pgain = pg;
    RN_WSYNC();
  }
  if (lane == 0 && wr) {
    g.last_period[s] = pitch_index;
    g.last_gain[s] = pgain;
    g.pitch[s] = pitch_index;
  }

  CLK_TAP(10);  // doubling decisions + 3 final dots
  K1_STOP(13);
  // ---- pitch-aligned frame -> P, Ep, Exp (src/denoise.c:371-377) ----
  float dctc[RN_NB_BANDS];  // this lane's DCT column, requested now, consumed after the band energies
  {
    float pr[15], pi[15], xr[8], xi[8];
    // opaque copies of the table pointers: otherwise the window values, twiddles and lane-derived indices the first
    // transform used are kept alive (and spilled to scratch) across the whole pitch stage instead of being re-read from L1/L2
    const float *hw2 = tb.half_window;
    const float2 *ftw2 = ftw;
    int lane2 = lane;
    asm volatile("" : "+s"(hw2), "+s"(ftw2), "+v"(lane2));
    lane2 &= WAVE - 1;  // (its range, for the compiler: table indices formed from it then fold into the loads' immediate offsets)
    window_to_regs(pr, pi, RN_PITCH_BUF_SIZE - RN_WINDOW_SIZE - __builtin_amdgcn_readfirstlane(pitch_index), hw2);
    // X is read back from HBM/L2 (this wave wrote it; its stores are long complete), in the lane-owns-bins layout
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int bin = WAVE * j + pos;
      const float2 v = reinterpret_cast<const float2 *>(gX)[bin < RN_FREQ_SIZE ? bin : 0];
      xr[j] = v.x;
      xi[j] = v.y;
    }
#pragma unroll
    for (int j = 0; j < RN_NB_BANDS; j++) dctc[j] = tb.dct[j * RN_NB_BANDS + (lane & 31)];
    K1_STOP(14);
    regfft960<RN_FFT_XLANE>(pr, pi, lane2, ftw2);
    K1_STOP(15);
    float *gP = g.spec_P[parity] + (size_t)s * RN_SPEC_STRIDE;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int bin = WAVE * j + pos;
      if (bin < RN_FREQ_SIZE && wr) reinterpret_cast<float2 *>(gP)[bin] = make_float2(pr[j], pi[j]);
    }
    band_products<2>(Qs, xr, xi, pr, pi, tb, pos, lane);  // <P, P> and <X, P> side by side
  }
  band_sums<2>(sums, Qs, tb, lane);
  if (lane < RN_NB_BANDS) {
    Ep[lane] = band_of_sums(sums, lane);
    Exp[lane] = band_of_sums(sums + 40, lane);
  }
  RN_WSYNC();
  K1_STOP(16);
  float *gE = g.spec_E[parity] + (size_t)s * 96;
  if (lane < RN_NB_BANDS) {
    Exp[lane] = (float)((double)Exp[lane] / sqrt(.001 + (double)(Ex[lane] * Ep[lane])));
    if (wr) {
      gE[lane] = Ex[lane];
      gE[32 + lane] = Ep[lane];
      gE[64 + lane] = Exp[lane];
    }
  }
  RN_WSYNC();

  // ---- features (src/denoise.c:378-397) ----
  float *feat = g.features + (size_t)s * 68;
  if (!solo && !spread && lane < RN_NB_BANDS) Ly[lane] = rn_log_energy(Ex[lane], tb.log_tab);
  RN_WSYNC();
  // log-energy follower + total energy (log_follower above): 32 serial steps
  float E = 0;
  if (solo) {  // (the follower and the first 32 features came from a spare wave during narrow phase 2)
    for (int i = 0; i < RN_NB_BANDS; i++) E += Ex[i];
  } else if (spread) {  // (run for the whole workgroup by the idle wave of narrow phase 3: Ly holds the followed values)
    E = mail[MAIL_E];
  } else {
    E = log_follower(Ly, Ex, lane == 0);  // (evaluated uniformly by every lane)
  }
  RN_WSYNC();
  // inference: silent frames zero the features and skip the network (src/denoise.c:389-393);
  // TRAINING build: features are always produced and "silence" means E < 0.1 (:389,:397)
  const int silence = TRAIN ? (((double)E < 0.1) ? 1 : 0) : (((double)E < 0.04) ? 1 : 0);
  const bool zero = !TRAIN && silence;
  // the two DCTs (src/denoise.c:387,396) in ONE pass: lanes 0..31 the column sums over the followed log energies (features 0..31),
  // lanes 32..63 the same columns over Exp (features 32..63) -- feature = lane.  (One-row workgroups: the first 32 exist already.)
  {
    const bool hi = solo || lane >= RN_NB_BANDS;
    float f = dct_lane(hi ? Exp : Ly, dctc, tb);
    if (!hi && lane == 0) f -= 12;
    if (!hi && lane == 1) f -= 4;
    const int fi = solo ? RN_NB_BANDS + lane : lane;
    if (wr && (!solo || lane < RN_NB_BANDS)) {
      feat[fi] = zero ? 0.f : f;
      if (TRAIN) tr.rec[(size_t)s * 98 + fi] = f;
    }
  }
  if (lane == 0 && wr) {
    const float fp = (float)(.01 * (double)(pitch_index - 300));
    feat[2 * RN_NB_BANDS] = zero ? 0.f : fp;
    if (TRAIN) tr.rec[(size_t)s * 98 + 2 * RN_NB_BANDS] = fp;
    g.silence[s] = silence;
  }
  if (TRAIN) {
    // rnn_frame_analysis of the CLEAN frame (src/dump_features.c:468) and the band-gain targets (:472-478)
    RN_WSYNC();
    float *cm = tr.clean_mem + (size_t)s * RN_FRAME_SIZE;
    const float *cx = tr.clean + (size_t)s * RN_FRAME_SIZE;
    float *Ey = Ep;  // Ep already went to HBM
    {
      float yr[15], yi[15];
      for (int i = lane; i < RN_WINDOW_SIZE; i += WAVE) {
        const float w = tb.half_window[i < RN_FRAME_SIZE ? i : RN_WINDOW_SIZE - 1 - i];
        S[i] = (i < RN_FRAME_SIZE ? cm[i] : cx[i - RN_FRAME_SIZE]) * w;
      }
      RN_WSYNC();
      const float *run = S + 15 * fft_lam(lane);
#pragma unroll
      for (int b = 0; b < 15; b++) {
        yr[b] = 0.0010416667f * run[fft_c(b)];
        yi[b] = 0.0010416667f * 0.f;
      }
      RN_WSYNC();
      regfft960<RN_FFT_XLANE>(yr, yi, lane, ftw);
      for (int i = lane; i < RN_FRAME_SIZE; i += WAVE) cm[i] = cx[i];
      const int lp = tr.lowpass[s];
#pragma unroll
      for (int j = 0; j < 8; j++)
        if (WAVE * j + pos >= lp) yr[j] = yi[j] = 0.f;
      band_products<1>(Qs, yr, yi, yr, yi, tb, pos, lane);
    }
    band_chain(Ey, Qs, sums, tb, lane);
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
}
#undef ARENA

// (4 waves per SIMD is what the LDS allows: 16 arenas of 10 KB per CU; without the cap the allocator spreads to 154 VGPRs)
extern "C" __global__ void __launch_bounds__(WAVE * K1_SPW) __attribute__((amdgpu_waves_per_eu(4, 4)))
rn_analysis_kernel(RnGroupDev g, RnTablesDev tb, int slot, int parity) {
  // Above the high-pass kernel's waves (priority 0), which run beside this kernel two frames ahead and are in no hurry:
  // one of them per SIMD, always ready with an old instruction, otherwise takes issue slots from four analysis waves
  // (29.14 -> 29.41 M frames/s at 65,536 streams; RNNOISE_AMD_K1_PRIO=0 switches it off for A/B runs).
  if (slot & 256) __builtin_amdgcn_s_setprio(1);
  analysis_body<false, K1_SPW>(g, tb, slot & ~256, parity, RnTrainArgs{});
}
// One stream per workgroup: batches that fit in one round of resident waves (<= 16 per CU) are bound by a wave's latency,
// not by instruction issue, and there the narrow phases are better run by every wave for itself.  Measured on MI355X
// (pipelined bench, M frames/s, 1 vs K1_SPW streams per workgroup): 2048 streams 13.9 / 14.0, 4096: 20.0 / 18.3,
// 8192: 17.7 / 19.6, 16,384: 19.0 / 22.1.  (Round 1's 80-VGPR "lean" build no longer pays at any size and is gone.)
extern "C" __global__ void __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(4, 4)))
rn_analysis_single_kernel(RnGroupDev g, RnTablesDev tb, int slot, int parity, RnRows rows) {
  // (a launch group of the one-frame API: the block's row at that row's own frame phase.  ONE instance of the body: two made the
  //  kernel 80 KB of code, more than the instruction cache two CUs share)
  const bool listed = rows.n > 0;
  const uint32_t re = listed ? rows.e[blockIdx.x] : 0u;
  analysis_body<false, 1>(g, tb, listed ? RN_ROW_RING(re) : slot, listed ? RN_ROW_SPEC(re) : parity, RnTrainArgs{},
                          listed ? RN_ROW_OF(re) : -1);
}

// A launch group of the one-frame API, for LATENCY: one workgroup of K1_SPW waves per listed row, every wave working on that
// row's frame.  The waves do the same wide work four times over (on four SIMDs that have nothing else to do: a group is at most
// 64 rows on 256 CUs) so that the frame's serial chains run side by side as in the batched kernel -- the fine
// cross-correlations beside the start energy and the fine running energy, the candidate dots beside yy_lookup -- instead of one
// after the other on the stream's only wave.  Only wave 0 stores.
extern "C" __global__ void __launch_bounds__(WAVE * K1_SPW) __attribute__((amdgpu_waves_per_eu(4, 4)))
rn_analysis_rows_kernel(RnGroupDev g, RnTablesDev tb, RnRows rows) {
  const uint32_t re = rows.e[blockIdx.x];
  analysis_body<false, K1_SPW>(g, tb, RN_ROW_RING(re), RN_ROW_SPEC(re), RnTrainArgs{}, RN_ROW_OF(re));
}

// TRAINING-mode variant: the inner loop of src/dump_features.c:466-491
extern "C" __global__ void __launch_bounds__(WAVE)
rn_train_features_kernel(RnGroupDev g, RnTablesDev tb, int slot, int parity, RnTrainArgs tr) {
  analysis_body<true, 1>(g, tb, slot, parity, tr);
}


struct SynthLds {
  float S[1052];  // band products (RnTablesDev::band_q slots); then the Hermitian-extended spectrum in natural order, the real
                  // parts and the imaginary parts one after the other (staging for the transform)
  float misc[192];
};
// 4,976 B: two of these waves fit into the LDS that four analysis workgroups (4 x 38,016 B) leave free on a CU -- with the
// complex spectrum staged in one piece (8,448 B) it was one, and synthesis mostly waited for analysis workgroups to drain
static_assert(sizeof(SynthLds) <= 5120 && RN_WINDOW_SIZE <= 1052 && RN_BAND_QSTRIDE <= 1052, "synthesis LDS");

// ---------------------------------------------------------------------------------------------
// K3: rnn_pitch_filter + gain smoothing/interpolation + frame_synthesis
// (src/denoise.c:474-496, 421-455, 140-154, 400-407, 200-217)
// Lane l owns bins p, p+64, ..., p+448 (p = fft_pos(l); bin 480 is the j = 7 bin of the lane with p = 32) in registers
// through the band stages; every HBM operand is requested before the first dependent instruction, so the wave pays one
// memory round trip instead of one per stage.  The inverse transform is the register-resident FFT of fft_reg.h: the
// Hermitian-extended spectrum passes once through LDS (natural order in, 15 consecutive bins out per lane) and the time
// samples come out in registers, lane l holding work-area positions 64*blk + p.  4.9 KB of LDS per wave.
// ---------------------------------------------------------------------------------------------
// (5 waves per SIMD: the transform needs ~95 registers, and the operands of the overlap-add are requested behind it -- see below)
#ifndef RN_K3_WAVES
#define RN_K3_WAVES 5  // (A/B builds: -DRN_K3_WAVES=6 spills 12 registers)
#endif
// LATE: the overlap-add operands behind the transform (the throughput form); !LATE: with everything else at the top (a handful of
// waves on an empty machine have nobody to cover the extra round trip: rn_synthesis_few_kernel)
// where a synthesis wave works: its stream, its lane number, its 4.9 KB of LDS (a one-wave workgroup of its own -- or, lab build, a wave
// of a fused analysis workgroup: rn_analysis_synth_kernel)
struct SynthPlace {
  int s, lane;
  char *lds;
};
// Where transform output b of a lane goes (src/denoise.c:213-216, 400-407).  The lane holds y[p], p = 64 b + pos, pos = fft_pos(lane) in
// 0..63; time sample n = (960 - p) % 960.  lo = "p == 0 or p > 480" = n < 480: an output sample, out[n] = 960 y w[n] + synthesis_mem[n];
// otherwise synthesis_mem[n - 480] = 960 y w[959 - n] = 960 y w[p - 1], n - 480 = 480 - p.  Returns the window index (n or p - 1) and sets
// `mem` to the synthesis_mem index (n or 480 - p).  b is a compile-time constant wherever this is called (unrolled loops), so each case
// is a constant plus pos or q = 63 - pos -- written out because the generic expression costs a signed modulo, two compares, an exec-mask
// branch and a 64-bit address per b when the compiler does not know pos's range (15 instructions x 15: a tenth of the kernel).
//   b = 0: lo iff pos == 0, and n = max(pos, 1) - 1 either way;   b = 1..6: never lo;   b = 7: lo iff pos > 32;   b = 8..14: always lo
__device__ __forceinline__ unsigned synth_index(int b, unsigned pos, bool &lo, unsigned &mem) {
  const unsigned q = 63u - pos;
  if (b == 0) {
    lo = pos == 0;
    const unsigned n = max(pos, 1u) - 1u;
    mem = lo ? 0u : 417u + q;
    return n;
  }
  if (b < 7) {
    lo = false;
    mem = (417u - 64u * b) + q;
    return (64u * b - 1u) + pos;
  }
  if (b == 7) {
    lo = pos > 32u;
    mem = lo ? 512u - pos : 32u - pos;
    return lo ? 512u - pos : 447u + pos;
  }
  lo = true;
  mem = (897u - 64u * b) + q;
  return mem;
}
template <bool LATE>
__device__ __forceinline__ void synthesis_body(const RnGroupDev &g, const RnTablesDev &tb, float *__restrict__ out, int parity_arg, int prev_arg,
                                               const RnRows &rows, const SynthPlace *place = nullptr) {
  // bit 8 of parity_arg: `out` holds int16 samples, written with the truncating conversion of the reference's only caller
  // (examples/rnnoise_demo.c:58: tmp[i] = x[i], float -> short as x86 compiles it: cvttss2si to 32 bits -- "integer
  // indefinite" 0x80000000 when out of range or NaN -- then the low 16 bits)
  const bool listed = rows.n > 0;  // a launch group of the one-frame API (rn_dev.h: RnRows)
  const uint32_t re = listed ? rows.e[blockIdx.x] : 0u;
  const int parity = listed ? RN_ROW_SPEC(re) : (parity_arg & 255);
  const int prev = listed ? (parity + RN_SPEC_SLOTS - 1) % RN_SPEC_SLOTS : prev_arg;
  const bool out_s16 = !listed && (parity_arg & 256);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  SynthLds &L = *reinterpret_cast<SynthLds *>(place ? place->lds : smem_raw);
  const int s = place ? place->s : (listed ? RN_ROW_OF(re) : (int)blockIdx.x), lane = place ? place->lane : (int)threadIdx.x, pos = fft_pos(lane);
  const float2 *dX = reinterpret_cast<const float2 *>(g.spec_X[prev] + (size_t)s * RN_SPEC_STRIDE);
  const float2 *dP = reinterpret_cast<const float2 *>(g.spec_P[prev] + (size_t)s * RN_SPEC_STRIDE);
  const float *dE = g.spec_E[prev] + (size_t)s * 96;
  const float *cE = g.spec_E[parity] + (size_t)s * 96;
  float *r = L.misc + 0, *gsm = L.misc + 32, *newE = L.misc + 64, *norm = L.misc + 96, *sums = L.misc + 128;
  float *Q = L.S;
  const int silence = g.silence[s];
  constexpr int NBIN = 8;  // bins pos + 64*j

  // ---- every HBM operand up front ----
  float2 X[NBIN], P[NBIN];
  float frac[NBIN];
  uint32_t bq[NBIN];  // RnTablesDev::band_q: slot of the (1-frac) term | slot of the frac term << 11 | band << 22
#pragma unroll
  for (int j = 0; j < NBIN; j++) {
    const int bin = pos + WAVE * j;
    const bool ok = bin < RN_FREQ_SIZE;
    X[j] = ok ? dX[bin] : make_float2(0.f, 0.f);
    P[j] = (ok && !silence) ? dP[bin] : make_float2(0.f, 0.f);
    bq[j] = (bin < 400) ? tb.band_q[bin] : 0u;
    frac[j] = (bin < 400) ? tb.band_frac[bin] : 0.f;
  }
  const int pad_slot = tb.band_pad[lane];  // (band_sums: the pad floats of the product array hold +0.0f)
  float e_ex = 0, e_ep = 0, e_exp = 0, c_ex = 0, gi = 0, lastg = 0;
  if (lane < RN_NB_BANDS && !silence) {
    e_ex = dE[lane]; e_ep = dE[32 + lane]; e_exp = dE[64 + lane]; c_ex = cE[lane];
    gi = g.gains[(size_t)s * RN_NB_BANDS + lane];
    lastg = g.lastg[(size_t)s * RN_NB_BANDS + lane];
  }
  // After the transform this lane holds y[p], p = 64*blk + pos, blk = 0..14; time sample n = (960 - p) % 960
  // (src/denoise.c:213-216).  n < 480: out[n] = 960*y*w[n] + synthesis_mem[n];  n >= 480: synthesis_mem[n - 480] =
  // 960*y*w[959 - n] = 960*y*w[p - 1]  (src/denoise.c:400-407).
  float *sm = g.synth_mem + (size_t)s * RN_FRAME_SIZE;
  float smv[15], wv[15];
  if (!LATE) {
#pragma unroll
    for (int b = 0; b < 15; b++) {
      bool lo;
      unsigned mi;
      const unsigned wi = synth_index(b, (unsigned)pos & 63u, lo, mi);
      wv[b] = tb.half_window[wi];
      smv[b] = lo ? sm[mi] : 0.f;
    }
  }

// src/denoise.c:140-154 per bin (bins >= 400 -> 0), from a 32-entry band vector in LDS
#define BAND(j) ((int)(bq[j] >> 22))
#define INTERP(vec, j)                                                                                      \
  ((pos + WAVE * (j)) >= 400 ? 0.f                                                                          \
   : BAND(j) == 0 ? (vec)[0]                                                                                \
   : BAND(j) == RN_NB_BANDS ? (vec)[RN_NB_BANDS - 1]                                                        \
                            : (1 - frac[j]) * (vec)[BAND(j) - 1] + frac[j] * (vec)[BAND(j)])

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
    Q[pad_slot] = 0.f;
    RN_WSYNC();
#pragma unroll
    for (int j = 0; j < NBIN; j++) {  // :441-445, then the products of compute_band_energy (:446)
      const int bin = pos + WAVE * j;
      const float rf = INTERP(r, j);
      X[j].x += rf * P[j].x;
      X[j].y += rf * P[j].y;
      if (bin < 400) {
        float tmp = X[j].x * X[j].x;
        tmp += X[j].y * X[j].y;
        Q[(bq[j] >> 11) & 0x7ff] = frac[j] * tmp;
        Q[bq[j] & 0x7ff] = (1 - frac[j]) * tmp;
      }
    }
    band_chain(newE, Q, sums, tb, lane);
    if (lane < RN_NB_BANDS) {
      norm[lane] = (float)sqrt((double)e_ex / (1e-8 + (double)newE[lane]));  // :447-449
      const float alpha = .6f;  // gain smoothing (src/denoise.c:479-487)
      gi = (gi > alpha * lastg) ? gi : alpha * lastg;
      double q = (double)gi * ((double)e_ex + 1e-3) / ((double)c_ex + 1e-3);
      g.lastg[(size_t)s * RN_NB_BANDS + lane] = (float)((1.f < q) ? 1.f : q);
      gsm[lane] = gi;
    }
    RN_WSYNC();
#pragma unroll
    for (int j = 0; j < NBIN; j++) {  // :450-454 then :488-493
      const float nf = INTERP(norm, j), gf = INTERP(gsm, j);
      X[j].x *= nf;
      X[j].y *= nf;
      X[j].x *= gf;
      X[j].y *= gf;
    }
    RN_WSYNC();
  }
#undef INTERP
#undef BAND
  // inverse_transform (src/denoise.c:200-217): Hermitian extension through the FORWARD FFT.  Natural order into LDS and
  // this lane's 15 consecutive bins out of it (fft_reg.h "Input"): first the real parts, then the imaginary parts
  float yr[15], yi[15];
  {
    const float *run = L.S + 15 * fft_lam(lane);
#pragma unroll
    for (int part = 0; part < 2; part++) {
#pragma unroll
      for (int j = 0; j < NBIN; j++) {
        const int bin = pos + WAVE * j;
        if (bin < RN_FREQ_SIZE) {
          const float v = 0.0010416667f * (part ? X[j].y : X[j].x);
          L.S[bin] = v;
          if (bin > 0 && bin < RN_FREQ_SIZE - 1) L.S[RN_WINDOW_SIZE - bin] = part ? 0.0010416667f * (-X[j].y) : v;
        }
      }
      RN_WSYNC();
#pragma unroll
      for (int b = 0; b < 15; b++) (part ? yi : yr)[b] = run[fft_c(b)];
      RN_WSYNC();
    }
  }
  regfft960<RN_FFT_XLANE>(yr, yi, lane, reinterpret_cast<const float2 *>(tb.fft_tw));
  // The operands of the overlap-add -- 15 window values (L1 / L2) and the lane's 7 or 8 synthesis_mem samples (HBM) -- are
  // requested only NOW.  Requested with the other operands at the top they sat in 30 registers through the band stages and the
  // transform: 142 VGPRs, three waves per SIMD, 0.29 ms at 65,536 streams; behind the transform the kernel needs 96, five
  // waves fit, and four of them cover the fifth's wait for these loads: 0.237 ms (profiles/r4_k3_occupancy.txt).  The index
  // goes through an empty asm together with a transform output so that the scheduler cannot hoist the loads back up.
  if (LATE) {
    int pos_late = pos;
    asm volatile("" : "+v"(pos_late), "+v"(yr[14]));
    const unsigned pl = (unsigned)pos_late & 63u;  // (its range, for the compiler)
#pragma unroll
    for (int b = 0; b < 15; b++) {
      bool lo;
      unsigned mi;
      const unsigned wi = synth_index(b, pl, lo, mi);
      wv[b] = tb.half_window[wi];
      smv[b] = lo ? sm[mi] : 0.f;
    }
  }
  // window + overlap-add (src/denoise.c:400-407), straight from the registers
  float *o = listed ? rows.io + (size_t)s * RN_ROW_IO + RN_FRAME_SIZE + 4 : out + (size_t)s * RN_FRAME_SIZE;
#pragma unroll
  for (int b = 0; b < 15; b++) {
    bool lo;
    unsigned n;  // (lo: the output sample's index; otherwise where the sample goes in synthesis_mem)
    (void)synth_index(b, (unsigned)pos & 63u, lo, n);
    float v = (float)RN_WINDOW_SIZE * yr[b];
    v *= wv[b];
    if (lo) {
      const float r = v + smv[b];
      if (out_s16) {
        const int q = (r >= -2147483648.f && r < 2147483648.f) ? (int)r : (int)0x80000000;
        reinterpret_cast<short *>(out)[(size_t)s * RN_FRAME_SIZE + n] = (short)q;
      } else {
        o[n] = r;
      }
    } else {
      sm[n] = v;
    }
  }
  if (listed) {
    // completion word of the row's request (the last word of its pinned block): the caller waiting for this frame polls it
    // instead of waiting for the whole stream to drain.  System-scope release: the frame and the VAD are visible before it.
    __threadfence_system();
    if (lane == 0) __hip_atomic_store(reinterpret_cast<uint32_t *>(rows.io + (size_t)s * RN_ROW_IO + RN_ROW_IO - 1), RN_ROW_SEQ(re),
                                      __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
extern "C" __global__ void __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(RN_K3_WAVES, RN_K3_WAVES)))
rn_synthesis_kernel(RnGroupDev g, RnTablesDev tb, float *__restrict__ out, int parity_arg, int prev_arg, RnRows rows) {
  synthesis_body<true>(g, tb, out, parity_arg, prev_arg, rows);
}
// the launch groups of the one-frame API and batches of up to RN_K3_FEW_MAX streams (one frame: 10.4 -> 9.6 us)
#define RN_K3_FEW_MAX 256
extern "C" __global__ void __launch_bounds__(WAVE)
rn_synthesis_few_kernel(RnGroupDev g, RnTablesDev tb, float *__restrict__ out, int parity_arg, int prev_arg, RnRows rows) {
  synthesis_body<false>(g, tb, out, parity_arg, prev_arg, rows);
}

#if RN_INSTRUMENT
// ---- LAB (instrumented build only; VERDICT r5 Next #2, profiles/r6_fused_k3k1.txt) ----
// Synthesis of frame t-1 as the PROLOGUE of the analysis of frame t: same wave = same stream, in the wave's own arena (4.9 of its 9.3 KB),
// at the analysis kernel's occupancy.  The question: does the stage with the most HBM traffic per instruction hide under the
// issue-bound one when they are phases of ONE kernel (waves of a CU drift apart between the six barriers of a workgroup), where as
// separate kernels they cannot co-reside (profiles/r5_overlap.txt)?  gs: the group as frame t-1 sees it (its features / silence /
// gains buffers); synth_cur < 0: no synthesis (first frame of a call).
extern "C" __global__ void __launch_bounds__(WAVE * K1_SPW) __attribute__((amdgpu_waves_per_eu(4, 4)))
rn_analysis_synth_kernel(RnGroupDev g, RnGroupDev gs, RnTablesDev tb, int slot, int parity, float *__restrict__ out, int synth_cur, int synth_prev) {
  if (slot & 256) __builtin_amdgcn_s_setprio(1);
  {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), s = (int)blockIdx.x * K1_SPW + wave;
    if (synth_cur >= 0 && s < gs.n_streams) {
      const SynthPlace pl{s, (int)(threadIdx.x & (WAVE - 1)), smem_raw + wave * sizeof(AnalysisLds)};
      synthesis_body<true>(gs, tb, out, synth_cur, synth_prev, RnRows{}, &pl);
      RN_WSYNC();
    }
  }
  analysis_body<false, K1_SPW>(g, tb, slot & ~256, parity, RnTrainArgs{});
}
extern "C" hipError_t rn_launch_analysis_synth(const RnGroupDev *g, const RnGroupDev *gs, const RnTablesDev *tb, int slot, int parity, void *out,
                                               int out_s16, int synth_cur, int synth_prev, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  const dim3 grid((g->n_streams + K1_SPW - 1) / K1_SPW), block(WAVE * K1_SPW);
  RN_LAUNCH(rn_analysis_synth_kernel, grid, block, K1_SPW * sizeof(AnalysisLds), st, e0, e1, *g, *gs, *tb, slot | 256, parity,
            static_cast<float *>(out), synth_cur < 0 ? -1 : (synth_cur | (out_s16 ? 256 : 0)), synth_prev);
  return hipGetLastError();
}
#endif

// host-visible launch helpers -----------------------------------------------------------------
// (K0 lives in hp_kernel.hip; K0 and K1 are launched separately so that the host may put K0 of the next frame on a side stream)
extern "C" hipError_t rn_launch_hp_passthrough(const RnGroupDev *g, const float *in, int slot, hipStream_t st);  // hp_kernel.hip
extern "C" hipError_t rn_launch_analysis(const RnGroupDev *g, const RnTablesDev *tb, int slot, int parity, hipStream_t st,
                                         hipEvent_t e0, hipEvent_t e1) {
  // A/B runs only: RNNOISE_AMD_K1_SPW=1 / 4 forces one / K1_SPW streams per workgroup; RNNOISE_AMD_K1_LDS -> a larger
  // LDS request per wave lowers the waves per CU (occupancy experiments)
  static const int spw_force = [] { const char *e = getenv("RNNOISE_AMD_K1_SPW"); return e ? atoi(e) : 0; }();
  static const size_t lds1 = [] { const char *e = RN_LAB_ENV("K1_LDS"); return e ? (size_t)atoi(e) : sizeof(AnalysisLds); }();
  const int n = g->n_streams;
  const bool single = spw_force == 1 || (spw_force == 0 && n < RN_K1_MULTI_MIN_STREAMS);
  if (single) {
    RN_LAUNCH(rn_analysis_single_kernel, dim3(n), dim3(WAVE), lds1, st, e0, e1, *g, *tb, slot, parity, RnRows{});
  } else {
    const dim3 grid((n + K1_SPW - 1) / K1_SPW), block(WAVE * K1_SPW);
    static const int prio = [] { const char *e = RN_LAB_ENV("K1_PRIO"); return (e && atoi(e) == 0) ? 0 : 256; }();
    static const int stop = [] { const char *e = RN_LAB_ENV("K1_STOP"); return (RN_INSTRUMENT && e) ? atoi(e) << 16 : 0; }();
    static const int noxrow = [] {
      const char *e = RN_LAB_ENV("K1_XROW"), *x = RN_LAB_ENV("K1_EXPERIMENT");  // (A/B bits 12, 14: see analysis_body)
      return ((e && atoi(e) == 0) ? 1024 : 0) | (x ? (atoi(x) & (4096 | 8192 | 16384)) : 0);
    }();
    RN_LAUNCH(rn_analysis_kernel, grid, block, K1_SPW * lds1, st, e0, e1, *g, *tb, slot | prio | stop | noxrow, parity);
  }
  return hipGetLastError();
}
extern "C" hipError_t rn_launch_train_features(const RnGroupDev *g, const RnTablesDev *tb, const float *noisy, int slot,
                                               int parity, const RnTrainArgs *tr, hipStream_t st) {
  hipError_t e = rn_launch_hp_passthrough(g, noisy, slot, st);  // training frames arrive filtered: K0 without the biquad
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(rn_train_features_kernel, dim3(g->n_streams), dim3(WAVE), sizeof(AnalysisLds), st, *g, *tb,
                     slot, parity, *tr);
  return hipGetLastError();
}
extern "C" hipError_t rn_launch_synthesis(const RnGroupDev *g, const RnTablesDev *tb, void *out, int out_s16, int cur, int prev,
                                          hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  if (g->n_streams <= RN_K3_FEW_MAX)
    RN_LAUNCH(rn_synthesis_few_kernel, dim3(g->n_streams), dim3(WAVE), sizeof(SynthLds), st, e0, e1, *g, *tb, static_cast<float *>(out),
              cur | (out_s16 ? 256 : 0), prev, RnRows{});
  else
    RN_LAUNCH(rn_synthesis_kernel, dim3(g->n_streams), dim3(WAVE), sizeof(SynthLds), st, e0, e1, *g, *tb, static_cast<float *>(out),
              cur | (out_s16 ? 256 : 0), prev, RnRows{});
  return hipGetLastError();
}
// K1 / K3 of a launch group of the one-frame API (rn_dev.h: RnRows): one one-wave workgroup per listed row
extern "C" hipError_t rn_launch_analysis_rows(const RnGroupDev *g, const RnTablesDev *tb, const RnRows *rows, hipStream_t st) {
  // RNNOISE_AMD_ROWS_K1=1 (A/B runs): one wave per row (rn_analysis_single_kernel) instead of a workgroup of four
  static const bool one_wave = [] { const char *e = RN_LAB_ENV("ROWS_K1"); return e && atoi(e) == 1; }();
  if (one_wave) hipLaunchKernelGGL(rn_analysis_single_kernel, dim3(rows->n), dim3(WAVE), sizeof(AnalysisLds), st, *g, *tb, 0, 0, *rows);
  else hipLaunchKernelGGL(rn_analysis_rows_kernel, dim3(rows->n), dim3(WAVE * K1_SPW), K1_SPW * sizeof(AnalysisLds), st, *g, *tb, *rows);
  return hipGetLastError();
}
extern "C" hipError_t rn_launch_synthesis_rows(const RnGroupDev *g, const RnTablesDev *tb, const RnRows *rows, hipStream_t st) {
  hipLaunchKernelGGL(rn_synthesis_few_kernel, dim3(rows->n), dim3(WAVE), sizeof(SynthLds), st, *g, *tb, static_cast<float *>(nullptr), 0, 0, *rows);
  return hipGetLastError();
}
