// nn_gru.h -- the GRU layer kernel body of the layer-wise network (nn_layers.hip) and what it is made of: the x86-profile
// activations in the form this kernel wants them, the LDS-DMA piece, the rolling A-fragment buffer.  Included by nn_layers.hip (the
// product's two instantiations: four waves / 72 KB and eight waves / 152 KB) and, in the instrumented build only, by
// lab/nn_gru_lab.hip (A/B variants, the row-buffer check of profiles/r5_gru_race.txt).
#pragma once
#include "nn_common.h"
#include <type_traits>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifndef RN_GRU_PACKED_ACT
#define RN_GRU_PACKED_ACT 1  // the activations of the layer kernel in packed FP32 (gru_body: PK)
#endif
#define GM 4  // 16-stream tiles per workgroup
#ifndef GW
#define GW 8  // waves per workgroup
#endif
#define GTHREADS (64 * GW)

// W: waves per GRU workgroup, HB: f32 row buffers per wave.  <4, 1>: one wave per SIMD and workgroup, 72 KB -- two workgroups per
// CU, or one beside analysis workgroups of the next frame (batches with more 64-stream groups than CUs); <8, 3>: one buffer per
// unit tile of a wave, every buffer written once per launch, 152 KB -- a workgroup owns its CU (smaller batches).
// FOLD (instrumented build only): dense_out / vad_dense advanced inside the layer kernel -- lab/nn_gru_fold.inc has the staging area
// (GruFoldLds), the code (spliced into gru_body at five places below) and the record of why the product does not do it
// (profiles/r6_dense_fold.txt).
#ifdef RN_GRU_LAB
#define RN_FOLD_PART 0
#include "lab/nn_gru_fold.inc"
#undef RN_FOLD_PART
#else
template <int W>
struct GruFoldLds {};  // (the product never instantiates FOLD)
#endif
struct GruNoFoldLds {};
template <int W, int HB, bool FOLD = false>
struct GruLdsT {
  uint16_t lut[4096];            // rcpps table (rn_dev.h: rcp16) -- FIRST: the *_lut0 activations take the table index for its LDS address
  int8_t xq[GM][KT * 64 * 16];   // layer input images
  int8_t hq[GM][KT * 64 * 16];   // recurrent state images
  float hrow[W][HB][GM * TS][16];  // per wave (and unit tile, HB == 24 / W): the f32 state of its 16 units for the workgroup's
                                   // 64 streams (the blend z*h + (1-z)*candidate needs them exact)
  typename std::conditional<FOLD, GruFoldLds<W>, GruNoFoldLds>::type fold;
};
static_assert(sizeof(GruLdsT<8, 3>) <= 160 * 1024, "one workgroup per CU, all of its LDS");
static_assert(sizeof(GruLdsT<4, 1>) <= 80 * 1024, "two workgroups per CU");
static_assert(sizeof(GruLdsT<8, 1, true>) <= 160 * 1024, "one workgroup per CU, all of its LDS");

// Addressing in the GRU kernel is (uniform base, unsigned 32-bit BYTE offset): one VGPR per address instead of a 64-bit
// pair per pointer (rn_launch_nn_layers refuses batches whose state plane exceeds 4 GB)
template <typename T>
__device__ __forceinline__ T ldg(const void *base, unsigned byte_off) {
  return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off);
}
template <typename T>
__device__ __forceinline__ void stg(void *base, unsigned byte_off, T v) {
  *reinterpret_cast<T *>(reinterpret_cast<char *>(base) + byte_off) = v;
}
// ---- the activations and the quantiser of nn_common.h with fewer VALU operations (this kernel's main loop is VALU-bound) ----
// Same bits for every finite argument -- which is all a GRU layer can see: its pre-activations are int32 sums times
// finite scales plus diag * h, and h stays in [-1, 1] from a zero or any finite start.  What differs from nn_common.h:
//   * the two clamps are one v_med3_f32 (differs from the x86 min/max pair only for a NaN argument);
//   * the u8 quantiser is v_rndne + v_cvt_pk_u8_f32 (saturating both ways like packs/packus; differs only for
//     |127 x + 127| >= 2^31, where cvtps2dq's "integer indefinite" turns a huge positive value into 0).
__device__ __forceinline__ float rcp_b(float x, const uint16_t *lut) { return rn_rcp_x86(x, lut); }
__device__ __forceinline__ float tanh_g(float x, const uint16_t *lut_b) {  // src/vec_avx.h:398-416
  const float N0 = 952.52801514f, N1 = 96.39235687f, N2 = 0.60863042f;
  const float D0 = 952.72399902f, D1 = 413.36801147f, D2 = 11.88600922f;
  const float x2 = x * x;
  float num = fmaf(fmaf(N2, x2, N1), x2, N0);
  const float den = fmaf(fmaf(D2, x2, D1), x2, D0);
  num = num * x;
  num = num * rcp_b(den, lut_b);
  return __builtin_amdgcn_fmed3f(num, -1.f, 1.f);
}
__device__ __forceinline__ float sigmoid_g(float x, const uint16_t *lut_b) {  // src/vec_avx.h:426-445
  const float N0 = 238.13200378f, N1 = 6.02452230f, N2 = 0.00950985f;
  const float D0 = 952.72399902f, D1 = 103.34200287f, D2 = 0.74287558f;
  const float x2 = x * x;
  float num = fmaf(fmaf(N2, x2, N1), x2, N0);
  const float den = fmaf(fmaf(D2, x2, D1), x2, D0);
  num = num * x;
  num = fmaf(num, rcp_b(den, lut_b), .5f);
  return __builtin_amdgcn_fmed3f(num, 0.f, 1.f);
}
// The same two activations in two halves, so that the table lookups of MANY elements are in flight together: the first half
// ends by requesting the element's rcpps table entry from LDS, the second half uses it.  Taken one element at a time (sigmoid,
// sigmoid, tanh, each waiting for its own lookup) the 48 activations of a unit tile spent 8-15 k cycles, most of them waiting
// for LDS round trips one after the other (shader-clock taps, tools/k1_cycles.py --layers).  Same operations in the same
// order per element: same bits.
struct ActPre {
  float numx;     // num * x
  uint32_t b, v;  // bits of den; its table entry (rn_rcp_x86)
};
__device__ __forceinline__ ActPre act_pre(float x, const uint16_t *lut, float N0, float N1, float N2, float D0, float D1, float D2) {
  const float x2 = x * x;
  const float num = fmaf(fmaf(N2, x2, N1), x2, N0);
  const float den = fmaf(fmaf(D2, x2, D1), x2, D0);
  ActPre a;
  a.numx = num * x;
  a.b = __float_as_uint(den);
  a.v = lut[(a.b >> 11) & 0xfff];
  return a;
}
__device__ __forceinline__ float act_rcp(const ActPre &a) { return __uint_as_float((a.v << 11) + (RN_RCP_K - (a.b & 0x7f800000u))); }
__device__ __forceinline__ ActPre sigmoid_pre(float x, const uint16_t *lut) {  // src/vec_avx.h:426-445
  return act_pre(x, lut, 238.13200378f, 6.02452230f, 0.00950985f, 952.72399902f, 103.34200287f, 0.74287558f);
}
__device__ __forceinline__ float sigmoid_fin(const ActPre &a) { return __builtin_amdgcn_fmed3f(fmaf(a.numx, act_rcp(a), .5f), 0.f, 1.f); }
__device__ __forceinline__ ActPre tanh_pre(float x, const uint16_t *lut) {  // src/vec_avx.h:398-416
  return act_pre(x, lut, 952.52801514f, 96.39235687f, 0.60863042f, 952.72399902f, 413.36801147f, 11.88600922f);
}
__device__ __forceinline__ float tanh_fin(const ActPre &a) { return __builtin_amdgcn_fmed3f(a.numx * act_rcp(a), -1.f, 1.f); }
// ... and on PAIRS of elements in packed math (v_pk_mul / v_pk_fma / v_pk_add_f32: two elements per instruction).  Each wave
// of this kernel is alone on its SIMD's VALU most of the time (its partner is in its MFMA block), and a lone wave issues one
// instruction per ~5 cycles whatever the instruction: halving the instruction count of the polynomial halves its time.  The
// packed forms round each component exactly like the scalar ones (an fma is an fma, a multiply a multiply).
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
struct ActPre2 {
  v2f numx;
  uint32_t b0, b1, v0, v1;
};
__device__ __forceinline__ ActPre2 act_pre2(v2f x, const uint16_t *lut, float N0, float N1, float N2, float D0, float D1, float D2) {
  const v2f x2 = x * x;
  const v2f num = pk_fma(pk_fma(v2f{N2, N2}, x2, v2f{N1, N1}), x2, v2f{N0, N0});
  const v2f den = pk_fma(pk_fma(v2f{D2, D2}, x2, v2f{D1, D1}), x2, v2f{D0, D0});
  ActPre2 a;
  a.numx = num * x;
  a.b0 = __float_as_uint(den.x);
  a.b1 = __float_as_uint(den.y);
  a.v0 = lut[(a.b0 >> 11) & 0xfff];
  a.v1 = lut[(a.b1 >> 11) & 0xfff];
  return a;
}
__device__ __forceinline__ v2f act_rcp2(const ActPre2 &a) {
  return v2f{__uint_as_float((a.v0 << 11) + (RN_RCP_K - (a.b0 & 0x7f800000u))), __uint_as_float((a.v1 << 11) + (RN_RCP_K - (a.b1 & 0x7f800000u)))};
}
__device__ __forceinline__ ActPre2 sigmoid_pre2(v2f x, const uint16_t *lut) {
  return act_pre2(x, lut, 238.13200378f, 6.02452230f, 0.00950985f, 952.72399902f, 103.34200287f, 0.74287558f);
}
__device__ __forceinline__ v2f sigmoid_fin2(const ActPre2 &a) {
  const v2f r = pk_fma(a.numx, act_rcp2(a), v2f{.5f, .5f});
  return v2f{__builtin_amdgcn_fmed3f(r.x, 0.f, 1.f), __builtin_amdgcn_fmed3f(r.y, 0.f, 1.f)};
}
__device__ __forceinline__ ActPre2 tanh_pre2(v2f x, const uint16_t *lut) {
  return act_pre2(x, lut, 952.52801514f, 96.39235687f, 0.60863042f, 952.72399902f, 413.36801147f, 11.88600922f);
}
__device__ __forceinline__ v2f tanh_fin2(const ActPre2 &a) {
  const v2f r = a.numx * act_rcp2(a);
  return v2f{__builtin_amdgcn_fmed3f(r.x, -1.f, 1.f), __builtin_amdgcn_fmed3f(r.y, -1.f, 1.f)};
}
// The same pair activations for kernels whose rcpps table sits at LDS address 0 (gru_body3 checks it): the table index IS the LDS
// address ((bits >> 10) & 0x1ffe: no base to add -- with a dynamic-LDS base the compiler emits `v_add_u32 v, 0, v` per lookup, the
// symbol being resolved after instruction selection), and the reconstruction is ((v << 11) + K) - exponent: v_lshl_add_u32 + v_sub
// instead of shift, subtract, add.  2 of ~22 VALU operations per activation; the same integers, so the same bits.
typedef const __attribute__((address_space(3))) uint16_t *lds_u16_ptr;
__device__ __forceinline__ ActPre2 act_pre2_lut0(v2f x, float N0, float N1, float N2, float D0, float D1, float D2) {
  const v2f x2 = x * x;
  const v2f num = pk_fma(pk_fma(v2f{N2, N2}, x2, v2f{N1, N1}), x2, v2f{N0, N0});
  const v2f den = pk_fma(pk_fma(v2f{D2, D2}, x2, v2f{D1, D1}), x2, v2f{D0, D0});
  ActPre2 a;
  a.numx = num * x;
  a.b0 = __float_as_uint(den.x);
  a.b1 = __float_as_uint(den.y);
  a.v0 = *(lds_u16_ptr)(size_t)((a.b0 >> 10) & 0x1ffeu);
  a.v1 = *(lds_u16_ptr)(size_t)((a.b1 >> 10) & 0x1ffeu);
  return a;
}
__device__ __forceinline__ v2f act_rcp2_k(const ActPre2 &a) {
  return v2f{__uint_as_float(((a.v0 << 11) + RN_RCP_K) - (a.b0 & 0x7f800000u)), __uint_as_float(((a.v1 << 11) + RN_RCP_K) - (a.b1 & 0x7f800000u))};
}
__device__ __forceinline__ ActPre2 sigmoid_pre2_lut0(v2f x) {
  return act_pre2_lut0(x, 238.13200378f, 6.02452230f, 0.00950985f, 952.72399902f, 103.34200287f, 0.74287558f);
}
__device__ __forceinline__ v2f sigmoid_fin2_k(const ActPre2 &a) {
  const v2f r = pk_fma(a.numx, act_rcp2_k(a), v2f{.5f, .5f});
  return v2f{__builtin_amdgcn_fmed3f(r.x, 0.f, 1.f), __builtin_amdgcn_fmed3f(r.y, 0.f, 1.f)};
}
__device__ __forceinline__ ActPre2 tanh_pre2_lut0(v2f x) {
  return act_pre2_lut0(x, 952.52801514f, 96.39235687f, 0.60863042f, 952.72399902f, 413.36801147f, 11.88600922f);
}
__device__ __forceinline__ v2f tanh_fin2_k(const ActPre2 &a) {
  const v2f r = a.numx * act_rcp2_k(a);
  return v2f{__builtin_amdgcn_fmed3f(r.x, -1.f, 1.f), __builtin_amdgcn_fmed3f(r.y, -1.f, 1.f)};
}
// ... and the same address / reconstruction shortcuts element by element (PK == false: no packed math in the kernel)
__device__ __forceinline__ ActPre act_pre_lut0(float x, float N0, float N1, float N2, float D0, float D1, float D2) {
  const float x2 = x * x;
  const float num = fmaf(fmaf(N2, x2, N1), x2, N0);
  const float den = fmaf(fmaf(D2, x2, D1), x2, D0);
  ActPre a;
  a.numx = num * x;
  a.b = __float_as_uint(den);
  a.v = *(lds_u16_ptr)(size_t)((a.b >> 10) & 0x1ffeu);
  return a;
}
__device__ __forceinline__ float act_rcp_k(const ActPre &a) { return __uint_as_float(((a.v << 11) + RN_RCP_K) - (a.b & 0x7f800000u)); }
__device__ __forceinline__ ActPre sigmoid_pre_lut0(float x) {
  return act_pre_lut0(x, 238.13200378f, 6.02452230f, 0.00950985f, 952.72399902f, 103.34200287f, 0.74287558f);
}
__device__ __forceinline__ float sigmoid_fin_k(const ActPre &a) { return __builtin_amdgcn_fmed3f(fmaf(a.numx, act_rcp_k(a), .5f), 0.f, 1.f); }
__device__ __forceinline__ ActPre tanh_pre_lut0(float x) {
  return act_pre_lut0(x, 952.52801514f, 96.39235687f, 0.60863042f, 952.72399902f, 413.36801147f, 11.88600922f);
}
__device__ __forceinline__ float tanh_fin_k(const ActPre &a) { return __builtin_amdgcn_fmed3f(a.numx * act_rcp_k(a), -1.f, 1.f); }
__device__ __forceinline__ int pack4_g(float a, float b, float c, float d) {  // src/vec_avx.h:326-341, then -128 per byte
  unsigned p = 0;
  p = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_rintf(fmaf(a, 127.f, 127.f)), 0, p);
  p = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_rintf(fmaf(b, 127.f, 127.f)), 1, p);
  p = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_rintf(fmaf(c, 127.f, 127.f)), 2, p);
  p = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_rintf(fmaf(d, 127.f, 127.f)), 3, p);
  return (int)(p ^ 0x80808080u);
}

// One LDS-DMA piece: 64 lanes x 16 bytes from per-lane global addresses to 1 KB of LDS at lds_dst (wave-uniform byte address).
// Issued from asm so that hipcc does not count it: it would otherwise drain the piece (vmcnt) before the next LDS read of
// ANY address.  The waits are explicit below; hipcc's own vmcnt(N) for its loads can only over-wait (in-order counter).
__device__ __forceinline__ void dma_1k(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);  // (wave-uniform by construction, but derived from threadIdx: not provably)
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void *p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const void *)p;
}
// acc[gate][t] += W(row tile 24 gate + u) . image[t]: the three gates of a unit tile share the layer input, so one B
// fragment read from LDS feeds three MFMAs and one A fragment from L2 four.  (Measured with the 1 x 4 blocking of the
// first version: a 16x16x64 MFMA takes 16 cycles on its SIMD, its 1 KB B fragment 8 cycles of the CU's one LDS port --
// four SIMDs re-reading B per MFMA are LDS-bound at half the MFMA rate.)
// The A fragments come from L2 (~600 cycles): a rolling buffer keeps them AD k-steps ahead of their MFMAs, across the
// boundary between the input and the recurrent matrix (step = 0..5 input, 6..11 recurrent).
// (AD is a template parameter of the kernel body; 3 and 4 k-steps ahead were measured too and change nothing: profiles/r4_gru_experiments.txt)
#ifdef RN_GRU_LAB
#define RN_FOLD_PART 6
#include "lab/nn_gru_fold.inc"
#undef RN_FOLD_PART
#endif
template <int AD>
struct AFrags {
  v4i f[AD + 1][3];
};
template <int AD>
__device__ __forceinline__ void a_fetch(AFrags<AD> &A, int step, const int8_t *__restrict__ wi, const int8_t *__restrict__ wr, unsigned a0) {
  const int8_t *a = step < KT ? wi : wr;
  const int kt = step < KT ? step : step - KT;
#pragma unroll
  for (int gate = 0; gate < 3; gate++) A.f[step % (AD + 1)][gate] = ldg<v4i>(a, a0 + (unsigned)((gate * 24 * KT + kt) * 1024));
}
// k-steps [s0, s0 + KT) of the rolling sequence: acc[gate][t] += A(step)[gate] . image[t]
template <int AD>
__device__ __forceinline__ void int8_gates(v4i acc[3][GM], AFrags<AD> &A, int s0, const int8_t *__restrict__ wi, const int8_t *__restrict__ wr,
                                           unsigned a0, int lane, const int8_t (*bq)[KT * 64 * 16]) {
  asm volatile("" : "+v"(lane));  // (the images do not change inside the kernel: keep the compiler from hoisting all 48 B fragments)
#pragma unroll
  for (int kt = 0; kt < KT; kt++) {
    const int step = s0 + kt;
    if (step + AD < 2 * KT) a_fetch(A, step + AD, wi, wr, a0);
    __builtin_amdgcn_sched_barrier(0);  // (else the scheduler sinks the fetch to its use to save registers: every A fragment an exposed L2 trip)
    v4i bf[GM];
#pragma unroll
    for (int t = 0; t < GM; t++) bf[t] = reinterpret_cast<const v4i *>(bq[t])[kt * 64 + lane];
#pragma unroll
    for (int gate = 0; gate < 3; gate++)
#pragma unroll
      for (int t = 0; t < GM; t++)
        acc[gate][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A.f[step % (AD + 1)][gate], bf[t], acc[gate][t], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// layer_arg: the layer, 0..2.  The instrumented build's A/B runs carry switches above it (the product ignores every bit but the
// layer's two): bit 2: activations element by element; bits 3-4: what stands between the wait for this wave's row DMA and its reads
// of those rows, beyond s_waitcnt vmcnt(0) -- 0 nothing, 1 = lgkmcnt(0) + s_sleep (256 clocks), 2 = a workgroup barrier,
// 3 = lgkmcnt(0) + s_nop ladder; bit 5: no issue priority for the older waves.
// PK: the activation polynomials on PAIRS of elements in packed math (v_pk_mul / v_pk_fma / v_pk_add_f32) -- see DESIGN.md section 2
// ("packed FP32 beside MFMAs") for which form the product takes and why.  CHK (lab/nn_gru_lab.hip only): the row-buffer check.
// FOLD: the output chains (below).
template <int AD, int W, int HB, bool CHK, bool DMA = true, bool PK = RN_GRU_PACKED_ACT, bool FOLD = false>
__device__ __forceinline__ void gru_body(const RnGroupDev &g, const RnModelDev &m, const RnTablesDev &tb, int layer_arg) {
  typedef GruLdsT<W, HB, FOLD> GruLds;
  static_assert(24 % W == 0 && (HB == 24 / W || HB <= 2), "unit tiles per wave; row buffers");
  static_assert(!FOLD || (W == 2 * GM && !CHK), "output chains: the older half of the waves owns them, wave w both row tiles of stream tile w");
  const int layer = layer_arg & 3;
#if RN_INSTRUMENT
  const bool batched_act = !(layer_arg & 4), no_prio = layer_arg & 32;
  const int settle = (layer_arg >> 3) & 3;
#else
  constexpr bool batched_act = true, no_prio = false;
  constexpr int settle = 0;
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  GruLds &L = *reinterpret_cast<GruLds *>(lds_raw);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 15, gq = lane >> 4;
  const int N = g.n_streams, n_tiles = (N + TS - 1) / TS, tile0 = blockIdx.x * GM;
  const uint16_t *lut = L.lut;
  if (lds_addr(L.lut) != 0) __builtin_trap();  // (the *_lut0 activations take the table index for its LDS address)
  float *st = g.gru_state + (size_t)layer * g.n_stride * RN_GRU;
  const int8_t *xin = g.act_q[layer];
  int8_t *himg = g.act_q[layer + 1];  // quantised state: read here, rewritten below (own tiles only)

  // (tests / profiling: shader-clock taps of wave 0, slots RN_DBG_CLK2 + 7 + 3 * layer + {0: prologue, 1: loads issued -> barrier, 2: tiles})
#if RN_INSTRUMENT
  float *dbg = (g.debug && tid == 0) ? g.debug + (size_t)tile0 * TS * RN_DBG_FLOATS + RN_DBG_CLK2 + 7 + 3 * layer : nullptr;
#else
  float *const dbg = nullptr;
#endif
  const unsigned long long clk0 = (RN_INSTRUMENT && g.debug) ? __builtin_amdgcn_s_memtime() : 0;
  int sn[GM], sil[GM];
  unsigned livemask = 0;  // bit t: stream (tile0 + t) * 16 + n exists and is not silent (one register through the unit tiles, not four)
#pragma unroll
  for (int t = 0; t < GM; t++) {
    const int s = (tile0 + t) * TS + n;
    sn[t] = s < N ? s : N - 1;
    sil[t] = g.silence[(unsigned)sn[t]];
  }
  // Prologue: the two images of the workgroup's GM tiles and the rcpps table go straight from HBM to LDS (1 KB per wave
  // instruction, no staging registers, no ds_write pass: the images are stored in exactly the order LDS wants), then
  // this wave's f32 rows for its first unit tile.
  // (DMA == false, A/B variant "w4nodma": the same pieces through registers and ds_write_b128)
  auto piece = [&](const void *gsrc, const void *lds_base, unsigned off) {
    if (DMA) dma_1k(gsrc, lds_addr(lds_base) + off);
    else *reinterpret_cast<v4i *>(const_cast<char *>(static_cast<const char *>(lds_base)) + off + lane * 16) = *static_cast<const v4i *>(gsrc);
  };
  auto row_buf = [&](int ui) { return HB == 24 / W ? ui : ui % HB; };
  auto rows_fetch = [&](int ui) {  // f32 state of units 16 u .. 16 u + 15, u = wave + W ui, of the 64 streams: 4 pieces
    const int u = wave + W * ui;
    // (the addresses are formed HERE, from an opaque copy of the lane number: hoisted out of the unit-tile loop they are eight registers
    //  through the MFMA blocks, and in the FOLD form of the kernel the allocator spilled them -- a spill reload is a vector memory load,
    //  its s_waitcnt vmcnt(0) sat between the DMA pieces, and every piece waited for the one before it to come back from HBM)
    int l_ = lane;
    asm volatile("" : "+v"(l_));
#pragma unroll
    for (int i = 0; i < GM * TS * 16 * 4 / 1024; i++) {
      const int idx = i * 64 + l_, row = idx >> 2, seg = idx & 3, s = tile0 * TS + row, sc = s < N ? s : N - 1;
      piece(st + ((size_t)sc * RN_GRU + 16 * u + 4 * seg), &L.hrow[wave][row_buf(ui)][0][0], i * 1024);
    }
  };
  {
    constexpr int NCHUNK = 2 * GM * KT;  // 1 KB pieces
#pragma unroll
    for (int j = 0; j < (NCHUNK + W - 1) / W; j++) {
      const int c = wave + j * W;  // wave-uniform
      if (c < NCHUNK) {
        const int which = c / (GM * KT), cc = c - which * (GM * KT), t = cc / KT, kt = cc - t * KT;
        const int tile = (tile0 + t < n_tiles) ? tile0 + t : n_tiles - 1;
        piece((which ? himg : xin) + ((size_t)tile * (KT * 64 * 16) + kt * 1024 + lane * 16), which ? L.hq[t] : L.xq[t], kt * 1024);
      }
    }
#pragma unroll
    for (int c = wave; c < 8; c += W)  // the LUT is 8 pieces
      piece(reinterpret_cast<const uint32_t *>(tb.rcp16) + c * 256 + lane * 4, L.lut, c * 1024);
    rows_fetch(0);
#ifdef RN_GRU_LAB
#define RN_FOLD_PART 1
#include "lab/nn_gru_fold.inc"
#undef RN_FOLD_PART
#endif
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  const unsigned long long clk1 = (RN_INSTRUMENT && g.debug) ? __builtin_amdgcn_s_memtime() : 0;
  __builtin_amdgcn_s_barrier();
  const unsigned long long clk2 = (RN_INSTRUMENT && g.debug) ? __builtin_amdgcn_s_memtime() : 0;
#ifdef RN_GRU_LAB  // (the row-buffer check: lab/nn_gru_lab.hip)
  auto image_check = [&](int when) {
    auto cmp = [&](const void *lds, const void *hbm, int words, int region) {
      for (int w = tid; w < words; w += 64 * W) {
        const unsigned got = reinterpret_cast<const unsigned *>(lds)[w], want = reinterpret_cast<const unsigned *>(hbm)[w];
        if (got != want) {
          atomicAdd(&rn_gru_race_log[3], 1u);
          const unsigned k = atomicAdd(&rn_gru_race_log[0], 1u);
          if (k < 40) {
            unsigned *rec = rn_gru_race_log + 4 + 12 * k;
            rec[0] = blockIdx.x; rec[1] = 0xffff0000u + 0x100u * when + region; rec[2] = w; rec[3] = got; rec[4] = want; rec[5] = wave;
          }
        }
      }
    };
    for (int t = 0; t < GM; t++) {
      const int tile = (tile0 + t < n_tiles) ? tile0 + t : n_tiles - 1;
      cmp(L.xq[t], xin + (size_t)tile * (KT * 64 * 16), KT * 64 * 4, t);
      if (when == 0) cmp(L.hq[t], himg + (size_t)tile * (KT * 64 * 16), KT * 64 * 4, 4 + t);
    }
    cmp(L.lut, tb.rcp16, 2048, 8);
  };
  if (CHK) {
    image_check(0);
    __builtin_amdgcn_s_barrier();  // (nobody rewrites its tiles' state image before everybody has compared it)
  }
#endif

#pragma unroll
  for (int t = 0; t < GM; t++) livemask |= ((tile0 + t) * TS + n < N && !sil[t]) ? 1u << t : 0u;  // silent streams keep their state (src/denoise.c:474)
  const RnLinearDev &wi = m.gru_in[layer], &wr = m.gru_rec[layer];
  // The two waves of a SIMD run the same phases from the same barrier: left alone they want the MFMA pipe together and
  // the VALU together.  Giving one of them issue priority lets it run ahead, after which one's MFMA block overlaps the
  // other's epilogue.
  if (wave < W / 2 && !no_prio) __builtin_amdgcn_s_setprio(2);
  [[maybe_unused]] v4f h_prev[GM] = {};
#ifdef RN_GRU_LAB
#define RN_FOLD_PART 2
#include "lab/nn_gru_fold.inc"
#undef RN_FOLD_PART
#endif
#pragma unroll 1
  for (int ui = 0; ui < 24 / W; ui++) {
    const int u = wave + W * ui, unit0 = 16 * u + 4 * gq;
    // (instrumented build, layer 0, wave 0: shader-clock deltas inside a unit tile -> slots 1376 + 5 ui + {0: input gates, 1: their
    //  conversion, 2: recurrent gates, 3: wait + rows + conversion, 4: activations and stores}; tools/k1_cycles.py --layers)
#if RN_INSTRUMENT
    unsigned long long tc = (dbg && layer == 0) ? __builtin_amdgcn_s_memtime() : 0;
#define GRU_TAP(i) do { if (dbg && layer == 0 && ui < 3) {  /* (three unit tiles' worth of slots: the four-wave variants have six) */ const unsigned long long n_ = __builtin_amdgcn_s_memtime(); dbg[1376 - (RN_DBG_CLK2 + 7) + 5 * ui + (i)] = (float)(n_ - tc); tc = n_; } } while (0)
#else
#define GRU_TAP(i) do { } while (0)
#endif
    v4i acc[3][GM];
    v4f gi[3][GM], h_old[GM];
    // (the accumulators start from 128 * rowsum(w): acc_x86 = acc_mfma + 128 rowsum, nn_mfma.hip, without an add per value)
#pragma unroll
    for (int gate = 0; gate < 3; gate++) {
      const v4i rs = ldg<v4i>(wi.rowsum128, (unsigned)(gate * RN_GRU + unit0) * 4u);
#pragma unroll
      for (int t = 0; t < GM; t++) acc[gate][t] = rs;
    }
    const unsigned a0 = (unsigned)(u * KT * 64 + lane) * 16u;  // byte offset of this lane's first A fragment
    AFrags<AD> A;
#pragma unroll
    for (int step = 0; step < AD; step++) a_fetch(A, step, wi.wmf, wr.wmf, a0);
    int8_gates(acc, A, 0, wi.wmf, wr.wmf, a0, lane, L.xq);
    GRU_TAP(0);
#pragma unroll
    for (int gate = 0; gate < 3; gate++) {  // float(acc_x86)*scale + subias (src/nnet_arch.h:145-151)
      const unsigned row4 = (unsigned)(gate * RN_GRU + unit0) * 4u;  // byte offset of the 4 rows
      const v4f sc = ldg<v4f>(wi.scale, row4);
      const v4f sb = ldg<v4f>(wi.bias, row4);
      const v4i rs = ldg<v4i>(wr.rowsum128, row4);
#pragma unroll
      for (int t = 0; t < GM; t++) {
#pragma unroll
        for (int r = 0; r < 4; r++) gi[gate][t][r] = (float)acc[gate][t][r] * sc[r] + sb[r];
        acc[gate][t] = rs;
      }
    }
    GRU_TAP(1);
    int8_gates(acc, A, KT, wi.wmf, wr.wmf, a0, lane, L.hq);
    GRU_TAP(2);
    // all of this wave's loads have landed (the last A fragment was just used): its f32 rows for this tile are in LDS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (settle == 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_sleep 4" ::: "memory");
    else if (settle == 2) __builtin_amdgcn_s_barrier();
    else if (settle == 3) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int t = 0; t < GM; t++) h_old[t] = *reinterpret_cast<const v4f *>(&L.hrow[wave][row_buf(ui)][TS * t + n][4 * gq]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef RN_GRU_LAB  // (the row-buffer check: lab/nn_gru_lab.hip)
    if (CHK) {
#pragma unroll
      for (int t = 0; t < GM; t++) {
        const v4f want = ldg<v4f>(st, (unsigned)(sn[t] * RN_GRU + unit0) * 4u);
        bool bad = false, stale = true;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          bad |= __float_as_uint(want[r]) != __float_as_uint(h_old[t][r]);
          stale &= __float_as_uint(h_prev[t][r]) == __float_as_uint(h_old[t][r]);
        }
        atomicAdd(&rn_gru_race_log[2], 1u);
        if (bad) {
          const unsigned k = atomicAdd(&rn_gru_race_log[0], 1u);
          if (stale && ui > 0) atomicAdd(&rn_gru_race_log[1], 1u);
          if (k < 40) {
            unsigned *rec = rn_gru_race_log + 4 + 12 * k;
            rec[0] = blockIdx.x;
            rec[1] = (unsigned)wave | (unsigned)ui << 8 | (unsigned)t << 16 | (unsigned)lane << 24;
#pragma unroll
            for (int r = 0; r < 4; r++) {
              rec[2 + r] = __float_as_uint(h_old[t][r]);
              rec[6 + r] = __float_as_uint(want[r]);
            }
            rec[10] = __float_as_uint(h_prev[t][0]);
            rec[11] = __float_as_uint(h_prev[t][1]) ;
          }
        }
        h_prev[t] = h_old[t];
      }
    }
#endif
    v4f gr[3][GM];
#pragma unroll
    for (int gate = 0; gate < 3; gate++) {
      const unsigned row4 = (unsigned)(gate * RN_GRU + unit0) * 4u;  // byte offset of the 4 rows
      const v4f sc = ldg<v4f>(wr.scale, row4);
      const v4f sb = ldg<v4f>(wr.bias, row4);
      const v4f dg = ldg<v4f>(wr.diag, row4);
#pragma unroll
      for (int t = 0; t < GM; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          gr[gate][t][r] = (float)acc[gate][t][r] * sc[r] + sb[r];
          gr[gate][t][r] += dg[r] * h_old[t][r];  // src/nnet_arch.h:153-161
        }
    }
    // The next tile's rows start their HBM trip here, under the ~4k cycles of activation VALU work that load nothing:
    // vmcnt retires in order, so any load issued behind them (the constants above, the next A fragments) waits them out.
    __builtin_amdgcn_sched_barrier(0);
    GRU_TAP(3);
    if (u + W < 24) rows_fetch(ui + 1);
    __builtin_amdgcn_sched_barrier(0);
    // one tile's 4 rows at a time: eight sigmoid lookups in flight, then four tanh lookups (instrumented build,
    // $RNNOISE_AMD_GRU_ACT=0 at launch: element by element without the batching -- A/B runs)
#pragma unroll
    for (int t = 0; t < GM; t++) {
      v4f hn;
      if (batched_act && !PK) {
        // the same batching element by element: eight sigmoid lookups in flight, then four tanh lookups
        ActPre az[4], ar[4], ah[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          az[r] = sigmoid_pre_lut0(gi[0][t][r] + gr[0][t][r]);
          ar[r] = sigmoid_pre_lut0(gi[1][t][r] + gr[1][t][r]);
        }
        __builtin_amdgcn_sched_barrier(0);
        float z[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          z[r] = sigmoid_fin_k(az[r]);
          ah[r] = tanh_pre_lut0(gi[2][t][r] + gr[2][t][r] * sigmoid_fin_k(ar[r]));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; r++) hn[r] = z[r] * h_old[t][r] + (1 - z[r]) * tanh_fin_k(ah[r]);
      } else if (batched_act) {
        ActPre2 az[2], ar[2], ah[2];
#pragma unroll
        for (int p = 0; p < 2; p++) {
          const v2f gz = {gi[0][t][2 * p], gi[0][t][2 * p + 1]}, rz = {gr[0][t][2 * p], gr[0][t][2 * p + 1]};
          const v2f gg = {gi[1][t][2 * p], gi[1][t][2 * p + 1]}, rr = {gr[1][t][2 * p], gr[1][t][2 * p + 1]};
          // (the *_lut0 / *_k forms: the table index is the LDS address -- the table is this kernel's first LDS member, checked at
          //  the top -- and the reciprocal is rebuilt in two operations: 2 VALU instructions fewer per activation, the same integers)
          az[p] = sigmoid_pre2_lut0(gz + rz);
          ar[p] = sigmoid_pre2_lut0(gg + rr);
        }
        __builtin_amdgcn_sched_barrier(0);
        v2f z[2];
#pragma unroll
        for (int p = 0; p < 2; p++) {
          z[p] = sigmoid_fin2_k(az[p]);
          const v2f gh = {gi[2][t][2 * p], gi[2][t][2 * p + 1]}, rh = {gr[2][t][2 * p], gr[2][t][2 * p + 1]};
          ah[p] = tanh_pre2_lut0(gh + rh * sigmoid_fin2_k(ar[p]));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < 2; p++) {
          const v2f ho = {h_old[t][2 * p], h_old[t][2 * p + 1]};
          const v2f hv = z[p] * ho + (v2f{1.f, 1.f} - z[p]) * tanh_fin2_k(ah[p]);
          hn[2 * p] = hv.x;
          hn[2 * p + 1] = hv.y;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const float z = sigmoid_g(gi[0][t][r] + gr[0][t][r], lut);
          const float rg = sigmoid_g(gi[1][t][r] + gr[1][t][r], lut);
          const float hh = tanh_g(gi[2][t][r] + gr[2][t][r] * rg, lut);
          hn[r] = z * h_old[t][r] + (1 - z) * hh;
        }
      }
      if (livemask >> t & 1) {  // (live implies tile0 + t < n_tiles and its stream < N)
        stg<v4f>(st, (unsigned)(((tile0 + t) * TS + n) * RN_GRU + unit0) * 4u, hn);
        stg<int>(himg, (unsigned)((tile0 + t) * (KT * 64 * 16) + frag_off(n, unit0)), pack4_g(hn[0], hn[1], hn[2], hn[3]));
      }
#ifdef RN_GRU_LAB
#define RN_FOLD_PART 3
#include "lab/nn_gru_fold.inc"
#undef RN_FOLD_PART
#endif
    }
#ifdef RN_GRU_LAB
#define RN_FOLD_PART 4
#include "lab/nn_gru_fold.inc"
#undef RN_FOLD_PART
#else
    GRU_TAP(4);
#endif
#undef GRU_TAP
  }
#ifdef RN_GRU_LAB  // (the row-buffer check: lab/nn_gru_lab.hip)
  if (CHK) image_check(1);
#endif
#ifdef RN_GRU_LAB
#define RN_FOLD_PART 5
#include "lab/nn_gru_fold.inc"
#undef RN_FOLD_PART
#endif
  if (dbg && tile0 * TS < N) {
    const unsigned long long clk3 = __builtin_amdgcn_s_memtime();
#ifdef RN_GRU_LAB
#define RN_FOLD_PART 7
#include "lab/nn_gru_fold.inc"
#undef RN_FOLD_PART
#endif
    dbg[0] = (float)(clk1 - clk0);
    dbg[1] = (float)(clk2 - clk1);
    dbg[2] = (float)(clk3 - clk2);
  }
}


// ---- host side: a form of the layer kernel, and its LDS opt-in ----
#include <atomic>
typedef void (*RnGruKernel)(RnGroupDev, RnModelDev, RnTablesDev, int);
struct RnGruVariant {
  const char *name;
  RnGruKernel k;
  int threads;
  size_t lds;
  bool persist;  // (lab forms) one workgroup per CU walking over the groups
  bool fold;     // the output chains run inside the layer launches (gru_body FOLD): the front kernel starts them, no dense kernel
};
// more than 64 KB of dynamic LDS is an opt-in per kernel and device; remembered per (kernel, device) -- at most 64 kernels x 64 devices
// (the product has two kernels).  Concurrent first launches may both set the attribute: idempotent.
static inline hipError_t rn_gru_opt_in(const RnGruVariant &v, int dev) {
  static std::atomic<const void *> seen[64][64];
  const void *key = reinterpret_cast<const void *>(v.k);
  for (int i = 0; i < 64; i++) {
    const void *have = seen[dev][i].load(std::memory_order_acquire);
    if (have == key) return hipSuccess;
    if (!have) {
      const hipError_t e = hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v.lds);
      if (e != hipSuccess) return e;
      const void *expect = nullptr;
      if (seen[dev][i].compare_exchange_strong(expect, key, std::memory_order_acq_rel) || expect == key) return hipSuccess;
      // (another thread took the slot for another kernel: keep looking)
    }
  }
  return hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v.lds);  // table full: set it every time
}
