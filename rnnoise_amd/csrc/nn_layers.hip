// nn_layers.hip -- K2 for large batches (from 16,384 streams up, batch.cpp: nn_layers_min_streams): the network layer by layer.
//
//   rn_nn_front_kernel (nn_mfma.hip)  conv1, conv2 per 16-stream tile; leaves the u8 image of the conv2 output in act_q[0]
//   rn_nn_gru_kernel    x 3           one GRU layer (src/nnet.c:65-94) for 64 streams per workgroup
//   rn_nn_dense_kernel                dense_out + vad_dense (src/rnn.c:53-58) for 64 streams per workgroup
//   rn_nn_requant_kernel              rebuilds the state images after anything else wrote the GRU state
//
// The fused tile kernel (nn_mfma.hip) walks all seven int8 matrices for 16 streams: every weight fragment a wave
// fetches from L2 feeds ONE MFMA, and its phases are latency-bound (rocprof: 61-66 % of wave time waiting).  Here a GRU
// workgroup holds the quantised inputs of GM = 4 tiles in LDS and works on 3 gates x 4 tiles per weight fragment.  A
// layer's input arrives as the B-fragment image the previous launch left in HBM (RnGroupDev::act_q), its output -- at
// once the next layer's input and this layer's recurrent operand of the next frame -- leaves the same way.
// Arithmetic per element is the fused kernel's, so the bits are too (tests/test_gpu_parity.py runs both; tools/ab_layers.py).
#include "nn_common.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define GM 4  // 16-stream tiles per workgroup
#ifndef GW
#define GW 8  // waves per workgroup
#endif
#define GTHREADS (64 * GW)

// W: waves per GRU workgroup, HB: f32 row buffers per wave.  The shipping kernel is <8, 3>: one buffer per unit tile of a wave,
// every buffer written once per launch.  <4, 1> (one wave per SIMD, 72 KB: two workgroups per CU, or one beside analysis
// workgroups of the next frame) is the round-4 variant that was not bit-stable in the pipelined schedule; it and the other
// instantiations below are kept as A/B variants ($RNNOISE_AMD_GRU_VARIANT; profiles/r5_gru_race.txt names what went wrong).
template <int W, int HB>
struct GruLdsT {
  uint16_t lut[4096];            // rcpps table (rn_dev.h: rcp16)
  int8_t xq[GM][KT * 64 * 16];   // layer input images
  int8_t hq[GM][KT * 64 * 16];   // recurrent state images
  float hrow[W][HB][GM * TS][16];  // per wave (and unit tile, HB == 24 / W): the f32 state of its 16 units for the workgroup's
                                   // 64 streams (the blend z*h + (1-z)*candidate needs them exact)
};
static_assert(sizeof(GruLdsT<8, 3>) <= 160 * 1024, "one workgroup per CU, all of its LDS");
static_assert(sizeof(GruLdsT<4, 1>) <= 80 * 1024, "two workgroups per CU");

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

#if RN_INSTRUMENT
// CHK instantiations (tools/gru_race.py): every h_old vector a lane takes from its LDS row buffer is compared with the same 16
// bytes loaded straight from HBM.  [0] = mismatching vectors seen, [1] = of them equal to the PREVIOUS unit tile's vector of that
// lane (stale buffer: the LDS-DMA had not landed), [2] = vectors checked (low 32 bits); then up to 40 records of 12 words:
// block | wave, ui, t, lane | got[4] | want[4] | previous tile's[2].  [3] = words of the LDS images (layer input, recurrent state,
// rcpps table) that differed from HBM behind the prologue's barrier or (input image, table) at the end of the kernel; their
// records: block | 0xffff0000 + 0x100 * (0 start, 1 end) + region (0..3 xq, 4..7 hq, 8 table) | word | got | want | wave
__device__ unsigned rn_gru_race_log[4 + 40 * 12];
#endif
// layer_arg: bits 0-1 layer; bit 2: activations element by element (A/B); bits 3-4: what stands between the wait for this wave's
// row DMA and its reads of those rows, beyond s_waitcnt vmcnt(0) -- 0 nothing, 1 = lgkmcnt(0) + s_sleep (256 clocks), 2 = a
// workgroup barrier, 3 = lgkmcnt(0) + buffer_inv-free s_nop ladder (A/B runs of the race hunt only)
template <int AD, int W, int HB, bool CHK, bool DMA = true>
__device__ __forceinline__ void gru_body(const RnGroupDev &g, const RnModelDev &m, const RnTablesDev &tb, int layer_arg) {
  typedef GruLdsT<W, HB> GruLds;
  static_assert(24 % W == 0 && (HB == 24 / W || HB <= 2), "unit tiles per wave; row buffers");
  const int layer = layer_arg & 3;
  const bool batched_act = !(layer_arg & 4);
  const int settle = (layer_arg >> 3) & 3;
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
  bool live[GM];
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
#pragma unroll
    for (int i = 0; i < GM * TS * 16 * 4 / 1024; i++) {
      const int idx = i * 64 + lane, row = idx >> 2, seg = idx & 3, s = tile0 * TS + row, sc = s < N ? s : N - 1;
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const unsigned long long clk1 = (RN_INSTRUMENT && g.debug) ? __builtin_amdgcn_s_memtime() : 0;
  __builtin_amdgcn_s_barrier();
  const unsigned long long clk2 = (RN_INSTRUMENT && g.debug) ? __builtin_amdgcn_s_memtime() : 0;
#if RN_INSTRUMENT
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
  for (int t = 0; t < GM; t++) live[t] = (tile0 + t) * TS + n < N && !sil[t];  // silent streams keep their state (src/denoise.c:474)
  const RnLinearDev &wi = m.gru_in[layer], &wr = m.gru_rec[layer];
  // The two waves of a SIMD run the same phases from the same barrier: left alone they want the MFMA pipe together and
  // the VALU together.  Giving one of them issue priority lets it run ahead, after which one's MFMA block overlaps the
  // other's epilogue.
  if (wave < W / 2 && !(layer_arg & 32)) __builtin_amdgcn_s_setprio(2);  // (bit 5: $RNNOISE_AMD_GRU_PRIO=0, A/B runs)
  [[maybe_unused]] v4f h_prev[GM] = {};
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
#if RN_INSTRUMENT
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
    // one tile's 4 rows at a time: eight sigmoid lookups in flight, then four tanh lookups ($RNNOISE_AMD_GRU_ACT=0 at launch:
    // element by element, as before -- A/B runs)
#pragma unroll
    for (int t = 0; t < GM; t++) {
      v4f hn;
      if (batched_act) {
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
      if (live[t]) {  // (live implies tile0 + t < n_tiles)
        stg<v4f>(st, (unsigned)(sn[t] * RN_GRU + unit0) * 4u, hn);
        stg<int>(himg, (unsigned)((tile0 + t) * (KT * 64 * 16) + frag_off(n, unit0)), pack4_g(hn[0], hn[1], hn[2], hn[3]));
      }
    }
    GRU_TAP(4);
#undef GRU_TAP
  }
#if RN_INSTRUMENT
  if (CHK) image_check(1);
#endif
  if (dbg && tile0 * TS < N) {
    const unsigned long long clk3 = __builtin_amdgcn_s_memtime();
    dbg[0] = (float)(clk1 - clk0);
    dbg[1] = (float)(clk2 - clk1);
    dbg[2] = (float)(clk3 - clk2);
  }
}

// ---- round 5: the same layer, three changes that can be switched one by one (template bits; A/B by $RNNOISE_AMD_GRU_VARIANT) ----
//   GRU_BD       the B fragments (LDS) of k-step k + 1 are requested BEFORE the MFMAs of k-step k.  The loop above reads a k-step's
//                four fragments and then needs them at once: an exposed LDS round trip per k-step -- the shader-clock taps put a
//                72-MFMA block (1,152 cycles of matrix pipe) at 2.7-3.1 k cycles.
//   GRU_DEEP     the rcpps lookups of ALL four tiles' z and r gates are in flight together, then all four tiles' candidates:
//                two dependent LDS round trips per unit tile instead of eight.
//   GRU_PERSIST  a workgroup walks over groups blockIdx.x, + gridDim.x, ... (one workgroup per CU): the two images of the NEXT group
//                arrive by LDS-DMA in a second pair of buffers under this group's arithmetic, the rcpps table is fetched once,
//                and the ~8 k cycles of prologue in which a CU does nothing else are paid once per launch instead of once per group.
//   GRU_AX       the first two A fragments (L2) and the 128 * rowsum vectors of the NEXT unit tile are requested before this unit
//                tile's activation stretch (the accumulators are dead there: the registers are free) instead of at its top.
// Arithmetic per element, and therefore every bit, as above (tests/test_gpu_parity.py runs the variants against each other).
#define GRU_BD 1
#define GRU_DEEP 2
#define GRU_PERSIST 4
#define GRU_AX 8
template <int NIMG>
struct GruLds2T {
  uint16_t lut[4096];
  int8_t xq[NIMG][GM][KT * 64 * 16];
  int8_t hq[NIMG][GM][KT * 64 * 16];
  float hrow[8][GM * TS][16];  // one row buffer per wave: refilled for the next unit tile once this one's rows are in registers
};
static_assert(sizeof(GruLds2T<2>) <= 160 * 1024, "persistent workgroup: one per CU");

template <bool BD>
__device__ __forceinline__ void b_fetch(v4i (&bf)[2][GM], int slot, const int8_t (*bq)[KT * 64 * 16], int kt, int lane) {
#pragma unroll
  for (int t = 0; t < GM; t++) bf[BD ? slot : 0][t] = reinterpret_cast<const v4i *>(bq[t])[kt * 64 + lane];
}
// k-steps [s0, s0 + KT): acc[gate][t] += A(step)[gate] . image[t].  BD: bf[0] holds k-step 0's fragments on entry; on exit bf[0]
// holds the first fragments of bq_next (if any)
template <int AD, bool BD>
__device__ __forceinline__ void int8_gates2(v4i acc[3][GM], AFrags<AD> &A, v4i (&bf)[2][GM], int s0, const int8_t *__restrict__ wi,
                                            const int8_t *__restrict__ wr, unsigned a0, int lane, const int8_t (*bq)[KT * 64 * 16],
                                            const int8_t (*bq_next)[KT * 64 * 16]) {
  static_assert(KT % 2 == 0, "the fragment slots alternate: an even number of k-steps per matrix");
  asm volatile("" : "+v"(lane));  // (keep the compiler from hoisting all 48 B fragments)
#pragma unroll
  for (int kt = 0; kt < KT; kt++) {
    const int step = s0 + kt;
    if (step + AD < 2 * KT) a_fetch(A, step + AD, wi, wr, a0);
    __builtin_amdgcn_sched_barrier(0);
    if (!BD) b_fetch<false>(bf, 0, bq, kt, lane);
    else if (kt + 1 < KT) b_fetch<true>(bf, (kt + 1) & 1, bq, kt + 1, lane);
    else if (bq_next) b_fetch<true>(bf, 0, bq_next, 0, lane);
#pragma unroll
    for (int gate = 0; gate < 3; gate++)
#pragma unroll
      for (int t = 0; t < GM; t++)
        acc[gate][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A.f[step % (AD + 1)][gate], bf[BD ? (kt & 1) : 0][t], acc[gate][t], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int OPT>
__device__ __forceinline__ void gru_body2(const RnGroupDev &g, const RnModelDev &m, const RnTablesDev &tb, int layer_arg) {
  constexpr bool BD = OPT & GRU_BD, DEEP = OPT & GRU_DEEP, PERSIST = OPT & GRU_PERSIST, AX = OPT & GRU_AX;
  constexpr int W = 8, AD = 2, UT = 24 / W;
  typedef GruLds2T<PERSIST ? 2 : 1> GruLds;
  const int layer = layer_arg & 3;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  GruLds &L = *reinterpret_cast<GruLds *>(lds_raw);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 15, gq = lane >> 4;
  const int N = g.n_streams, n_tiles = (N + TS - 1) / TS, n_groups = (n_tiles + GM - 1) / GM;
  const uint16_t *lut = L.lut;
  float *st = g.gru_state + (size_t)layer * g.n_stride * RN_GRU;
  const int8_t *xin = g.act_q[layer];
  int8_t *himg = g.act_q[layer + 1];
  const RnLinearDev &wi = m.gru_in[layer], &wr = m.gru_rec[layer];
#if RN_INSTRUMENT
  float *dbg = (g.debug && tid == 0 && blockIdx.x * GM * TS < N) ? g.debug + (size_t)blockIdx.x * GM * TS * RN_DBG_FLOATS + RN_DBG_CLK2 + 7 + 3 * layer : nullptr;
#else
  float *const dbg = nullptr;
#endif
  const unsigned long long clk0 = (RN_INSTRUMENT && g.debug) ? __builtin_amdgcn_s_memtime() : 0;
  // (instrumented build, $RNNOISE_AMD_GRU_TIMELINE=1 -> bit 6 of layer_arg, layer 0, a workgroup's first group: EVERY wave's shader
  //  clock at the six boundaries of its unit tiles -- row 1 + wave of the workgroup's debug block, words 6 ui .. 6 ui + 5 as raw
  //  low 32 bits, word 18 = the wave's clock at kernel entry, 19 = behind the prologue's barrier; words 20 .. 37: the same for the
  //  workgroup's second group (persistent variants); tools/gru_timeline.py)
#if RN_INSTRUMENT
  unsigned *tl = (g.debug && (layer_arg & 64) && layer == 0 && lane == 0 && (blockIdx.x * GM * TS + 1 + wave) < N)
                     ? reinterpret_cast<unsigned *>(g.debug + (size_t)(blockIdx.x * GM * TS + 1 + wave) * RN_DBG_FLOATS) : nullptr;
  if (tl) tl[18] = (unsigned)clk0;
#define GRU_TL(i) do { if (tl && it < 2) tl[20 * it + 6 * ui + (i)] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define GRU_TL(i) do { } while (0)
#endif

  // (address arithmetic of the fetches is redone at every call from an opaque copy of the lane number: hoisted out of the group
  //  loop, the per-piece offsets would sit in registers through the MFMA blocks -- the persistent variants spilled 20-80 dwords)
  auto opaque_lane = [&] {
    int l = lane;
    asm volatile("" : "+v"(l));
    return l;
  };
  // the two images of a group's GM tiles: 48 pieces of 1 KB, six per wave, HBM -> LDS without staging registers
  auto images_fetch = [&](int grp, int ib) {
    const int l = opaque_lane();
#pragma unroll
    for (int j = 0; j < 2 * GM * KT / W; j++) {
      const int c = wave + j * W, which = c / (GM * KT), cc = c - which * (GM * KT), t = cc / KT, kt = cc - t * KT;
      const int tile = (grp * GM + t < n_tiles) ? grp * GM + t : n_tiles - 1;
      dma_1k((which ? himg : xin) + ((size_t)tile * (KT * 64 * 16) + kt * 1024 + l * 16),
             lds_addr(which ? L.hq[ib][t] : L.xq[ib][t]) + kt * 1024);
    }
  };
  // f32 state of units 16 u .. 16 u + 15, u = wave + W ui, of the group's 64 streams: 4 pieces into this wave's row buffer
  auto rows_fetch = [&](int grp, int ui) {
    const int u = wave + W * ui, l = opaque_lane();
#pragma unroll
    for (int i = 0; i < GM * TS * 16 * 4 / 1024; i++) {
      const int idx = i * 64 + l, row = idx >> 2, seg = idx & 3, s = grp * GM * TS + row, sc = s < N ? s : N - 1;
      dma_1k(st + ((size_t)sc * RN_GRU + 16 * u + 4 * seg), lds_addr(&L.hrow[wave][0][0]) + i * 1024);
    }
  };
  auto a_offset = [&](int ui) { return (unsigned)((wave + W * ui) * KT * 64 + lane) * 16u; };
  AFrags<AD> A;
  v4i rs_in[3];
  auto tile_heads_fetch = [&](int ui) {  // what a unit tile needs first: 128 * rowsum of its 3 x 4 input-matrix rows, A fragments 0 .. AD - 1
    const int unit0 = 16 * (wave + W * ui) + 4 * gq;
#pragma unroll
    for (int gate = 0; gate < 3; gate++) rs_in[gate] = ldg<v4i>(wi.rowsum128, (unsigned)(gate * RN_GRU + unit0) * 4u);
#pragma unroll
    for (int step = 0; step < AD; step++) a_fetch(A, step, wi.wmf, wr.wmf, a_offset(ui));
  };

  int grp = blockIdx.x;
  images_fetch(grp, 0);
  dma_1k(reinterpret_cast<const uint32_t *>(tb.rcp16) + wave * 256 + lane * 4, lds_addr(L.lut) + wave * 1024);  // 8 pieces, W == 8
  rows_fetch(grp, 0);
  if (AX) tile_heads_fetch(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long clk1 = (RN_INSTRUMENT && g.debug) ? __builtin_amdgcn_s_memtime() : 0;
  __builtin_amdgcn_s_barrier();
  const unsigned long long clk2 = (RN_INSTRUMENT && g.debug) ? __builtin_amdgcn_s_memtime() : 0;
#if RN_INSTRUMENT
  if (tl) tl[19] = (unsigned)clk2;
#endif
  if (wave < W / 2 && !(layer_arg & 32)) __builtin_amdgcn_s_setprio(2);  // (see gru_body; bit 5: $RNNOISE_AMD_GRU_PRIO=0)

#pragma unroll 1
  for (int it = 0;; it++) {
    const int ib = PERSIST ? (it & 1) : 0, tile0 = grp * GM;
    const int next_grp = grp + (int)gridDim.x;
    const bool has_next = PERSIST && next_grp < n_groups;
    // (one register per group across the unit tiles: bit t = stream (tile0 + t) * 16 + n exists and is not silent)
    unsigned livemask = 0;
#pragma unroll
    for (int t = 0; t < GM; t++) {
      const int s = (tile0 + t) * TS + n, sc = s < N ? s : N - 1;
      livemask |= (s < N && !g.silence[(unsigned)sc]) ? 1u << t : 0u;  // silent streams keep their state (src/denoise.c:474)
    }
#pragma unroll 1
    for (int ui = 0; ui < UT; ui++) {
      const int u = wave + W * ui, unit0 = 16 * u + 4 * gq;
#if RN_INSTRUMENT
      unsigned long long tc = (dbg && layer == 0 && it == 0) ? __builtin_amdgcn_s_memtime() : 0;
#define GRU_TAP(i) do { if (dbg && layer == 0 && it == 0) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); dbg[1376 - (RN_DBG_CLK2 + 7) + 5 * ui + (i)] = (float)(n_ - tc); tc = n_; } } while (0)
#else
#define GRU_TAP(i) do { } while (0)
#endif
      GRU_TL(0);
      v4i acc[3][GM], bf[2][GM];
      v4f gi[3][GM], h_old[GM];
      const unsigned a0 = a_offset(ui);
      if (!AX) tile_heads_fetch(ui);
#pragma unroll
      for (int gate = 0; gate < 3; gate++)
#pragma unroll
        for (int t = 0; t < GM; t++) acc[gate][t] = rs_in[gate];  // (acc_x86 = acc_mfma + 128 rowsum(w))
      if (BD) b_fetch<true>(bf, 0, L.xq[ib], 0, lane);
      int8_gates2<AD, BD>(acc, A, bf, 0, wi.wmf, wr.wmf, a0, lane, L.xq[ib], L.hq[ib]);
      GRU_TAP(0);
      GRU_TL(1);
#pragma unroll
      for (int gate = 0; gate < 3; gate++) {  // float(acc_x86)*scale + subias (src/nnet_arch.h:145-151)
        const unsigned row4 = (unsigned)(gate * RN_GRU + unit0) * 4u;
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
      GRU_TL(2);
      int8_gates2<AD, BD>(acc, A, bf, KT, wi.wmf, wr.wmf, a0, lane, L.hq[ib], nullptr);
      GRU_TAP(2);
      GRU_TL(3);
      // all of this wave's loads have landed (the last A fragment was just used): its f32 rows for this tile are in LDS
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int t = 0; t < GM; t++) h_old[t] = *reinterpret_cast<const v4f *>(&L.hrow[wave][TS * t + n][4 * gq]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      v4f gr[3][GM];
#pragma unroll
      for (int gate = 0; gate < 3; gate++) {
        const unsigned row4 = (unsigned)(gate * RN_GRU + unit0) * 4u;
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
      // What the NEXT unit tile (of this group, or the first of the next group) needs starts its trip here, under the activation
      // stretch that loads nothing: vmcnt retires in order, so any load issued behind these waits them out.
      __builtin_amdgcn_sched_barrier(0);
      GRU_TAP(3);
      GRU_TL(4);
      const bool more = ui + 1 < UT;
      if (AX && (more || has_next)) tile_heads_fetch(more ? ui + 1 : 0);
      if (more) rows_fetch(grp, ui + 1);
      else if (has_next) rows_fetch(next_grp, 0);
      if (PERSIST && ui == 0 && has_next) images_fetch(next_grp, ib ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      auto store_tile = [&](int t, const v4f &hn) {
        if (livemask >> t & 1) {  // (live implies tile0 + t < n_tiles and its stream < N)
          stg<v4f>(st, (unsigned)(((tile0 + t) * TS + n) * RN_GRU + unit0) * 4u, hn);
          stg<int>(himg, (unsigned)((tile0 + t) * (KT * 64 * 16) + frag_off(n, unit0)), pack4_g(hn[0], hn[1], hn[2], hn[3]));
        }
      };
      if (DEEP) {
        ActPre2 az[GM][2], ar[GM][2], ah[GM][2];
        v2f z[GM][2];
#pragma unroll
        for (int t = 0; t < GM; t++)
#pragma unroll
          for (int p = 0; p < 2; p++) {
            const v2f gz = {gi[0][t][2 * p], gi[0][t][2 * p + 1]}, rz = {gr[0][t][2 * p], gr[0][t][2 * p + 1]};
            const v2f gg = {gi[1][t][2 * p], gi[1][t][2 * p + 1]}, rr = {gr[1][t][2 * p], gr[1][t][2 * p + 1]};
            az[t][p] = sigmoid_pre2(gz + rz, lut);
            ar[t][p] = sigmoid_pre2(gg + rr, lut);
          }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < GM; t++)
#pragma unroll
          for (int p = 0; p < 2; p++) {
            z[t][p] = sigmoid_fin2(az[t][p]);
            const v2f gh = {gi[2][t][2 * p], gi[2][t][2 * p + 1]}, rh = {gr[2][t][2 * p], gr[2][t][2 * p + 1]};
            ah[t][p] = tanh_pre2(gh + rh * sigmoid_fin2(ar[t][p]), lut);
          }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < GM; t++) {
          v4f hn;
#pragma unroll
          for (int p = 0; p < 2; p++) {
            const v2f ho = {h_old[t][2 * p], h_old[t][2 * p + 1]};
            const v2f hv = z[t][p] * ho + (v2f{1.f, 1.f} - z[t][p]) * tanh_fin2(ah[t][p]);
            hn[2 * p] = hv.x;
            hn[2 * p + 1] = hv.y;
          }
          store_tile(t, hn);
        }
      } else {
#pragma unroll
        for (int t = 0; t < GM; t++) {
          v4f hn;
          ActPre2 az[2], ar[2], ah[2];
#pragma unroll
          for (int p = 0; p < 2; p++) {
            const v2f gz = {gi[0][t][2 * p], gi[0][t][2 * p + 1]}, rz = {gr[0][t][2 * p], gr[0][t][2 * p + 1]};
            const v2f gg = {gi[1][t][2 * p], gi[1][t][2 * p + 1]}, rr = {gr[1][t][2 * p], gr[1][t][2 * p + 1]};
            az[p] = sigmoid_pre2(gz + rz, lut);
            ar[p] = sigmoid_pre2(gg + rr, lut);
          }
          __builtin_amdgcn_sched_barrier(0);
          v2f z[2];
#pragma unroll
          for (int p = 0; p < 2; p++) {
            z[p] = sigmoid_fin2(az[p]);
            const v2f gh = {gi[2][t][2 * p], gi[2][t][2 * p + 1]}, rh = {gr[2][t][2 * p], gr[2][t][2 * p + 1]};
            ah[p] = tanh_pre2(gh + rh * sigmoid_fin2(ar[p]), lut);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int p = 0; p < 2; p++) {
            const v2f ho = {h_old[t][2 * p], h_old[t][2 * p + 1]};
            const v2f hv = z[p] * ho + (v2f{1.f, 1.f} - z[p]) * tanh_fin2(ah[p]);
            hn[2 * p] = hv.x;
            hn[2 * p + 1] = hv.y;
          }
          store_tile(t, hn);
        }
      }
      GRU_TAP(4);
      GRU_TL(5);
#undef GRU_TAP
    }
    if (!has_next) break;
    grp = next_grp;
    // Every wave's pieces of the next group's images were issued in its unit tile 0 and drained by the vmcnt(0) of its unit tiles
    // 1 and 2: behind this barrier they are all in LDS, and nobody reads this group's images any more (the group after next
    // refills them from the next group's unit tile 0 on).
    __builtin_amdgcn_s_barrier();
  }
  if (dbg) {
    const unsigned long long clk3 = __builtin_amdgcn_s_memtime();
    dbg[0] = (float)(clk1 - clk0);
    dbg[1] = (float)(clk2 - clk1);
    dbg[2] = (float)(clk3 - clk2);
  }
}
#undef GRU_TL
#define GRU2_KERNEL(name, opt)                                                                                          \
  extern "C" __global__ void __launch_bounds__(512) name(RnGroupDev g, RnModelDev m, RnTablesDev tb, int layer) {    \
    gru_body2<opt>(g, m, tb, layer);                                                                                    \
  }
GRU2_KERNEL(rn_nn_gru2_o0_kernel, 0)
GRU2_KERNEL(rn_nn_gru2_bd_kernel, GRU_BD)
GRU2_KERNEL(rn_nn_gru2_deep_kernel, GRU_DEEP)
GRU2_KERNEL(rn_nn_gru2_ax_kernel, GRU_AX)
GRU2_KERNEL(rn_nn_gru2_bdx_kernel, GRU_BD | GRU_DEEP | GRU_AX)
GRU2_KERNEL(rn_nn_gru2_p_kernel, GRU_PERSIST)
GRU2_KERNEL(rn_nn_gru2_pbd_kernel, GRU_PERSIST | GRU_BD)
GRU2_KERNEL(rn_nn_gru2_pall_kernel, GRU_PERSIST | GRU_BD | GRU_DEEP | GRU_AX)
GRU2_KERNEL(rn_nn_gru2_pbdx_kernel, GRU_PERSIST | GRU_BD | GRU_AX)


// ---- round 5, second step: THREE waves per SIMD --------------------------------------------------------------------------------
// What the timelines (tools/gru_timeline.py, profiles/r5_gru_timeline_o0_p.txt) say about the kernels above: a 64-stream group costs
// a CU ~66 k cycles -- 16 k of prologue in which it does nothing else, then three unit tiles per wave of 13-16 k each, of which the
// activation stretch (~800 VALU instructions) takes 7-9 k: ONE instruction per 9-11 cycles.  A wave alone on its SIMD's VALU issues
// at most one instruction per ~5 cycles, and its only partner is in its MFMA block (whose issue comes first).  The VALU pipe could
// take an instruction every 2.2-4.1 cycles from two waves; with 230 VGPRs per wave there is no third wave to offer them.
// Here the register block of a unit tile is split by GATES: first the update and reset gates (2 gates x 4 tiles: 32 accumulators,
// 32 converted input sums), then the candidate gate (16 + 16) with z, r and h_old (48) live -- ~135 registers at the peak instead
// of ~215, so a workgroup is TWELVE waves, two unit tiles each, three per SIMD: while one is in an MFMA block two can share the
// VALU.  Price: the B fragments (LDS) of a unit tile are read twice, 96 KB instead of 48 -- the LDS port has the room (it was 18 %
// busy); the A fragments (L2 -> L1 at 64 B per clock and CU, the scarcer path) still feed four MFMAs each.
// Persistent like GRU_PERSIST above (8 + 96 + 48 KB of LDS: lut, two image pairs, one row buffer per wave).
// Bits of OPT: GRU_BD, GRU_PERSIST, GRU3_MPRIO (a wave raises its issue priority for its MFMA blocks: the matrix pipe then never
// waits behind a partner's VALU stream).
#define GRU3_MPRIO 16
#define GRU3_NOMFMA 32  // timing experiments (wrong results): the MFMA instructions / the activation arithmetic left out
#define GRU3_NOACT 64
#define GRU3_HITA 128   // ... every A-fragment fetch an L1 hit (the same fragment again)
#define G3W 12
template <int NIMG>
struct GruLds3T {
  uint16_t lut[4096];
  int8_t xq[NIMG][GM][KT * 64 * 16];
  int8_t hq[NIMG][GM][KT * 64 * 16];
  float hrow[G3W][GM * TS][16];
};
static_assert(sizeof(GruLds3T<2>) <= 160 * 1024, "persistent 12-wave workgroup: one per CU");

template <int AD, int NG>
struct AFragsG {
  v4i f[AD + 1][NG];
};
// A fragments of gates G0 .. G0 + NG - 1 of unit-tile row u (a0 = its lane's byte offset), k-step `step` of the rolling sequence
template <int AD, int NG, int G0, bool HITA = false>
__device__ __forceinline__ void a_fetch_g(AFragsG<AD, NG> &A, int step, const int8_t *__restrict__ wi, const int8_t *__restrict__ wr, unsigned a0) {
  const int8_t *a = step < KT ? wi : wr;
  const int kt = HITA ? 0 : (step < KT ? step : step - KT);  // (HITA, timing experiment: every fetch re-reads k-step 0's fragment -- an L1 hit)
#pragma unroll
  for (int gi_ = 0; gi_ < NG; gi_++) A.f[step % (AD + 1)][gi_] = ldg<v4i>(HITA ? wi : a, a0 + (unsigned)(((G0 + gi_) * 24 * KT + kt) * 1024));
}
template <int AD, int NG, int G0, bool BD, bool NOMFMA = false, bool HITA = false>
__device__ __forceinline__ void int8_gates_g(v4i (&acc)[NG][GM], AFragsG<AD, NG> &A, v4i (&bf)[2][GM], int s0, const int8_t *__restrict__ wi,
                                             const int8_t *__restrict__ wr, unsigned a0, int lane, const int8_t (*bq)[KT * 64 * 16],
                                             const int8_t (*bq_next)[KT * 64 * 16]) {
  asm volatile("" : "+v"(lane));
#pragma unroll
  for (int kt = 0; kt < KT; kt++) {
    const int step = s0 + kt;
    if (step + AD < 2 * KT) a_fetch_g<AD, NG, G0, HITA>(A, step + AD, wi, wr, a0);
    __builtin_amdgcn_sched_barrier(0);
    if (!BD) b_fetch<false>(bf, 0, bq, kt, lane);
    else if (kt + 1 < KT) b_fetch<true>(bf, (kt + 1) & 1, bq, kt + 1, lane);
    else if (bq_next) b_fetch<true>(bf, 0, bq_next, 0, lane);
#pragma unroll
    for (int gi_ = 0; gi_ < NG; gi_++)
#pragma unroll
      for (int t = 0; t < GM; t++) {
        if (!NOMFMA) acc[gi_][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A.f[step % (AD + 1)][gi_], bf[BD ? (kt & 1) : 0][t], acc[gi_][t], 0, 0, 0);
        else asm volatile("" : "+v"(acc[gi_][t]) : "v"(A.f[step % (AD + 1)][gi_]), "v"(bf[BD ? (kt & 1) : 0][t]));  // (timing experiment: operands fetched, no MFMA)
      }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int OPT>
__device__ __forceinline__ void gru_body3(const RnGroupDev &g, const RnModelDev &m, const RnTablesDev &tb, int layer_arg) {
  constexpr bool BD = OPT & GRU_BD, PERSIST = OPT & GRU_PERSIST, MPRIO = OPT & GRU3_MPRIO, NOMFMA = OPT & GRU3_NOMFMA, NOACT = OPT & GRU3_NOACT, HITA = OPT & GRU3_HITA;
  constexpr int W = G3W, AD = 2, UT = 24 / W;
  typedef GruLds3T<PERSIST ? 2 : 1> GruLds;
  const int layer = layer_arg & 3;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  GruLds &L = *reinterpret_cast<GruLds *>(lds_raw);
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, n = lane & 15, gq = lane >> 4;
  const int N = g.n_streams, n_tiles = (N + TS - 1) / TS, n_groups = (n_tiles + GM - 1) / GM;
  if (lds_addr(L.lut) != 0) __builtin_trap();  // (the *_lut0 activations take the table index for its LDS address)
  float *st = g.gru_state + (size_t)layer * g.n_stride * RN_GRU;
  const int8_t *xin = g.act_q[layer];
  int8_t *himg = g.act_q[layer + 1];
  const RnLinearDev &wi = m.gru_in[layer], &wr = m.gru_rec[layer];
#if RN_INSTRUMENT
  // (timeline taps as in gru_body2: row 1 + wave of the workgroup's debug block; words 10 ui + {0 start, 1 z/r input block, 2 its conversion,
  //  3 z/r recurrent block, 4 rows + conversion + sigmoids, 5 candidate input block, 6 candidate recurrent block, 7 tanh + blend + stores};
  //  38 = entry, 39 = behind the prologue's barrier; the workgroup's second group: + 40)
  unsigned *tl = (g.debug && (layer_arg & 64) && layer == 0 && lane == 0 && (blockIdx.x * GM * TS + 1 + wave) < N)
                     ? reinterpret_cast<unsigned *>(g.debug + (size_t)(blockIdx.x * GM * TS + 1 + wave) * RN_DBG_FLOATS) : nullptr;
  if (tl) tl[38] = (unsigned)__builtin_amdgcn_s_memtime();
#define GRU_TL(i) do { if (tl && it < 2) tl[40 * it + 10 * ui + (i)] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define GRU_TL(i) do { } while (0)
#endif
  auto opaque_lane = [&] {
    int l = lane;
    asm volatile("" : "+v"(l));
    return l;
  };
  auto images_fetch = [&](int grp, int ib) {  // 48 pieces of 1 KB over 12 waves
    const int l = opaque_lane();
#pragma unroll
    for (int j = 0; j < 2 * GM * KT / W; j++) {
      const int c = wave + j * W, which = c / (GM * KT), cc = c - which * (GM * KT), t = cc / KT, kt = cc - t * KT;
      const int tile = (grp * GM + t < n_tiles) ? grp * GM + t : n_tiles - 1;
      dma_1k((which ? himg : xin) + ((size_t)tile * (KT * 64 * 16) + kt * 1024 + l * 16),
             lds_addr(which ? L.hq[ib][t] : L.xq[ib][t]) + kt * 1024);
    }
  };
  auto rows_fetch = [&](int grp, int ui) {
    const int u = wave + W * ui, l = opaque_lane();
#pragma unroll
    for (int i = 0; i < GM * TS * 16 * 4 / 1024; i++) {
      const int idx = i * 64 + l, row = idx >> 2, seg = idx & 3, s = grp * GM * TS + row, sc = s < N ? s : N - 1;
      dma_1k(st + ((size_t)sc * RN_GRU + 16 * u + 4 * seg), lds_addr(&L.hrow[wave][0][0]) + i * 1024);
    }
  };

  int grp = blockIdx.x;
  images_fetch(grp, 0);
  if (wave < 8) dma_1k(reinterpret_cast<const uint32_t *>(tb.rcp16) + wave * 256 + lane * 4, lds_addr(L.lut) + wave * 1024);
  rows_fetch(grp, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#if RN_INSTRUMENT
  if (tl) tl[39] = (unsigned)__builtin_amdgcn_s_memtime();
#endif

#pragma unroll 1
  for (int it = 0;; it++) {
    const int ib = PERSIST ? (it & 1) : 0, tile0 = grp * GM;
    const int next_grp = grp + (int)gridDim.x;
    const bool has_next = PERSIST && next_grp < n_groups;
    unsigned livemask = 0;
#pragma unroll
    for (int t = 0; t < GM; t++) {
      const int s = (tile0 + t) * TS + n, sc = s < N ? s : N - 1;
      livemask |= (s < N && !g.silence[(unsigned)sc]) ? 1u << t : 0u;  // silent streams keep their state (src/denoise.c:474)
    }
#pragma unroll 1
    for (int ui = 0; ui < UT; ui++) {
      const int u = wave + W * ui, unit0 = 16 * u + 4 * gq;
      const unsigned a0 = (unsigned)(u * KT * 64 + lane) * 16u;
      auto row4 = [&](int gate) { return (unsigned)(gate * RN_GRU + unit0) * 4u; };  // byte offset of the lane's 4 rows of a gate
      GRU_TL(0);
      v4i bf[2][GM];
      v4f h_old[GM], z[GM], rg[GM];
      {  // ---- update and reset gates ----
        v4i acc[2][GM];
        v4f gi[2][GM];
        AFragsG<AD, 2> A;
#pragma unroll
        for (int gate = 0; gate < 2; gate++) {
          const v4i rs = ldg<v4i>(wi.rowsum128, row4(gate));  // (acc_x86 = acc_mfma + 128 rowsum(w))
#pragma unroll
          for (int t = 0; t < GM; t++) acc[gate][t] = rs;
        }
#pragma unroll
        for (int step = 0; step < AD; step++) a_fetch_g<AD, 2, 0, HITA>(A, step, wi.wmf, wr.wmf, a0);
        if (BD) b_fetch<true>(bf, 0, L.xq[ib], 0, lane);
        if (MPRIO) __builtin_amdgcn_s_setprio(2);
        int8_gates_g<AD, 2, 0, BD, NOMFMA, HITA>(acc, A, bf, 0, wi.wmf, wr.wmf, a0, lane, L.xq[ib], L.hq[ib]);
        GRU_TL(1);
#pragma unroll
        for (int gate = 0; gate < 2; gate++) {  // float(acc_x86)*scale + subias (src/nnet_arch.h:145-151)
          const v4f sc = ldg<v4f>(wi.scale, row4(gate));
          const v4f sb = ldg<v4f>(wi.bias, row4(gate));
          const v4i rs = ldg<v4i>(wr.rowsum128, row4(gate));
#pragma unroll
          for (int t = 0; t < GM; t++) {
#pragma unroll
            for (int r = 0; r < 4; r++) gi[gate][t][r] = (float)acc[gate][t][r] * sc[r] + sb[r];
            acc[gate][t] = rs;
          }
        }
        GRU_TL(2);
        int8_gates_g<AD, 2, 0, BD, NOMFMA, HITA>(acc, A, bf, KT, wi.wmf, wr.wmf, a0, lane, L.hq[ib], nullptr);
        if (MPRIO) __builtin_amdgcn_s_setprio(0);
        GRU_TL(3);
        // all of this wave's loads have landed (the last A fragment was just used): its f32 rows for this unit tile are in LDS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int t = 0; t < GM; t++) h_old[t] = *reinterpret_cast<const v4f *>(&L.hrow[wave][TS * t + n][4 * gq]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // the row buffer is free: the next unit tile's rows (and, once per group, the next group's images) start their trip under the
        // sigmoids below, which load nothing but six constant vectors
        __builtin_amdgcn_sched_barrier(0);
        if (ui + 1 < UT) rows_fetch(grp, ui + 1);
        else if (has_next) rows_fetch(next_grp, 0);
        if (PERSIST && ui == 0 && has_next) images_fetch(next_grp, ib ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        // gate by gate (the update gate's accumulators and input sums are dead before the reset gate's conversion starts: the
        // register peak of the unit tile), the eight pair lookups of a gate's four tiles in flight together
#pragma unroll
        for (int gate = 0; gate < 2; gate++) {
          const v4f sc = ldg<v4f>(wr.scale, row4(gate));
          const v4f sb = ldg<v4f>(wr.bias, row4(gate));
          const v4f dg = ldg<v4f>(wr.diag, row4(gate));
          // (two tiles = four pair lookups in flight at a time: with all four tiles' the compiler ran out of its 168 registers and
          //  spilled the looked-up entries one by one)
#pragma unroll
          for (int th = 0; th < GM; th += 2) {
            ActPre2 ap[2][2];
#pragma unroll
            for (int t = th; t < th + 2; t++) {
              v4f gr;
#pragma unroll
              for (int r = 0; r < 4; r++) {
                gr[r] = (float)acc[gate][t][r] * sc[r] + sb[r];
                gr[r] += dg[r] * h_old[t][r];  // src/nnet_arch.h:153-161
              }
#pragma unroll
              for (int p = 0; p < 2; p++) {
                const v2f gv = {gi[gate][t][2 * p], gi[gate][t][2 * p + 1]}, rv = {gr[2 * p], gr[2 * p + 1]};
                if (!NOACT) ap[t - th][p] = sigmoid_pre2_lut0(gv + rv);
                else ap[t - th][p].numx = gv + rv;  // (timing experiment: no activation arithmetic, no lookups)
              }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = th; t < th + 2; t++)
#pragma unroll
              for (int p = 0; p < 2; p++) {
                v2f o = NOACT ? ap[t - th][p].numx : sigmoid_fin2_k(ap[t - th][p]);
                // (pinned: the compiler otherwise sinks this half of the activation to its use behind the candidate's MFMA blocks
                //  and keeps its six inputs per pair alive instead of the two results -- 30 dwords of scratch per unit tile)
                asm volatile("" : "+v"(o.x), "+v"(o.y));
                if (gate == 0) {
                  z[t][2 * p] = o.x;
                  z[t][2 * p + 1] = o.y;
                } else {
                  rg[t][2 * p] = o.x;
                  rg[t][2 * p + 1] = o.y;
                }
              }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      GRU_TL(4);
      {  // ---- candidate gate, blend, stores ----
        v4i acc[1][GM];
        v4f gi[GM];
        AFragsG<AD, 1> A;
        {
          const v4i rs = ldg<v4i>(wi.rowsum128, row4(2));
#pragma unroll
          for (int t = 0; t < GM; t++) acc[0][t] = rs;
        }
#pragma unroll
        for (int step = 0; step < AD; step++) a_fetch_g<AD, 1, 2, HITA>(A, step, wi.wmf, wr.wmf, a0);
        if (BD) b_fetch<true>(bf, 0, L.xq[ib], 0, lane);
        if (MPRIO) __builtin_amdgcn_s_setprio(2);
        int8_gates_g<AD, 1, 2, BD, NOMFMA, HITA>(acc, A, bf, 0, wi.wmf, wr.wmf, a0, lane, L.xq[ib], L.hq[ib]);
        GRU_TL(5);
        {
          const v4f sc = ldg<v4f>(wi.scale, row4(2));
          const v4f sb = ldg<v4f>(wi.bias, row4(2));
          const v4i rs = ldg<v4i>(wr.rowsum128, row4(2));
#pragma unroll
          for (int t = 0; t < GM; t++) {
#pragma unroll
            for (int r = 0; r < 4; r++) gi[t][r] = (float)acc[0][t][r] * sc[r] + sb[r];
            acc[0][t] = rs;
          }
        }
        int8_gates_g<AD, 1, 2, BD, NOMFMA, HITA>(acc, A, bf, KT, wi.wmf, wr.wmf, a0, lane, L.hq[ib], nullptr);
        if (MPRIO) __builtin_amdgcn_s_setprio(0);
        GRU_TL(6);
        const v4f sc = ldg<v4f>(wr.scale, row4(2));
        const v4f sb = ldg<v4f>(wr.bias, row4(2));
        const v4f dg = ldg<v4f>(wr.diag, row4(2));
#pragma unroll
        for (int t = 0; t < GM; t++) {
          v4f gr, hn;
#pragma unroll
          for (int r = 0; r < 4; r++) {
            gr[r] = (float)acc[0][t][r] * sc[r] + sb[r];
            gr[r] += dg[r] * h_old[t][r];
          }
          ActPre2 ah[2];
#pragma unroll
          for (int p = 0; p < 2; p++) {
            const v2f gh = {gi[t][2 * p], gi[t][2 * p + 1]}, rh = {gr[2 * p], gr[2 * p + 1]}, rv = {rg[t][2 * p], rg[t][2 * p + 1]};
            if (!NOACT) ah[p] = tanh_pre2_lut0(gh + rh * rv);
            else ah[p].numx = gh + rh * rv;
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int p = 0; p < 2; p++) {
            const v2f ho = {h_old[t][2 * p], h_old[t][2 * p + 1]}, zz = {z[t][2 * p], z[t][2 * p + 1]};
            const v2f hv = zz * ho + (v2f{1.f, 1.f} - zz) * (NOACT ? ah[p].numx : tanh_fin2_k(ah[p]));
            hn[2 * p] = hv.x;
            hn[2 * p + 1] = hv.y;
          }
          if (livemask >> t & 1) {  // (live implies tile0 + t < n_tiles and its stream < N)
            stg<v4f>(st, (unsigned)(((tile0 + t) * TS + n) * RN_GRU + unit0) * 4u, hn);
            stg<int>(himg, (unsigned)((tile0 + t) * (KT * 64 * 16) + frag_off(n, unit0)), pack4_g(hn[0], hn[1], hn[2], hn[3]));
          }
        }
      }
      GRU_TL(7);
    }
    if (!has_next) break;
    grp = next_grp;
    // (see gru_body2: every wave's image pieces were issued in its first unit tile and drained by the vmcnt(0) of its second)
    __builtin_amdgcn_s_barrier();
  }
#undef GRU_TL
}
#define GRU3_KERNEL(name, opt)                                                                                                        \
  extern "C" __global__ void __launch_bounds__(64 * G3W) name(RnGroupDev g, RnModelDev m, RnTablesDev tb, int layer) {             \
    gru_body3<opt>(g, m, tb, layer);                                                                                                  \
  }
GRU3_KERNEL(rn_nn_gru3_kernel, GRU_PERSIST | GRU_BD)
GRU3_KERNEL(rn_nn_gru3_nobd_kernel, GRU_PERSIST)
GRU3_KERNEL(rn_nn_gru3_np_kernel, GRU_BD)
GRU3_KERNEL(rn_nn_gru3_mprio_kernel, GRU_PERSIST | GRU_BD | GRU3_MPRIO)
GRU3_KERNEL(rn_nn_gru3_nomfma_kernel, GRU_PERSIST | GRU_BD | GRU3_NOMFMA)
GRU3_KERNEL(rn_nn_gru3_noact_kernel, GRU_PERSIST | GRU_BD | GRU3_NOACT)
GRU3_KERNEL(rn_nn_gru3_neither_kernel, GRU_PERSIST | GRU_BD | GRU3_NOACT | GRU3_NOMFMA)
GRU3_KERNEL(rn_nn_gru3_hita_kernel, GRU_PERSIST | GRU_BD | GRU3_HITA)
GRU3_KERNEL(rn_nn_gru3_hita_neither_kernel, GRU_PERSIST | GRU_BD | GRU3_HITA | GRU3_NOACT | GRU3_NOMFMA)

// The shipping form since round 5: four waves, one row buffer per wave, 72 KB -- two workgroups per CU (round 4 shipped the eight-wave
// 152 KB form below; profiles/r5_gru_bound.txt has the A/B and profiles/r5_gru_race.txt why this one could not ship before)
extern "C" __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
rn_nn_gru_kernel(RnGroupDev g, RnModelDev m, RnTablesDev tb, int layer) {
  gru_body<2, 4, 1, false>(g, m, tb, layer);
}
// A/B variants (not taken by default): see GruLdsT
extern "C" __global__ void __launch_bounds__(512) rn_nn_gru_w8_kernel(RnGroupDev g, RnModelDev m, RnTablesDev tb, int layer) {
  gru_body<2, 8, 3, false>(g, m, tb, layer);
}
extern "C" __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
rn_nn_gru_w4b2_kernel(RnGroupDev g, RnModelDev m, RnTablesDev tb, int layer) {
  gru_body<2, 4, 2, false>(g, m, tb, layer);
}
extern "C" __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
rn_nn_gru_w4nodma_kernel(RnGroupDev g, RnModelDev m, RnTablesDev tb, int layer) {
  gru_body<2, 4, 1, false, false>(g, m, tb, layer);
}
extern "C" __global__ void __launch_bounds__(512) rn_nn_gru_w8b1_kernel(RnGroupDev g, RnModelDev m, RnTablesDev tb, int layer) {
  gru_body<2, 8, 1, false>(g, m, tb, layer);
}
#if RN_INSTRUMENT
extern "C" __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
rn_nn_gru_w4_chk_kernel(RnGroupDev g, RnModelDev m, RnTablesDev tb, int layer) {
  gru_body<2, 4, 1, true>(g, m, tb, layer);
}
extern "C" __global__ void __launch_bounds__(512) rn_nn_gru_w8b1_chk_kernel(RnGroupDev g, RnModelDev m, RnTablesDev tb, int layer) {
  gru_body<2, 8, 1, true>(g, m, tb, layer);
}
extern "C" __global__ void __launch_bounds__(512) rn_nn_gru_chk_kernel(RnGroupDev g, RnModelDev m, RnTablesDev tb, int layer) {
  gru_body<2, 8, 3, true>(g, m, tb, layer);
}
// copies the race log to the host and clears it
extern "C" hipError_t rn_gru_race_log_read(unsigned *out, int words) {
  const size_t n = sizeof(rn_gru_race_log);
  if ((size_t)words * 4 < n) return hipErrorInvalidValue;
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpyFromSymbol(out, HIP_SYMBOL(rn_gru_race_log), n);
  static const unsigned zero[sizeof(rn_gru_race_log) / 4] = {};
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(rn_gru_race_log), zero, n);
  return e;
}
#endif

extern "C" hipError_t rn_launch_nn_gru_layer(const RnGroupDev *g, const RnModelDev *m, const RnTablesDev *tb, int layer, hipStream_t st,
                                             hipEvent_t e0, hipEvent_t e1) {
  const int n_tiles = (g->n_streams + TS - 1) / TS;
  typedef void (*Kernel)(RnGroupDev, RnModelDev, RnTablesDev, int);
  struct Variant { const char *name; Kernel k; int threads; size_t lds; bool persist; };
  static const Variant variants[] = {
      {"w4", rn_nn_gru_kernel, 256, sizeof(GruLdsT<4, 1>), false},          {"w8", rn_nn_gru_w8_kernel, 512, sizeof(GruLdsT<8, 3>), false},
      {"w4b2", rn_nn_gru_w4b2_kernel, 256, sizeof(GruLdsT<4, 2>), false},   {"w8b1", rn_nn_gru_w8b1_kernel, 512, sizeof(GruLdsT<8, 1>), false},
      {"w4nodma", rn_nn_gru_w4nodma_kernel, 256, sizeof(GruLdsT<4, 1>), false},  // no LDS-DMA: pieces through registers
      {"w4big", rn_nn_gru_kernel, 256, sizeof(GruLdsT<8, 3>), false},  // the w4 kernel asking for a whole CU's LDS: one workgroup per CU
      // round 5 (gru_body2): o0 = the restructured body with nothing switched on, then one change at a time, then together
      {"o0", rn_nn_gru2_o0_kernel, 512, sizeof(GruLds2T<1>), false},        {"bd", rn_nn_gru2_bd_kernel, 512, sizeof(GruLds2T<1>), false},
      {"deep", rn_nn_gru2_deep_kernel, 512, sizeof(GruLds2T<1>), false},    {"ax", rn_nn_gru2_ax_kernel, 512, sizeof(GruLds2T<1>), false},
      {"bdx", rn_nn_gru2_bdx_kernel, 512, sizeof(GruLds2T<1>), false},      {"p", rn_nn_gru2_p_kernel, 512, sizeof(GruLds2T<2>), true},
      {"pbd", rn_nn_gru2_pbd_kernel, 512, sizeof(GruLds2T<2>), true},       {"pall", rn_nn_gru2_pall_kernel, 512, sizeof(GruLds2T<2>), true},
      {"pbdx", rn_nn_gru2_pbdx_kernel, 512, sizeof(GruLds2T<2>), true},
      // ... second step (gru_body3): twelve waves, the unit tile's register block split by gates
      {"v3", rn_nn_gru3_kernel, 768, sizeof(GruLds3T<2>), true},            {"v3nobd", rn_nn_gru3_nobd_kernel, 768, sizeof(GruLds3T<2>), true},
      {"v3np", rn_nn_gru3_np_kernel, 768, sizeof(GruLds3T<1>), false},      {"v3mprio", rn_nn_gru3_mprio_kernel, 768, sizeof(GruLds3T<2>), true},
      // timing experiments, wrong results (what a part costs = what leaving it out saves):
      {"v3nomfma", rn_nn_gru3_nomfma_kernel, 768, sizeof(GruLds3T<2>), true}, {"v3noact", rn_nn_gru3_noact_kernel, 768, sizeof(GruLds3T<2>), true},
      {"v3neither", rn_nn_gru3_neither_kernel, 768, sizeof(GruLds3T<2>), true},
      {"v3hita", rn_nn_gru3_hita_kernel, 768, sizeof(GruLds3T<2>), true},   {"v3hitaneither", rn_nn_gru3_hita_neither_kernel, 768, sizeof(GruLds3T<2>), true},
#if RN_INSTRUMENT
      {"w4chk", rn_nn_gru_w4_chk_kernel, 256, sizeof(GruLdsT<4, 1>), false}, {"w8b1chk", rn_nn_gru_w8b1_chk_kernel, 512, sizeof(GruLdsT<8, 1>), false},
      {"w8chk", rn_nn_gru_chk_kernel, 512, sizeof(GruLdsT<8, 3>), false},
#endif
  };
  constexpr int NV = sizeof(variants) / sizeof(variants[0]);
  // $RNNOISE_AMD_GRU_VARIANT (A/B runs; an unknown name is an error, not a silent default).  Unset: by batch size --
  // the four-wave form (two workgroups per CU: 1-3 % under the eight-wave one stand-alone in every A/B of profiles/r5_gru_bound.txt)
  // once there are more groups than CUs; the eight-wave form while every group has a CU to itself (a four-wave workgroup would then
  // leave each SIMD with ONE wave: 16,384 streams 0.200 against 0.174 ms for the three layers + front + dense)
  static const int vi_env = [] {
    const char *e = getenv("RNNOISE_AMD_GRU_VARIANT");
    if (!e || !*e) return -2;
    for (int i = 0; i < NV; i++)
      if (!strcmp(e, variants[i].name)) return i;
    fprintf(stderr, "[rnnoise_amd] RNNOISE_AMD_GRU_VARIANT=%s: no such variant in this build\n", e);
    return -1;
  }();
  if (vi_env == -1) return hipErrorInvalidValue;
  static int cus[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  if (!cus[dev] && (hipDeviceGetAttribute(&cus[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus[dev] <= 0)) cus[dev] = 256;
  const int n_groups_ = (n_tiles + GM - 1) / GM;
  const int vi = vi_env >= 0 ? vi_env : (n_groups_ > cus[dev] ? 0 : 1);  // variants[0] = w4, [1] = w8
  const Variant &v = variants[vi];
  // more than 64 KB of LDS is an opt-in, per kernel and device (a process may hold batches on several GPUs)
  static bool opted[NV][64] = {};
  if (!opted[vi][dev]) {
    const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(v.k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)v.lds);
    if (attr != hipSuccess) return attr;
    opted[vi][dev] = true;
  }
  // persistent variants: one workgroup per CU, each walking over groups b, b + grid, ... -- $RNNOISE_AMD_GRU_GRID overrides the count (A/B)
  static const int grid_env = [] {
    const char *e = getenv("RNNOISE_AMD_GRU_GRID");
    return e ? atoi(e) : 0;
  }();
  const int n_groups = (n_tiles + GM - 1) / GM;
  const int grid = !v.persist ? n_groups : (grid_env > 0 ? (grid_env < n_groups ? grid_env : n_groups) : (cus[dev] < n_groups ? cus[dev] : n_groups));
  // bit 2: $RNNOISE_AMD_GRU_ACT=0; bits 3-4: $RNNOISE_AMD_GRU_SETTLE (gru_body)
  static const int flags = [] {
    const char *e = getenv("RNNOISE_AMD_GRU_ACT"), *s = getenv("RNNOISE_AMD_GRU_SETTLE"), *p = getenv("RNNOISE_AMD_GRU_PRIO"),
               *t = getenv("RNNOISE_AMD_GRU_TIMELINE");
    return ((e && atoi(e) == 0) ? 4 : 0) | ((s ? atoi(s) & 3 : 0) << 3) | ((p && atoi(p) == 0) ? 32 : 0) | ((t && atoi(t)) ? 64 : 0);
  }();
  RN_LAUNCH(v.k, dim3(grid), dim3(v.threads), v.lds, st, e0, e1, *g, *m, *tb, layer | flags);
  return hipGetLastError();
}

// ---- dense_out (1536 -> 32) and vad_dense (1536 -> 1) for 64 streams per workgroup (src/rnn.c:53-58) ------------------------
// cat = [conv2 out | gru1 | gru2 | gru3], f32, from HBM.  Both layers are serial chains over the 1536 inputs (bit parity
// fixes the summation order): dense_out as 384 dependent v_mfma_f32_16x16x4_f32 per (16-stream tile, 16-output row tile),
// vad_dense as 1536 unfused multiply-adds per stream (src/vec_avx.h:732-736).  The 16-stream tile kernel runs two MFMA
// chains and one VALU chain on its 8 waves; here every wave owns one MFMA chain (tile wave / 2, row tile wave % 2) and,
// in lanes 0..7, the VAD chains of 8 streams, whose multiply-adds issue in the shadow of the wave's own dependent MFMAs.
// Everything the chains consume arrives by LDS-DMA, three chunks of 64 inputs in flight or in use:
//   activations in the order the MFMA B operand wants, [input / 4][stream of the tile][input % 4] -- lane (n, gq) of step j
//     reads word 64 j + 4 n + gq, the 64 lanes 64 consecutive words; a VAD lane reads its stream's 16 bytes of the same
//     group; a DMA piece is 1 KB = 4 such groups, each lane fetching its 16 bytes from wherever its stream's row lives;
//   dense_out weights in MFMA A-operand order (model.cpp: stage_linear), one piece per row tile and group of four steps.
// No compiler-counted vector load is left in the loop: vmcnt retires in order, so a counted load issued behind a DMA piece
// would make its consumer wait for that piece's HBM trip (first version: weights through a register ring, 132 us; the
// chains stalled once per chunk).
// Measured dead ends: four / five activation chunks in flight instead of two with the fetching split by role (waves 0-3
// activations, 4-5 weights: 134-153 us against 111); the chains run inside the GRU kernels on the state they have just
// written to LDS, partial sums handed on through g.gains / g.vad (no dense kernel, no f32 re-read: the 96-step tails are
// MFMA-pipe bound at 2 chains per SIMD and wait for the slowest wave: +36 us per layer, the same total).
#define DKC 64  // inputs per staged chunk
#define DNB 3   // chunks in LDS
struct DenseLds {
  float vadw[RN_CAT];
  struct {
    float x[GM][DKC / 4][TS][4];
    float w[2][DKC / 16][64][4];
  } buf[DNB];
};
static_assert(2 * sizeof(DenseLds) <= 160 * 1024, "two workgroups per CU");

// one chunk of this wave's chains.  VAD: the wave also runs the VAD chains, lane = stream (tile lane / 16): its
// multiply-adds (8 VALU operations per step) issue in the shadow of the dependent MFMA in front of them
template <bool VAD>
__device__ __forceinline__ void dense_chunk(const DenseLds &L, int bi, int c, int t, int rt, int lane, v4f &dacc, float &vacc) {
  const int n = lane & 15, gq = lane >> 4;
  struct Ops {
    v4f a;       // MFMA A operands of four steps
    float b[4];  // MFMA B operands
    v4f x[4];    // VAD: this lane's stream, the inputs of the same steps ...
    v4f w[4];    // ... and their weights (same address in every lane: an LDS broadcast)
  };
  const float *bx = &L.buf[bi].x[t][0][n][gq];
  const float *vx = &L.buf[bi].x[lane >> 4][0][lane & 15][0];
  const float *ax = &L.buf[bi].w[rt][0][lane][0];
  const float *vw = L.vadw + c * DKC;
  auto operands = [&](int t4, Ops &o) {
    o.a = *reinterpret_cast<const v4f *>(ax + t4 * 256);
#pragma unroll
    for (int e = 0; e < 4; e++) {
      o.b[e] = bx[(4 * t4 + e) * (TS * 4)];
      if (VAD) {
        o.x[e] = *reinterpret_cast<const v4f *>(vx + (4 * t4 + e) * (TS * 4));
        o.w[e] = *reinterpret_cast<const v4f *>(vw + 16 * t4 + 4 * e);
      }
    }
  };
  Ops ops[2];
  operands(0, ops[0]);
#pragma unroll
  for (int t4 = 0; t4 < DKC / 16; t4++) {
    if (t4 + 1 < DKC / 16) operands(t4 + 1, ops[(t4 + 1) & 1]);  // one group ahead: the chains never wait for LDS
    __builtin_amdgcn_sched_barrier(0);  // (left alone the scheduler hoists the whole chunk's reads: 256 VGPRs and spills)
    const Ops &o = ops[t4 & 1];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      dacc = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[e], o.b[e], dacc, 0, 0, 0);
      if (VAD) {
#pragma unroll
        for (int i = 0; i < 4; i++) vacc = vacc + o.w[e][i] * o.x[e][i];  // unfused, src/vec_avx.h:732-736
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

extern "C" __global__ void __launch_bounds__(GTHREADS) rn_nn_dense_kernel(RnGroupDev g, RnModelDev m, RnTablesDev tb) {
  __shared__ __attribute__((aligned(16))) DenseLds L;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 15, gq = lane >> 4;
  const int N = g.n_streams, tile0 = blockIdx.x * GM;
  const int t = wave >> 1, rt = wave & 1;             // this wave's MFMA chain
  constexpr int NCH = RN_CAT / DKC;                   // 24 chunks
  constexpr int XP = GM * DKC * TS * 4 / 1024;        // 16 activation pieces per chunk
  constexpr int WP = 2 * (DKC / 16);                  // 8 weight pieces per chunk
  constexpr int PPW = (XP + WP) / GW;                 // 3 pieces per wave and chunk
  constexpr int NW4 = RN_CAT / 16;
  static_assert((XP + WP) % GW == 0 && XP % GW == 0 && GW == 2 * GM && RN_GRU % DKC == 0 && DKC == 64, "dense staging");

  auto stage_fetch = [&](int c) {
    const int seg = (c * DKC) / RN_GRU, k0 = c * DKC - seg * RN_GRU, bi = c % DNB;
    const float *src = seg == 0 ? g.nn_act : g.gru_state + (size_t)(seg - 1) * g.n_stride * RN_GRU;
#pragma unroll
    for (int j = 0; j < XP / GW; j++) {  // piece p = (tile p / 4, groups 4 (p % 4) .. + 3); lane l = (group l / 16, stream l % 16)
      const int p = wave + j * GW, pt = p >> 2, grp = 4 * (p & 3) + (lane >> 4);
      const int s = (tile0 + pt) * TS + (lane & 15), sc = s < N ? s : N - 1;
      dma_1k(src + ((size_t)sc * RN_GRU + k0 + 4 * grp), lds_addr(&L.buf[bi].x[pt][4 * (p & 3)][0][0]));
    }
    {  // weights: piece = (row tile wave / 4, group wave % 4 of the chunk's four)
      const int wr = wave >> 2, wg = wave & 3;
      dma_1k(m.dense_out.fwm + (((size_t)wr * NW4 + (DKC / 16) * c + wg) * 64 + lane) * 4, lds_addr(&L.buf[bi].w[wr][wg][0][0]));
    }
  };
  static_assert(RN_CAT * 4 / 1024 <= GW, "vad_dense weights: at most one piece per wave");
  if (wave < RN_CAT * 4 / 1024) dma_1k(m.vad_dense.fw + wave * 256 + lane * 4, lds_addr(L.vadw) + wave * 1024);
  stage_fetch(0);
  stage_fetch(1);

  v4f dacc = {0, 0, 0, 0};
  float vacc = 0;
  const bool vad_wave = wave == GW - 1;
#pragma unroll 1
  for (int c = 0; c < NCH; c++) {
    // chunk c has landed: this wave's pieces (in order; chunk c + 1's may still fly), then everybody's -- and everybody
    // is done with chunk c - 1, whose buffer the fetch below refills
    if (c + 1 < NCH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (c + 2 < NCH) stage_fetch(c + 2);
    if (vad_wave) dense_chunk<true>(L, c % DNB, c, t, rt, lane, dacc, vacc);
    else dense_chunk<false>(L, c % DNB, c, t, rt, lane, dacc, vacc);
  }
  // (five activations per lane: the rcpps table straight from memory)
  {
    const int s = (tile0 + t) * TS + n, sc = s < N ? s : N - 1;
    const bool live = s < N && !g.silence[sc];
    const int row0 = 16 * rt + 4 * gq;
    const v4f bs = *reinterpret_cast<const v4f *>(m.dense_out.bias + row0);
    v4f o;
#pragma unroll
    for (int r = 0; r < 4; r++) o[r] = live ? sigmoid_x86(dacc[r] + bs[r], tb.rcp16) : 0.f;
    if (s < N) *reinterpret_cast<v4f *>(g.gains + (size_t)sc * RN_NB_BANDS + row0) = o;
  }
  if (vad_wave) {
    const int vs = tile0 * TS + lane;
    if (vs < N) g.vad[vs] = g.silence[vs] ? 0.f : sigmoid_x86(vacc + m.vad_dense.bias[0], tb.rcp16);
  }
}

extern "C" hipError_t rn_launch_nn_dense(const RnGroupDev *g, const RnModelDev *m, const RnTablesDev *tb, hipStream_t st, hipEvent_t e0,
                                         hipEvent_t e1) {
  const int n_tiles = (g->n_streams + TS - 1) / TS;
  RN_LAUNCH(rn_nn_dense_kernel, dim3((n_tiles + GM - 1) / GM), dim3(GTHREADS), 0, st, e0, e1, *g, *m, *tb);
  return hipGetLastError();
}

// Rebuilds the state images act_q[1..3] from the f32 GRU state (after a reset, an import, or steps taken by the other
// network kernels): one workgroup per (tile, layer).
extern "C" __global__ void __launch_bounds__(256) rn_nn_requant_kernel(RnGroupDev g) {
  const int tile = blockIdx.x, layer = blockIdx.y, N = g.n_streams;
  const float *st = g.gru_state + (size_t)layer * g.n_stride * RN_GRU;
  int8_t *img = g.act_q[layer + 1] + (size_t)tile * (KT * 64 * 16);
  for (int e = threadIdx.x; e < TS * 96; e += 256) {
    const int q = e / 96, c4 = (e - q * 96) << 2, s = (tile * TS + q < N) ? tile * TS + q : N - 1;
    const v4f h = *reinterpret_cast<const v4f *>(st + (size_t)s * RN_GRU + c4);
    *reinterpret_cast<int *>(img + frag_off(q, c4)) = pack4(h[0], h[1], h[2], h[3]);
  }
}
extern "C" hipError_t rn_launch_nn_requant(const RnGroupDev *g, hipStream_t st) {
  hipLaunchKernelGGL(rn_nn_requant_kernel, dim3((g->n_streams + TS - 1) / TS, 3), dim3(256), 0, st, *g);
  return hipGetLastError();
}
