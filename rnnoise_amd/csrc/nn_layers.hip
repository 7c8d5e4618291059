// nn_layers.hip -- K2 for large batches (from 10,240 streams up, batch.cpp: nn_layers_min_streams): the network layer by layer.
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
#include "nn_gru.h"

// four waves, one row buffer per wave, 72 KB: two workgroups per CU (profiles/r5_gru_bound.txt has the A/B against the eight-wave form
// and profiles/r5_gru_race.txt why this one could not ship before round 5)
extern "C" __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
rn_nn_gru_kernel(RnGroupDev g, RnModelDev m, RnTablesDev tb, int layer) {
  gru_body<2, 4, 1, false>(g, m, tb, layer);
}
// eight waves, a row buffer per unit tile, 152 KB: a workgroup owns its CU
extern "C" __global__ void __launch_bounds__(512) rn_nn_gru_w8_kernel(RnGroupDev g, RnModelDev m, RnTablesDev tb, int layer) {
  gru_body<2, 8, 3, false>(g, m, tb, layer);
}

#if RN_INSTRUMENT
extern "C" const RnGruVariant *rn_gru_lab_variant(const char *name);  // lab/nn_gru_lab.hip: the A/B forms of the instrumented build
#endif

static const RnGruVariant gru_product[2] = {{"w4", rn_nn_gru_kernel, 256, sizeof(GruLdsT<4, 1>), false, false},
                                             {"w8", rn_nn_gru_w8_kernel, 512, sizeof(GruLdsT<8, 3>), false, false}};
static const RnGruVariant *gru_forced_variant() {
  static const RnGruVariant *const forced = []() -> const RnGruVariant * {

    const char *e = getenv("RNNOISE_AMD_GRU_VARIANT");
    if (!e || !*e) return nullptr;
    for (const RnGruVariant &v : gru_product)
      if (!strcmp(e, v.name)) return &v;
#if RN_INSTRUMENT
    if (const RnGruVariant *v = rn_gru_lab_variant(e)) return v;
#endif
    fprintf(stderr, "[rnnoise_amd] RNNOISE_AMD_GRU_VARIANT=%s: no such form of the layer kernel in this build (w4 | w8)\n", e);
    static const RnGruVariant none = {nullptr, nullptr, 0, 0, false, false};
    return &none;
  }();
  return forced;
}
// does the layer-wise network of this process fold the output chains into its layer launches?  (nn_mfma.hip: rn_launch_nn_layers)
extern "C" int rn_nn_layers_fold(void) {
  const RnGruVariant *f = gru_forced_variant();
  return f && f->k && f->fold;
}

extern "C" hipError_t rn_launch_nn_gru_layer(const RnGroupDev *g, const RnModelDev *m, const RnTablesDev *tb, int layer, hipStream_t st,
                                             hipEvent_t e0, hipEvent_t e1) {
  const int n_tiles = (g->n_streams + TS - 1) / TS, n_groups = (n_tiles + GM - 1) / GM;
  // The product has exactly these two forms, with the same bits.  By batch size: the four-wave form (two workgroups per CU: 1-3 %
  // under the eight-wave one stand-alone in every A/B of profiles/r5_gru_bound.txt) once there are more groups than CUs; the eight-wave
  // form while every group has a CU to itself (a four-wave workgroup would then leave each SIMD with ONE wave: 16,384 streams 0.200
  // against 0.174 ms for the three layers + front + dense).  $RNNOISE_AMD_GRU_VARIANT = w4 | w8 forces one (tests run both at every
  // size); any other name is an error, not a silent default -- the instrumented build knows more names (lab/nn_gru_lab.hip).
  static const RnGruVariant *const forced = gru_forced_variant();
  if (forced && !forced->k) return hipErrorInvalidValue;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  // (per-device caches written from whichever thread launches first: relaxed atomics, every writer stores the same value)
  static std::atomic<int> cus[64];
  int ncu = cus[dev].load(std::memory_order_relaxed);
  if (!ncu) {
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
    cus[dev].store(ncu, std::memory_order_relaxed);
  }
  const RnGruVariant &v = forced ? *forced : gru_product[n_groups > ncu ? 0 : 1];
  // more than 64 KB of LDS is an opt-in, per kernel and device (a process may hold batches on several GPUs)
  if (hipError_t e = rn_gru_opt_in(v, dev)) return e;
  int grid = n_groups, flags = 0;
#if RN_INSTRUMENT
  // persistent lab forms: one workgroup per CU, each walking over groups b, b + grid, ... -- $RNNOISE_AMD_GRU_GRID overrides the count
  static const int grid_env = [] { const char *e = RN_LAB_ENV("GRU_GRID"); return e ? atoi(e) : 0; }();
  if (v.persist) grid = grid_env > 0 ? (grid_env < n_groups ? grid_env : n_groups) : (ncu < n_groups ? ncu : n_groups);
  // gru_body: bit 2 $RNNOISE_AMD_GRU_ACT=0; bits 3-4 $RNNOISE_AMD_GRU_SETTLE; bit 5 $RNNOISE_AMD_GRU_PRIO=0; bit 6 _GRU_TIMELINE (lab forms)
  static const int flags_env = [] {
    const char *e = RN_LAB_ENV("GRU_ACT"), *s = RN_LAB_ENV("GRU_SETTLE"), *p = RN_LAB_ENV("GRU_PRIO"), *t = RN_LAB_ENV("GRU_TIMELINE");
    return ((e && atoi(e) == 0) ? 4 : 0) | ((s ? atoi(s) & 3 : 0) << 3) | ((p && atoi(p) == 0) ? 32 : 0) | ((t && atoi(t)) ? 64 : 0);
  }();
  flags = flags_env;
#endif
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

