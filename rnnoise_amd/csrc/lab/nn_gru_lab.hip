// lab/nn_gru_lab.hip -- INSTRUMENTED BUILD ONLY (librnnoise_amd_instr.so; the product libraries are not linked against this file).
// The forms of the GRU layer kernel that were measured against the two the product ships (nn_layers.hip: w4, w8), kept so that the
// A/B tables of profiles/r4_gru_experiments.txt, r5_gru_bound.txt and r5_gru_race.txt stay reproducible (tools/gru_variants.py,
// tools/gru_race.py, tools/gru_timeline.py select them by $RNNOISE_AMD_GRU_VARIANT on the instrumented library):
//   gru_body (nn_gru.h) once more: w4b2, w8b1 (row buffers), w4nodma (no LDS-DMA), w4big (one workgroup per CU), *chk (every row
//     vector a lane takes from LDS compared with HBM: the race hunt), w4pk / w4sc (activations in packed / scalar math, whichever the
//     product does not take);
//   gru_body2: eight waves restructured -- B fragments a k-step ahead, deeper lookup batching, prefetched tile heads, persistent;
//   gru_body3: twelve waves, the unit tile's register block split by gates -- among them FIVE TIMING-ONLY BUILDS WITH WRONG RESULTS
//     (v3nomfma, v3noact, v3neither, v3hita, v3hitaneither: what a part costs = what leaving it out saves).
#if !RN_INSTRUMENT
#error "lab/nn_gru_lab.hip belongs to the instrumented build only"
#endif
#define RN_GRU_LAB 1
#include "../nn_common.h"
// CHK instantiations (tools/gru_race.py): every h_old vector a lane takes from its LDS row buffer is compared with the same 16
// bytes loaded straight from HBM.  [0] = mismatching vectors seen, [1] = of them equal to the PREVIOUS unit tile's vector of that
// lane (stale buffer: the LDS-DMA had not landed), [2] = vectors checked (low 32 bits); then up to 40 records of 12 words:
// block | wave, ui, t, lane | got[4] | want[4] | previous tile's[2].  [3] = words of the LDS images (layer input, recurrent state,
// rcpps table) that differed from HBM behind the prologue's barrier or (input image, table) at the end of the kernel; their
// records: block | 0xffff0000 + 0x100 * (0 start, 1 end) + region (0..3 xq, 4..7 hq, 8 table) | word | got | want | wave
__device__ unsigned rn_gru_race_log[4 + 40 * 12];
#include "../nn_gru.h"
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

// ---- gru_body (nn_gru.h) in the forms the product does not take ----
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

// the product's two kernels (nn_layers.hip), for the entries that reuse them with other launch parameters
extern "C" __global__ void rn_nn_gru_kernel(RnGroupDev g, RnModelDev m, RnTablesDev tb, int layer);
// the activations in the math the product does NOT use (RN_GRU_PACKED_ACT), same bits either way
extern "C" __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
rn_nn_gru_w4_altact_kernel(RnGroupDev g, RnModelDev m, RnTablesDev tb, int layer) {
  gru_body<2, 4, 1, false, true, !RN_GRU_PACKED_ACT>(g, m, tb, layer);
}

// round 6 (profiles/r6_dense_fold.txt): eight waves, one row buffer per wave, AND the output chains (dense_out, vad_dense) advanced
// inside the layer launch as the units are produced (nn_gru.h: gru_body FOLD; the front kernel starts them: rn_nn_front_fold_kernel,
// nn_mfma.hip) -- four launches instead of five, no 6 KB-per-stream re-read of the f32 state.  Bit-exact, and slower than the
// five-launch network it was meant to replace.
extern "C" __global__ void __launch_bounds__(512) rn_nn_gru_fold_kernel(RnGroupDev g, RnModelDev m, RnTablesDev tb, int layer) {
  gru_body<2, 8, 1, false, true, RN_GRU_PACKED_ACT, true>(g, m, tb, layer);
}

extern "C" const RnGruVariant *rn_gru_lab_variant(const char *name) {
  static const RnGruVariant variants[] = {
      {"w8f", rn_nn_gru_fold_kernel, 512, sizeof(GruLdsT<8, 1, true>), false, true},
      {"w4b2", rn_nn_gru_w4b2_kernel, 256, sizeof(GruLdsT<4, 2>), false, false},   {"w8b1", rn_nn_gru_w8b1_kernel, 512, sizeof(GruLdsT<8, 1>), false, false},
      {"w4nodma", rn_nn_gru_w4nodma_kernel, 256, sizeof(GruLdsT<4, 1>), false, false},  // no LDS-DMA: pieces through registers
      {"w4big", rn_nn_gru_kernel, 256, sizeof(GruLdsT<8, 3>), false, false},  // the product's w4 kernel asking for a whole CU's LDS: one workgroup per CU
      {RN_GRU_PACKED_ACT ? "w4sc" : "w4pk", rn_nn_gru_w4_altact_kernel, 256, sizeof(GruLdsT<4, 1>), false, false},
      // round 5 (gru_body2): o0 = the restructured body with nothing switched on, then one change at a time, then together
      {"o0", rn_nn_gru2_o0_kernel, 512, sizeof(GruLds2T<1>), false, false},        {"bd", rn_nn_gru2_bd_kernel, 512, sizeof(GruLds2T<1>), false, false},
      {"deep", rn_nn_gru2_deep_kernel, 512, sizeof(GruLds2T<1>), false, false},    {"ax", rn_nn_gru2_ax_kernel, 512, sizeof(GruLds2T<1>), false, false},
      {"bdx", rn_nn_gru2_bdx_kernel, 512, sizeof(GruLds2T<1>), false, false},      {"p", rn_nn_gru2_p_kernel, 512, sizeof(GruLds2T<2>), true, false},
      {"pbd", rn_nn_gru2_pbd_kernel, 512, sizeof(GruLds2T<2>), true, false},       {"pall", rn_nn_gru2_pall_kernel, 512, sizeof(GruLds2T<2>), true, false},
      {"pbdx", rn_nn_gru2_pbdx_kernel, 512, sizeof(GruLds2T<2>), true, false},
      // ... second step (gru_body3): twelve waves, the unit tile's register block split by gates
      {"v3", rn_nn_gru3_kernel, 768, sizeof(GruLds3T<2>), true, false},            {"v3nobd", rn_nn_gru3_nobd_kernel, 768, sizeof(GruLds3T<2>), true, false},
      {"v3np", rn_nn_gru3_np_kernel, 768, sizeof(GruLds3T<1>), false, false},      {"v3mprio", rn_nn_gru3_mprio_kernel, 768, sizeof(GruLds3T<2>), true, false},
      // timing experiments, WRONG RESULTS (what a part costs = what leaving it out saves):
      {"v3nomfma", rn_nn_gru3_nomfma_kernel, 768, sizeof(GruLds3T<2>), true, false}, {"v3noact", rn_nn_gru3_noact_kernel, 768, sizeof(GruLds3T<2>), true, false},
      {"v3neither", rn_nn_gru3_neither_kernel, 768, sizeof(GruLds3T<2>), true, false},
      {"v3hita", rn_nn_gru3_hita_kernel, 768, sizeof(GruLds3T<2>), true, false},   {"v3hitaneither", rn_nn_gru3_hita_neither_kernel, 768, sizeof(GruLds3T<2>), true, false},
      {"w4chk", rn_nn_gru_w4_chk_kernel, 256, sizeof(GruLdsT<4, 1>), false, false}, {"w8b1chk", rn_nn_gru_w8b1_chk_kernel, 512, sizeof(GruLdsT<8, 1>), false, false},
      {"w8chk", rn_nn_gru_chk_kernel, 512, sizeof(GruLdsT<8, 3>), false, false},
  };
  for (const RnGruVariant &v : variants)
    if (!strcmp(name, v.name)) return &v;
  return nullptr;
}
