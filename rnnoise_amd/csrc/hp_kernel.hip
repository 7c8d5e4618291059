// hp_kernel.hip -- K0: the per-stream serial front end (high-pass, pitch-buffer decimation, 5-lag autocorrelation,
// Levinson), transposed to lane = stream.  Compiled WITHOUT the SLP vectoriser since round 5: with it (~9 % fewer
// instructions) the autocorrelation chains became v_pk_mul_f32 / v_pk_add_f32 with op_sel operand selects, and those take the
// wrong operand half in lanes 48..63 whenever a wave issuing v_mfma_i32_16x16x64_i8 shares the SIMD (profiles/r5_gru_race.txt;
// tests/test_kernel_budgets_cpu.py keeps every such instruction out of the library).  hp_slp.hip is the old build, for A/B runs.
// Numerics contract as in dsp_kernels.hip: -ffp-contract=off, reference order of every float sum.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "rn_dev.h"

#define WAVE 64

// ---------------------------------------------------------------------------------------------
// K0: rnn_biquad (src/denoise.c:409-419, coefficients :469-470), transposed: lane = stream.
// The recurrence is strictly serial per stream (every step rounds its state to float), so the
// wave-per-frame kernel would idle 63 of 64 lanes for 480 steps; here 64 streams advance in
// lock-step instead.  Output goes straight into the stream's pitch ring (slot `slot`).
// a0*yi and a1*yi are products of two 24-bit significands, exact in double, so
// fma(-a, yi, b*xi) rounds once exactly like the reference's (b*xi - a*yi).
// ---------------------------------------------------------------------------------------------
#ifndef RN_HP_KERNEL_NAME
#define RN_HP_KERNEL_NAME rn_hp_kernel
#endif
#ifndef RN_HP_BLK
#define RN_HP_BLK 8   // float4 per block and stream: 8 = one 128-byte line
#endif
#ifndef RN_HP_ATTR
#define RN_HP_ATTR
#endif
// The body is compiled once per (high-pass applied, int16 input): as run-time flags the two were tested inside the block loop, per
// float4, and behind those joins the compiler could no longer count the loads in flight -- it waited with vmcnt(0) in front of every
// block's arithmetic, i.e. for the NEXT block's loads it had just issued: the one-block prefetch hid nothing (round 6, last day).
template <bool HP_ON, bool IN_S16>
__device__ __forceinline__ void hp_body(const RnGroupDev &g, const float *__restrict__ in, int slot, int mode) {
  // mode: bit 0 = apply the high-pass (inference); bit 1 = `in` holds int16 samples, converted as the reference's only caller
  // does (examples/rnnoise_demo.c:56: x[i] = tmp[i], short -> float, exact)
  constexpr bool apply_hp = HP_ON, in_s16 = IN_S16;
  // bits 12-13: streams per wave = 64 >> k.  The kernel is bound by how many loads its waves keep in flight (one wave per SIMD at 64 streams
  // per wave and 65,536 streams: each lane's HBM round trips are its own), not by issue: half-empty waves are twice as many waves
  const int spw = WAVE >> ((mode >> 12) & 3);
  // (one wave per workgroup; or, A/B, four whole waves: blockDim.x = 256, one wave per SIMD of the CU that gets the workgroup)
  const int s = blockDim.x > WAVE ? blockIdx.x * blockDim.x + threadIdx.x : blockIdx.x * spw + threadIdx.x;
  if ((blockDim.x == WAVE && (int)threadIdx.x >= spw) || s >= g.n_streams) return;
  // (race hunt, $RNNOISE_AMD_HP_AB: 512 = raised issue priority, 1024 = drain the tap stores before the wave ends; 256 drained the
  //  wave's stores before the pitch ring was read back -- nothing is read back any more)
  if (RN_INSTRUMENT && (mode & 512)) __builtin_amdgcn_s_setprio(3);
  const float a0 = -1.99599f, a1 = 0.99600f, b0 = -2.f;
  const double na0 = -(double)a0, na1 = -(double)a1, b0d = (double)b0;
  float m0 = g.mem_hp[2 * s], m1 = g.mem_hp[2 * s + 1];
  const float4 *x = reinterpret_cast<const float4 *>(in + (size_t)s * RN_FRAME_SIZE);
  float4 *y = reinterpret_cast<float4 *>(g.pitch_ring + (size_t)s * RN_RING_SIZE + slot * RN_FRAME_SIZE);
  // the slot's 240 decimated samples (rn_dev.h: RN_XRING_SLOT) are formed from the filtered frame as it leaves the registers; the
  // first one needs the last sample of the previous slot
  float4 *y2 = reinterpret_cast<float4 *>(g.xlp_ring + (size_t)s * RN_XRING_SIZE + slot * RN_XRING_SLOT);
  const float *ring = g.pitch_ring + (size_t)s * RN_RING_SIZE;
  const float *xring = g.xlp_ring + (size_t)s * RN_XRING_SIZE;
  const int ring0 = RN_RING0(slot), x0 = ring0 / 2;
  float left = ring[(slot * RN_FRAME_SIZE + RN_RING_SIZE - 1) % RN_RING_SIZE];
  // x_lp[0] has no left neighbour (src/pitch.c:166): formed here from pitch_buf[0], pitch_buf[1]
  const float2 pb01 = *reinterpret_cast<const float2 *>(ring + ring0);
  const float xlp0 = .5f * (.5f * (pb01.y) + pb01.x);
  // BLK float4 (32 samples = one 128-byte line per stream at BLK = 8) per block, the next block's loads in flight while this one is
  // consumed: with one wave per SIMD nothing else hides the HBM round trip
  constexpr int BLK = RN_HP_BLK;  // float4 per block
  float4 cur[BLK], nxt[BLK];
  const short4 *x16 = reinterpret_cast<const short4 *>(reinterpret_cast<const short *>(in) + (size_t)s * RN_FRAME_SIZE);
  auto load4 = [&](int idx) -> float4 {
    if (in_s16) {
      const short4 q = x16[idx];
      return make_float4((float)q.x, (float)q.y, (float)q.z, (float)q.w);
    }
    return x[idx];
  };

  // ---- the serial half of rnn_pitch_downsample (src/pitch.c:146-214) rides along: 5-lag autocorrelation of the 2x decimated
  // pitch_buf (src/celt_lpc.c:92-174), lag window, order-4 Levinson (src/celt_lpc.c:38-89) -> the 5 FIR taps.  In the wave-per-frame
  // kernel these 5 chains of 864 steps used 5 lanes of 64; here every lane runs them over its own stream, keeping the last 4
  // decimated samples in registers.  For sample t and lag k the product xlp[t-k]*xlp[t] is term i = t-k of the reference's sum for
  // lag k: terms i < 860 go to the main chain (rnn_pitch_xcorr over fastN), later ones to the tail chain `d`.
  // Order of the work (round 6): FIRST the 624 decimated samples that older frames left in the decimated ring (no dependence on
  // this frame: 156 float4 from x0, the ring size a multiple of 4 so that a float4 never straddles the wrap), THEN the frame itself,
  // block by block: biquad -> ring slot -> its 16 decimated samples -> decimated slot AND straight on into the chains.  The new
  // samples are never read back, and the frame's first block is requested while the last old block is consumed.
  float ac[5] = {0, 0, 0, 0, 0};
  float w1 = 0, w2 = 0, w3 = 0, w4 = 0;  // xlp[t-1..t-4]; zeros before the start add exact +0 products
#define AC_MAIN(xv)                      \
  {                                      \
    const float x0_ = (xv);              \
    ac[0] = ac[0] + x0_ * x0_;           \
    ac[1] = ac[1] + w1 * x0_;            \
    ac[2] = ac[2] + w2 * x0_;            \
    ac[3] = ac[3] + w3 * x0_;            \
    ac[4] = ac[4] + w4 * x0_;            \
    w4 = w3; w3 = w2; w2 = w1; w1 = x0_; \
  }
  // t = 860 + e: term i = t-k is < 860 for k > e, else it belongs to the tail
#define AC_TAIL(xv, e)                                                        \
  {                                                                           \
    const float x0_ = (xv);                                                   \
    d[0] = d[0] + x0_ * x0_;                                                  \
    if ((e) >= 1) d[1] = d[1] + x0_ * w1; else ac[1] = ac[1] + w1 * x0_;      \
    if ((e) >= 2) d[2] = d[2] + x0_ * w2; else ac[2] = ac[2] + w2 * x0_;      \
    if ((e) >= 3) d[3] = d[3] + x0_ * w3; else ac[3] = ac[3] + w3 * x0_;      \
    ac[4] = ac[4] + w4 * x0_;                                                 \
    w4 = w3; w3 = w2; w2 = w1; w1 = x0_;                                      \
  }
  constexpr int OLD4 = (RN_PITCH_BUF_SIZE - RN_FRAME_SIZE) / 2 / 4;  // 156 float4 of old decimated samples
  constexpr int OLD_BLOCKS = (OLD4 + BLK - 1) / BLK, NEW_BLOCKS = RN_FRAME_SIZE / 4 / BLK;
  static_assert(RN_FRAME_SIZE / 4 % BLK == 0 && BLK % 2 == 0 && (RN_PITCH_BUF_SIZE / 2 - 860) == 4, "whole frame blocks; the tail terms are the frame's last two float4");
  auto old_block = [&](int b, float4 (&dst)[BLK]) {
#pragma unroll
    for (int j = 0; j < BLK; j++) {
      int p = x0 + 4 * min(BLK * b + j, OLD4 - 1);  // (a partial last block re-reads its last float4 and drops it)
      p = (p >= RN_XRING_SIZE) ? p - RN_XRING_SIZE : p;
      dst[j] = *reinterpret_cast<const float4 *>(xring + p);
    }
  };
  old_block(0, nxt);
  for (int b = 0; b < OLD_BLOCKS; b++) {
#pragma unroll
    for (int j = 0; j < BLK; j++) cur[j] = nxt[j];
    if (b + 1 < OLD_BLOCKS) {
      old_block(b + 1, nxt);
    } else {
#pragma unroll
      for (int j = 0; j < BLK; j++) nxt[j] = load4(j);
    }
#pragma unroll
    for (int j = 0; j < BLK; j++) {
      const int c = b * BLK + j;
      if (OLD4 % BLK == 0 || c < OLD4) {
        const float4 v = cur[j];
        AC_MAIN(c == 0 ? xlp0 : v.x) AC_MAIN(v.y) AC_MAIN(v.z) AC_MAIN(v.w)
      }
    }
  }
  for (int blk = 0; blk < NEW_BLOCKS; blk++) {
#pragma unroll
    for (int j = 0; j < BLK; j++) cur[j] = nxt[j];
    if (blk + 1 < NEW_BLOCKS) {
#pragma unroll
      for (int j = 0; j < BLK; j++) nxt[j] = load4((blk + 1) * BLK + j);
    }
#define HP_STEP(xi, yo)                                              \
    {                                                                \
      const float yi = (xi) + m0;                                    \
      const double xd = (double)(xi), yd = (double)yi;               \
      m0 = (float)((double)m1 + fma(na0, yd, b0d * xd));             \
      m1 = (float)fma(na1, yd, xd);                                  \
      (yo) = yi;                                                     \
    }
    float2 dec[BLK];
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
      dec[j].x = .5f * (.5f * (left + o.y) + o.x);  // (src/pitch.c:155-160)
      dec[j].y = .5f * (.5f * (o.y + o.w) + o.z);
      left = o.w;
    }
#undef HP_STEP
#pragma unroll
    for (int j = 0; j < BLK; j += 2) y2[(blk * BLK + j) / 2] = make_float4(dec[j].x, dec[j].y, dec[j + 1].x, dec[j + 1].y);
    if (blk + 1 < NEW_BLOCKS) {
#pragma unroll
      for (int j = 0; j < BLK; j++) { AC_MAIN(dec[j].x) AC_MAIN(dec[j].y) }
    } else {  // the frame's last block ends with decimated samples 860..863
#pragma unroll
      for (int j = 0; j < BLK - 2; j++) { AC_MAIN(dec[j].x) AC_MAIN(dec[j].y) }
      float d[5] = {0, 0, 0, 0, 0};
      AC_TAIL(dec[BLK - 2].x, 0) AC_TAIL(dec[BLK - 2].y, 1) AC_TAIL(dec[BLK - 1].x, 2) AC_TAIL(dec[BLK - 1].y, 3)
#pragma unroll
      for (int k = 0; k < 5; k++) ac[k] = ac[k] + d[k];
    }
  }
#undef AC_MAIN
#undef AC_TAIL
  if (apply_hp) {
    g.mem_hp[2 * s] = m0;
    g.mem_hp[2 * s + 1] = m1;
  }
  {
    float taps[5];
    rn_fir_taps_from_ac(ac, taps);  // (lag window, Levinson, bandwidth expansion: rn_dev.h)
    float *o = g.lpc2 + ((size_t)slot * g.n_stride + s) * 8;  // one copy per ring slot: K0 runs up to 2 frames ahead of K1
#pragma unroll
    for (int k = 0; k < 5; k++) o[k] = taps[k];
    if (RN_INSTRUMENT && g.debug) {
#pragma unroll
      for (int k = 0; k < 5; k++) g.debug[(size_t)s * RN_DBG_FLOATS + RN_DBG_AC + k] = ac[k];
    }
    if (RN_INSTRUMENT && (mode & 1024)) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    }
  }
}
// ONE kernel for the four forms.  (Tried: a kernel of its own for int16 input, so that the float forms keep their 101 registers
// instead of the 128 of the hungriest form -- slower, 0.147 against 0.135 ms at 65,536 streams: inside the common kernel the scheduler
// works the float forms to the looser budget too, and that is the faster code.  profiles/r6_hp_specialised.txt)
extern "C" __global__ void __launch_bounds__(4 * WAVE) RN_HP_ATTR
RN_HP_KERNEL_NAME(RnGroupDev g, const float *__restrict__ in, int slot, int mode) {
  if (mode & 2) {
    if (mode & 1) hp_body<true, true>(g, in, slot, mode);
    else hp_body<false, true>(g, in, slot, mode);
  } else {
    if (mode & 1) hp_body<true, false>(g, in, slot, mode);
    else hp_body<false, false>(g, in, slot, mode);
  }
}
#ifndef RN_HP_VARIANT_ONLY
// ---------------------------------------------------------------------------------------------
// K0 for a handful of streams (the one-stream states behind rnnoise_process_frame, and batches of up to 64 streams): the
// same arithmetic with ONE WAVE PER STREAM instead of one lane, arranged for latency.  What is serial stays serial -- the
// biquad recurrence runs once per wave (every lane computes the same values from broadcast LDS reads), each autocorrelation
// lag is one lane's chain in the reference's order -- but the five lags run side by side on lanes 0-4 instead of one after
// the other in a lane, the 2x decimation is spread over the wave, and global memory is touched in coalesced rows only
// (frame and pitch_buf come in through LDS).  43 us -> 22 us for one stream (the biquad chain alone is ~8 us).
// ---------------------------------------------------------------------------------------------
// One array (6.9 KB per wave: 16 waves fit a CU and a 4,096-stream batch is ONE round):
//   pb[0 .. 623]      the 624 decimated samples of pitch_buf that older frames left in the decimated ring (rn_dev.h: RN_XRING_SLOT),
//   pb[624 .. 863]    the frame's own 240, formed once the biquad is through; + zeros up to 935 for the reads of the idle lanes of the
//                     lag chains;
//   pb[1248 .. 1727]  first the UNFILTERED frame, then the filtered one: the biquad runs in place (a block's samples are in registers,
//                     and the next block's too, before its outputs are stored).
// (Rounds 1-5 loaded the 1248 old samples of the pitch ring and decimated all 864 in place: 14 passes instead of 4.)
struct HpOneLds {
  float pb[RN_PITCH_BUF_SIZE];
};
static_assert(864 + 64 + 8 <= RN_PITCH_BUF_SIZE - RN_FRAME_SIZE, "the decimated signal and its pad stay below the new frame");

// (the body once per input type, like hp_body above: behind the run-time test per load the frame's two loads and the three loads of
//  old samples were each followed by its own s_waitcnt vmcnt(0) -- five serial round trips, two of them to pinned host memory in the
//  one-frame API, in front of a kernel of ~20 us)
template <bool IN_S16>
__device__ __forceinline__ void hp_one_body(HpOneLds &L, const RnGroupDev &g, const float *__restrict__ in, const float *__restrict__ in_row,
                                            bool listed, int s, int slot, int slot_arg) {
  constexpr bool in_s16 = IN_S16;
  const int lane = threadIdx.x;
  const float a0 = -1.99599f, a1 = 0.99600f, b0 = -2.f;
  const double na0 = -(double)a0, na1 = -(double)a1, b0d = (double)b0;
  float m0 = g.mem_hp[2 * s], m1 = g.mem_hp[2 * s + 1];
  float *ring = g.pitch_ring + (size_t)s * RN_RING_SIZE;
  const int ring0 = RN_RING0(slot);
  const float *xring = g.xlp_ring + (size_t)s * RN_XRING_SIZE;
  // what the decimated signal needs of the undecimated one: the last sample of the previous slot (left neighbour of the slot's first
  // decimated sample) and pitch_buf[0], pitch_buf[1] (x_lp[0] has no left neighbour: src/pitch.c:166)
  const float left = ring[(slot * RN_FRAME_SIZE + RN_RING_SIZE - 1) % RN_RING_SIZE];
  const float2 pb01 = *reinterpret_cast<const float2 *>(ring + ring0);
  {  // frame -> pb[1248..] (120 float4); the 624 decimated samples older frames left in the decimated ring -> pb[0..623] (156 float4 from
     // ring0 / 2, a multiple of 16; the decimated ring's size is a multiple of 4, so a float4 never straddles the wrap)
    const float4 *x = reinterpret_cast<const float4 *>(in_row);
    const short4 *x16 = reinterpret_cast<const short4 *>(reinterpret_cast<const short *>(in) + (size_t)s * RN_FRAME_SIZE);
    constexpr int OLD4 = (RN_PITCH_BUF_SIZE - RN_FRAME_SIZE) / 2 / 4;
    float4 f[2], o[3];
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int q = min(lane + 64 * i, RN_FRAME_SIZE / 4 - 1);  // (lanes past the end re-read the last 16 bytes and drop them)
      if (in_s16) {
        const short4 v = x16[q];
        f[i] = make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w);
      } else {
        f[i] = x[q];
      }
    }
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const int q = min(lane + 64 * i, OLD4 - 1);
      int p = ring0 / 2 + 4 * q;
      p = (p >= RN_XRING_SIZE) ? p - RN_XRING_SIZE : p;
      o[i] = *reinterpret_cast<const float4 *>(xring + p);
    }
    // (stores at the clamped index too, without a branch: a lane past the end holds the last float4 and rewrites it with its own value.
    //  Behind a branch per store the compiler sank every load to its store and waited for it there: five round trips one after the other)
#pragma unroll
    for (int i = 0; i < 2; i++)
      reinterpret_cast<float4 *>(L.pb + (RN_PITCH_BUF_SIZE - RN_FRAME_SIZE))[min(lane + 64 * i, RN_FRAME_SIZE / 4 - 1)] = f[i];
#pragma unroll
    for (int i = 0; i < 3; i++) reinterpret_cast<float4 *>(L.pb)[min(lane + 64 * i, OLD4 - 1)] = o[i];
  }
  __syncthreads();
  // rnn_biquad (src/denoise.c:409-419), once per wave: every lane reads the same samples and computes the same states
  {
    float4 *yo4 = reinterpret_cast<float4 *>(L.pb + (RN_PITCH_BUF_SIZE - RN_FRAME_SIZE));
    const float4 *xi4 = yo4;  // (in place)
    constexpr int BLK = 8;
    float4 cur[BLK], nxt[BLK];
#pragma unroll
    for (int j = 0; j < BLK; j++) nxt[j] = xi4[j];
    for (int blk = 0; blk < RN_FRAME_SIZE / 4 / BLK; blk++) {
#pragma unroll
      for (int j = 0; j < BLK; j++) cur[j] = nxt[j];
      if (blk + 1 < RN_FRAME_SIZE / 4 / BLK) {
#pragma unroll
        for (int j = 0; j < BLK; j++) nxt[j] = xi4[(blk + 1) * BLK + j];
      }
#define HP_STEP(xi, yo)                                              \
      {                                                              \
        const float yi = (xi) + m0;                                  \
        const double xd = (double)(xi), yd = (double)yi;             \
        m0 = (float)((double)m1 + fma(na0, yd, b0d * xd));           \
        m1 = (float)fma(na1, yd, xd);                                \
        (yo) = yi;                                                   \
      }
#pragma unroll
      for (int j = 0; j < BLK; j++) {
        const float4 v = cur[j];
        float4 o;
        HP_STEP(v.x, o.x) HP_STEP(v.y, o.y) HP_STEP(v.z, o.z) HP_STEP(v.w, o.w)
        if (lane == ((blk * BLK + j) & 63)) yo4[blk * BLK + j] = o;  // (one writer per 16 bytes)
      }
#undef HP_STEP
    }
    if (lane == 0) {
      g.mem_hp[2 * s] = m0;
      g.mem_hp[2 * s + 1] = m1;
    }
  }
  __syncthreads();
  {  // the filtered frame -> its ring slot (coalesced); its 240 decimated samples (src/pitch.c:155-160) -> the decimated ring
     // (rn_dev.h: RN_XRING_SLOT), which the analysis kernel reads, and behind the 624 old ones in pb
    float4 *y = reinterpret_cast<float4 *>(ring + slot * RN_FRAME_SIZE);
    const float4 *src = reinterpret_cast<const float4 *>(L.pb + (RN_PITCH_BUF_SIZE - RN_FRAME_SIZE));
#pragma unroll
    for (int i = 0; i < 2; i++)
      if (lane + 64 * i < RN_FRAME_SIZE / 4) y[lane + 64 * i] = src[lane + 64 * i];
    float *y2 = g.xlp_ring + (size_t)s * RN_XRING_SIZE + slot * RN_XRING_SLOT;
    const float *nb = L.pb + (RN_PITCH_BUF_SIZE - RN_FRAME_SIZE);
    float *xlp = L.pb;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int t = lane + 64 * i;
      if (t < RN_XRING_SLOT) {
        const float v = .5f * (.5f * ((t ? nb[2 * t - 1] : left) + nb[2 * t + 1]) + nb[2 * t]);
        y2[t] = v;
        xlp[(RN_PITCH_BUF_SIZE - RN_FRAME_SIZE) / 2 + t] = v;
      }
    }
    // a listed row whose analysis runs as a four-wave workgroup gets its 5 FIR taps there, on a spare wave, beside the
    // transform of X (rn_analysis_rows_kernel): the lags, a third of this kernel's time, are not formed here (bit 8 of
    // slot_arg set by the launcher: they ARE wanted -- the one-wave analysis of $RNNOISE_AMD_ROWS_K1=1)
    if (listed && !(slot_arg & 256)) return;
    if (lane == 0) xlp[0] = .5f * (.5f * pb01.y + pb01.x);
    if (lane < 8) xlp[864 + lane] = 0;
    xlp[872 + lane] = 0;
  }
  __syncthreads();
  // 5-lag autocorrelation (src/celt_lpc.c:92-174): lane k = lag k, terms in the reference's order; terms i >= 860 form the
  // tail chain d (rnn_pitch_xcorr runs over fastN = 860 terms, the rest is added afterwards)
  float acl = 0, dl = 0;
  {
    const float *xa = L.pb, *xb = L.pb + lane;
#pragma unroll 20
    for (int i = 0; i < 860; i++) acl = acl + xa[i] * xb[i];
#pragma unroll
    for (int i = 860; i < 864; i++)
      if (i + lane <= 863) dl = dl + xb[i] * xa[i];
    acl = acl + dl;
  }
  float ac[5];
#pragma unroll
  for (int k = 0; k < 5; k++) ac[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acl), k));  // (the builtin moves ints)
  float taps[5];
  rn_fir_taps_from_ac(ac, taps);  // (lag window, Levinson, bandwidth expansion: rn_dev.h; the same on every lane)
  if (lane == 0) {
    float *o = g.lpc2 + ((size_t)slot * g.n_stride + s) * 8;
#pragma unroll
    for (int k = 0; k < 5; k++) o[k] = taps[k];
    if (RN_INSTRUMENT && g.debug) {
#pragma unroll
      for (int k = 0; k < 5; k++) g.debug[(size_t)s * RN_DBG_FLOATS + RN_DBG_AC + k] = ac[k];
    }
  }
}
extern "C" __global__ void __launch_bounds__(WAVE)
rn_hp_one_kernel(RnGroupDev g, const float *__restrict__ in, int slot_arg, int in_s16, RnRows rows) {
  __shared__ __attribute__((aligned(16))) HpOneLds L;
  // (rows: the launch groups of the one-frame API, rn_dev.h -- the block's stream, ring slot and frame buffer come from the list)
  const bool listed = rows.n > 0;
  const uint32_t re = listed ? rows.e[blockIdx.x] : 0u;
  const int s = listed ? RN_ROW_OF(re) : (int)blockIdx.x, slot = listed ? RN_ROW_RING(re) : slot_arg;
  const float *in_row = listed ? rows.io + (size_t)s * RN_ROW_IO : in + (size_t)s * RN_FRAME_SIZE;
  if (in_s16) hp_one_body<true>(L, g, in, in_row, listed, s, slot, slot_arg);
  else hp_one_body<false>(L, g, in, in_row, listed, s, slot, slot_arg);
}


// (up to RN_HP_ONE_MAX streams one wave per stream is the faster form.  With the lane = stream kernel's block loop waiting properly
// (profiles/r6_hp_specialised.txt) that kernel is one wave's latency chain of ~36 us whatever the batch, from 1,024 to 5,120 streams, and the
// one-wave form 25 / 35 / 45 / 55 / 65 us at 1,024 / 2,048 / 3,072 / 4,096 / 5,120: the switch sits at 2,048 -- alone on the machine
// (one frame per call: 0.258 -> 0.239 ms per step at 4,096 streams) and inside pipelined calls (3,072 streams: 20.8 -> 21.1 M frames/s;
// at 2,048 the one-wave form is the better one by 6 %).  Rounds 5-6 until then: 5,120 / 3,072.)
#define RN_HP_ONE_MAX 2048
#define RN_HP_ONE_MAX_PIPELINED 2048
#define RN_HP_SPW 64  // streams per wave of the lane = stream kernel
#if RN_INSTRUMENT
extern "C" __global__ void rn_hp_slp_kernel(RnGroupDev g, const float *__restrict__ in, int slot, int mode);  // hp_slp.hip
#else
#define rn_hp_slp_kernel rn_hp_kernel
#endif
extern "C" hipError_t rn_launch_hp(const RnGroupDev *g, const void *in, int in_s16, int slot, hipStream_t st, hipEvent_t e0, hipEvent_t done) {
  // bit 9 of `slot` (batch.cpp): the call is one frame of a PIPELINED multi-frame call -- this kernel then runs on a side stream beside the
  // analysis and network kernels of other frames (the two switches are kept apart because they were different ones until round 6's
  // last day, and may be again)
  static const int one_max_env = [] { const char *e = getenv("RNNOISE_AMD_HP_ONE_MAX"); return e ? atoi(e) : -1; }();  // (A/B runs)
  const bool beside_others = slot & 512;
  slot &= 511;
  const int one_max = one_max_env >= 0 ? one_max_env : (beside_others ? RN_HP_ONE_MAX_PIPELINED : RN_HP_ONE_MAX);
  if (g->n_streams <= one_max) {
    RN_LAUNCH(rn_hp_one_kernel, dim3(g->n_streams), dim3(WAVE), 0, st, e0, done, *g, static_cast<const float *>(in), slot, in_s16, RnRows{});
    return hipGetLastError();
  }
  static const int ab = [] { const char *e = RN_LAB_ENV("HP_AB"); return e ? atoi(e) & (256 | 512 | 1024) : 0; }();  // (A/B runs)
#if RN_INSTRUMENT
  static const bool slp = [] { const char *e = RN_LAB_ENV("HP_AB"); return e && (atoi(e) & 2048); }();  // (A/B: hp_slp.hip)
#else
  const bool slp = false;
#endif
  // $RNNOISE_AMD_HP_SPW = 64 | 32 | 16 streams per wave (A/B; default below)
  static const int spw_shift = [] {
    const char *e = RN_LAB_ENV("HP_SPW");
    const int v = e ? atoi(e) : RN_HP_SPW;
    return v == 16 ? 2 : (v == 32 ? 1 : 0);
  }();
  const int spw = WAVE >> spw_shift;
  // (rounds 5-6 had a second build of this kernel with 16-sample blocks at 64 VGPRs -- rn_hp_lean_kernel, a wave of which fits a SIMD
  // beside four analysis waves: +0.5 % at 65,536 streams while the autocorrelation pass re-read the whole pitch ring.  Over the decimated
  // ring the 32-sample form is the faster one inside the pipeline too (profiles/r6_xlp_ring.txt), and with the chains fed from the
  // biquad's registers the 64-register budget spills: gone.)
  static const int wpb = [] { const char *e = RN_LAB_ENV("HP_WPB"); return e && atoi(e) == 4 ? 4 : 1; }();  // (A/B: waves per workgroup)
  const int per_block = wpb > 1 ? wpb * WAVE : spw;
  RN_LAUNCH(slp ? rn_hp_slp_kernel : rn_hp_kernel, dim3((g->n_streams + per_block - 1) / per_block), dim3(wpb * WAVE),
            0, st, e0, done, *g, static_cast<const float *>(in), slot, 1 | (in_s16 ? 2 : 0) | ab | (spw_shift << 12));
  return hipGetLastError();
}
extern "C" hipError_t rn_launch_hp_passthrough(const RnGroupDev *g, const float *in, int slot, hipStream_t st) {
  hipLaunchKernelGGL(rn_hp_kernel, dim3((g->n_streams + WAVE - 1) / WAVE), dim3(WAVE), 0, st, *g, in, slot, 0);
  return hipGetLastError();
}
// K0 of a launch group of the one-frame API (rn_dev.h: RnRows): one wave per listed row, float frames from the pool's pinned blocks
extern "C" hipError_t rn_launch_hp_rows(const RnGroupDev *g, const RnRows *rows, hipStream_t st) {
  static const int taps_here = [] { const char *e = RN_LAB_ENV("ROWS_K1"); return (e && atoi(e) == 1) ? 256 : 0; }();  // (dsp_kernels.hip: rn_launch_analysis_rows)
  hipLaunchKernelGGL(rn_hp_one_kernel, dim3(rows->n), dim3(WAVE), 0, st, *g, static_cast<const float *>(nullptr), taps_here, 0, *rows);
  return hipGetLastError();
}
#endif  // RN_HP_VARIANT_ONLY
