// rn_dev.h -- device-side data model shared by the HIP kernels and the host shim.
//
// Layout in HBM (one RnGroupDev per group of N streams on one GPU):
//   every per-stream field is its own dense [N][len] fp32 array, so a wavefront that
//   owns one stream-frame reads and writes contiguous, coalesced rows, and the batched
//   network kernel sees (streams x features) row-major matrices.
//   The spectra that the reference copies into delayed_* every frame
//   (src/denoise.c:498-502) rotate through RN_SPEC_SLOTS = 3 slots instead: the analysis
//   kernel of frame t writes slot t%3, the synthesis kernel of frame t reads slot (t-1)%3
//   as "delayed" -- no copy -- and the third slot lets analysis(t+1) overlap synthesis(t).
//   analysis_mem (src/denoise.c:73) is not stored: it always equals the last 480 samples
//   of the previous pitch_buf (both are the previous high-passed frame), which sit in
//   the pitch ring (RN_RING_SLOTS = 6 slots of 480 samples = 2880 floats per stream).
#pragma once
#include <stdint.h>
// 1: the instrumented build (librnnoise_amd_instr.so) -- stage taps and shader-clock probes compiled into the kernels, probe
// kernels and the rnnoise_amd_debug.h entry points present.  0: the product library.
#ifndef RN_INSTRUMENT
#define RN_INSTRUMENT 0
#endif
#include "../../include/rn_layout.h"
#include <stdlib.h>
// The environment variables this library reads come in two classes:
//   getenv("RNNOISE_AMD_...")   product knobs: configuration, and dispatch thresholds that select between kernels with the
//                               same bits.  Every one of them is listed in INTEGRATION.md ("Environment variables");
//                               tests/test_product_surface_cpu.py compares the names found in the product .so with that list.
//   RN_LAB_ENV("...")           A/B switches, timing experiments and fault injection: read by the INSTRUMENTED build only.  In the
//                               product the macro is a null constant -- the name is not even in the binary, and no environment
//                               can steer a drop-in librnnoise.so.0 onto an experiment.
#if RN_INSTRUMENT
#define RN_LAB_ENV(name) getenv("RNNOISE_AMD_" name)
#else
#define RN_LAB_ENV(name) (static_cast<const char *>(nullptr))
#endif

// floats of one band-product array in LDS (the layout of RnTablesDev::band_q, tables.cpp: tables_for_device); the analysis
// kernel forms two band vectors at once from two arrays this far apart
#define RN_BAND_QSTRIDE 1044
#define RN_SPEC_STRIDE 964  // 481 complex = 962 floats, padded to a 16-byte multiple
// spectra slots: frame t writes slot t%3, synthesis of frame t reads slots t%3 (Ex) and (t-1)%3 (the
// reference's delayed_*); the third slot lets the analysis of frame t+1 run beside synthesis of frame t
#define RN_SPEC_SLOTS 3
// pitch ring: the 1728-sample pitch_buf (3.6 frames) lives in a ring of 480-sample slots and is never
// shifted.  Six slots rather than four: the analysis of frame t reads slots t-3..t, so the high-pass
// kernel may run up to two frames ahead (it writes slot t+1 or t+2) without touching what is being read.
#define RN_RING_SLOTS 6
#define RN_RING_SIZE (RN_RING_SLOTS * RN_FRAME_SIZE)
// ... and beside it the same ring 2x DECIMATED (src/pitch.c:155-160: x_lp[i] = .5 (.5 (x[2i-1] + x[2i+1]) + x[2i])): 240 samples per slot.
// pitch_buf[0] always sits at an even ring position, so a decimated sample is a function of three neighbouring ring samples whichever
// frame asks for it -- except x_lp[0], which has no left neighbour and is formed by its reader.  The high-pass kernel that writes a
// slot of the ring writes the slot's 240 decimated samples too (it has the filtered frame in registers); its autocorrelation pass and
// the analysis kernel then read 864 floats per frame instead of decimating 1728 again each: 3.4 KB per stream and frame less, in each.
// Derived data: rebuilt by the state scatter kernel after an import, zero after a reset (as the ring).
#define RN_XRING_SLOT (RN_FRAME_SIZE / 2)
#define RN_XRING_SIZE (RN_RING_SLOTS * RN_XRING_SLOT)
// ring position of pitch_buf[0] when the newest frame sits in `slot` (its last sample = pitch_buf[1727])
#define RN_RING0(slot) (((slot) * RN_FRAME_SIZE + RN_RING_SIZE - (RN_PITCH_BUF_SIZE - RN_FRAME_SIZE)) % RN_RING_SIZE)

struct RnTablesDev {
  const float *half_window;   // [480]   src/rnnoise_tables.c:570 (by formula)
  const float *dct;           // [32*32] src/rnnoise_tables.c:669
  const float *twiddles;      // [960*2] src/rnnoise_tables.c:77
  const float *band_frac;     // [400]   (float)j/band_size of each bin (src/denoise.c:100)
  const uint16_t *bitrev;     // [960]   digit reversal of the 5.3.4.4.4 FFT (src/rnnoise_tables.c:10, by formula), padded position
  const uint8_t *band_of_bin; // [400]   band index i with eband[i] <= bin < eband[i+1]
  const uint16_t *rcp16;      // [4096]  x86 rcpps stand-in of the active host profile (rcp_profiles.h, oracle/rcp_capture.c):
                              //         entry i = (bits(rcp(1 + i/4096)) - 0x3f000000) >> 11; see rn_rcp_bits()
  const float *fft_tw;        // [16][64][2] per-lane twiddles of the register-resident FFT (fft_reg.h: RN_FTW_*)
  const uint32_t *band_q;     // [400]  per bin: LDS slot of its (1-frac) term | slot of its frac term << 11 | band << 22
  const uint32_t *band_chain; // [34]   per band accumulator: first slot (16-byte aligned) | number of terms << 16
  const uint16_t *band_pad;   // [64]   the floats behind an accumulator's last term up to the end of its last 16-byte slot
                              //        (48 of them; the table repeats the last): they hold +0.0f while the sums are formed
  const double *log_tab;      // [128][2] {1/c, log c}: the table of the host libm's log() (log10_glibc.h) -- the feature stage then evaluates
                              //         log10 operation for operation as the reference's host does; null: the device library's log10
                              //         ($RNNOISE_AMD_LOG10=ocml, or a host whose libm is not the modelled one; tables.cpp)
  double dct_scale;           // sqrt(2./22), src/denoise.c:168
};

// one linear layer of the network, repacked for the GPU (model.cpp: stage_linear)
struct RnLinearDev {
  const float *bias;      // float layers: bias; int8 layers: subias (x86 profile, nnet_arch.h:145-147)
  const float *fw;        // float weights, column-major W[j*N + i]
  const float *fwm;       // the same in v_mfma_f32_16x16x4_f32 operand order [row tile][step/4][lane][step%4] (nout % 16 == 0), or null
  const float *scale;     // per-output scale (already /127, c_export/common.py:248)
  const float *diag;      // recurrent diagonal [3*M] or null
  const int8_t *w;        // int8 blocks, 32 bytes each = [8 rows][4 cols]
  const int8_t *wmf;      // same weights zero-filled to dense, in MFMA A-fragment order
                          //   [row tile][k tile][lane][16 B]: row = 16*rt + (lane&15), k = 64*kt + 16*(lane>>4) + byte
  const int *rowsum128;   // 128 * sum_j w[i][j]  (offset that turns s8 x s8 dots into s8 x u8)
  const int *grp_start;   // [nout/8 + 1] first block of each 8-row group
  const uint16_t *cols;   // [nblocks] first input column of each block
  // the same int8 weights once more, row-major for the vector path (model.cpp: model_on_device): a row's bytes of FOUR
  // consecutive blocks of its group are one 16-byte chunk, a group's list padded with zero blocks to a multiple of four --
  //   wrow [(((grp4[g] + c) * 8 + (row & 7)) * 4 .. + 3]       chunk c of a row of group g (grp4[g+1] - grp4[g] chunks; the eight
  //                                                            rows' chunks c are one 128-byte line)
  //   cq   [grp4[g] + c]                                       input dword (column / 4) of the chunk's four blocks, one byte each
  // so a lane fetches four blocks per load instruction instead of one (the texture addresser spends as long on a 4-byte
  // load as on a 16-byte one: it, not the cache, paced the row products)
  const int *wrow;
  const uint32_t *cq;
  const int *grp4;        // [nout/8 + 1]
  const float *fw4;       // dense_out only: the float weights as [input / 4][output][input % 4] -- a lane of the one-stream
                          //   kernel's chain wave reads the weights of four consecutive steps of ITS output as one 16-byte LDS read
  int nin, nout;
};

struct RnModelDev {
  RnLinearDev conv1, conv2, gru_in[3], gru_rec[3], dense_out, vad_dense;
};

struct RnGroupDev {
  int n_streams;   // streams this launch works on (rows 0..n_streams-1 of every array below)
  int n_stride;    // streams the arrays were laid out for: plane stride of gru_state / lpc2.  A kernel may be pointed at a
                   // sub-range of a batch (the pooled one-stream states behind rnnoise_create): every pointer advanced by
                   // first_stream * row length, n_streams = count, n_stride = the batch's size
  // persistent per-stream state
  float *mem_hp;       // [N][2]
  float *pitch_ring;   // [N][RN_RING_SIZE = 2880] ring of high-passed frames; pitch_buf (src/denoise.c:76) = its latest 1728 samples
  float *xlp_ring;     // [N][RN_XRING_SIZE = 1440] the same ring 2x decimated (derived: see RN_XRING_SLOT)
  float *synth_mem;    // [N][480]
  float *last_gain;    // [N]
  int *last_period;    // [N]
  float *lastg;        // [N][32]
  float *conv1_state;  // [N][130]
  float *conv2_state;  // [N][256]
  float *gru_state;    // [3][N][384]
  float *spec_X[RN_SPEC_SLOTS];  // [N][RN_SPEC_STRIDE]  rotating: current, delayed, (free for the next frame)
  float *spec_P[RN_SPEC_SLOTS];  // [N][RN_SPEC_STRIDE]
  float *spec_E[RN_SPEC_SLOTS];  // [N][96] = Ex | Ep | Exp
  // per-step scratch
  float *features;     // [N][68] (65 used)
  int *silence;        // [N]
  int *pitch;          // [N]   final period (debug/tests)
  float *features_b;   // second copy of the three per-step arrays above (the host alternates them per frame
  int *silence_b;      //   so that the analysis of frame t+1 may overlap network/synthesis of frame t)
  int *pitch_b;
  float *gains;        // [N][32] raw network gains of the current step
  float *vad;          // [N]
  float *lpc2;         // [RN_RING_SLOTS][N][8] (5 used) FIR taps of rnn_pitch_downsample, produced by K0, consumed by K1
  float *nn_act;       // [N][384] conv2 output in f32 (MFMA path: input of dense_out)
  int8_t *act_q[4];    // [ceil(N/16)][6144] layer-wise network: u8-quantised activations per 16-stream tile, B-fragment order:
                       //   [0] conv2 output (scratch of the step), [1 + k] GRU state k -- at once the input of layer k + 1 and
                       //   the recurrent operand of layer k at the NEXT frame, so the layer kernels never re-quantise the
                       //   f32 state.  Derived data: valid only while every state change went through the layer kernels
                       //   (RNNoiseBatch::img_valid; rn_launch_nn_requant rebuilds them).  Whole batches only, never offset by views.
  float *train_clean_mem;  // [N][480] analysis memory of the clean stream (training-feature extraction only)
  float *debug;        // [N][RN_DBG_FLOATS] pitch stage taps, or null (tests only)
};

// Row list of the one-frame API (dropin.cpp: the combiner behind rnnoise_process_frame).  Concurrent rnnoise_process_frame calls on
// states of one pool are gathered into ONE launch group: block b of the latency kernels (rn_hp_one_kernel,
// rn_analysis_rows_kernel, rn_nn_one_kernel, rn_synthesis_few_kernel) then works on pool row RN_ROW_OF(e[b]) at that row's own
// frame phase -- ring slot RN_ROW_RING(e[b]), spectra slot RN_ROW_SPEC(e[b]) -- and exchanges the frame through the row's block
// of the pool's pinned host memory: io + row * RN_ROW_IO = in[480] | pad[4] | out[480] | vad | pad[2] | done, where the last
// kernel of the group stores the request's sequence number RN_ROW_SEQ(e[b]) into `done` once frame and VAD are out.  n == 0: no
// list -- block b is stream b of the group and the launch's own arguments apply (every batched call).  Passed by value: the list
// rides in the kernel arguments, so a group costs no copy and no extra memory round trip.
// A list has at most RN_ROWS_MAX entries; a pool has up to RN_POOL_ROWS_MAX rows (round 5: 1024 instead of 64, so that a thousand
// states share ONE combiner and its three streams instead of opening a pool -- and three streams -- per 64).
#define RN_ROWS_MAX 64
#define RN_POOL_ROWS_MAX 1024
#define RN_ROW_ENTRY(row, ring, spec, seq) ((uint32_t)(row) | (uint32_t)(ring) << 10 | (uint32_t)(spec) << 13 | (uint32_t)(seq) << 16)
#define RN_ROW_OF(e) ((int)((e) & 1023u))
#define RN_ROW_RING(e) ((int)(((e) >> 10) & 7u))
#define RN_ROW_SPEC(e) ((int)(((e) >> 13) & 3u))
#define RN_ROW_SEQ(e) ((e) >> 16)
#define RN_ROW_IO 968
struct RnRows {
  float *io;
  int n;
  uint32_t e[RN_ROWS_MAX];
};

// per-step arguments of the training-feature extraction kernel (src/dump_features.c:466-491)
struct RnTrainArgs {
  const float *clean;      // [N][480] clean target frames (already filtered/scaled by the caller's mixer)
  float *clean_mem;        // [N][480] analysis memory of the clean stream (the `st` state of dump_features)
  const float *vad;        // [N] VAD targets, passed through to the record
  const int *lowpass;      // [N] first zeroed bin (denoise.c:340-343); 481 = none
  const int *band_lp;      // [N] bands above this get target -1 (dump_features.c:475); 32 = none
  const int *noise_free;   // [N] noise_gain==0 && fgnoise_gain==0 (dump_features.c:477)
  float *rec;              // [N][98] out: features[65] | gain targets[32] | vad
};

// bits(rcpps(x)) for a positive normal x with bits b, from the 16-bit table entry v = rcp16[(b >> 11) & 0xfff]:
//   (v << 11) + 0x3f000000 - ((b & 0x7f800000) - 0x3f800000)
#define RN_RCP_K 0x7e800000u
#ifdef __HIPCC__
__device__ __forceinline__ float rn_rcp_x86(float x, const uint16_t *lut16) {
  const uint32_t b = __float_as_uint(x);
  const uint32_t v = lut16[(b >> 11) & 0xfff];
  return __uint_as_float((v << 11) + (RN_RCP_K - (b & 0x7f800000u)));
}
// The 5 FIR taps of rnn_pitch_downsample from the 5 raw autocorrelation lags (src/pitch.c:176-199: lag window, order-4
// Levinson of src/celt_lpc.c:38-89, bandwidth expansion, the c1 = .8 zero): one body for the high-pass kernels, which
// compute the lags lane- or wave-per-stream, and for the one-row analysis workgroup, which computes them on a spare wave.
__device__ __forceinline__ void rn_fir_taps_from_ac(float (&ac)[5], float (&o)[5]) {
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
  o[0] = lpc[0] + .8f;
  o[1] = lpc[1] + c1 * lpc[0];
  o[2] = lpc[2] + c1 * lpc[1];
  o[3] = lpc[3] + c1 * lpc[2];
  o[4] = c1 * lpc[3];
}
#include <hip/hip_ext.h>
// Launch with optional start / stop events: they are bound to the dispatch packet itself (hipExtLaunchKernel), so
// timing a kernel or publishing its completion to another stream adds no packets to the queue.
#define RN_LAUNCH(kernel, grid, block, shmem, st, e0, e1, ...)                                          \
  do {                                                                                                  \
    if ((e0) || (e1)) hipExtLaunchKernelGGL(kernel, grid, block, shmem, st, e0, e1, 0, __VA_ARGS__);              \
    else hipLaunchKernelGGL(kernel, grid, block, shmem, st, __VA_ARGS__);                                \
  } while (0)
#endif
