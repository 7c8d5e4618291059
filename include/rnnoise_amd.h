/* rnnoise_amd.h -- additive batched C API of the MI355X RNNoise back end.
 *
 * The reference API (include/rnnoise.h of xiph/rnnoise, mirrored by our include/rnnoise.h)
 * is one-frame / one-stream / synchronous (rnnoise.h:94); thousands of concurrent streams
 * cannot be expressed through it.  These entry points are what a maintainer's
 * FFI would bind for the throughput path.  Plain C ABI: pointers and sizes only.
 *
 * A "batch" is N independent 48 kHz mono streams resident on one GPU, advancing in
 * lock-step one 480-sample frame per step.  Frame buffers are stream-major:
 *     in / out : [n_frames][n_streams][480] float, int16-scaled like rnnoise_demo.c:56
 *     vad      : [n_frames][n_streams]      (return value of rnnoise_process_frame)
 *     gains    : [n_frames][n_streams][32]  raw band gains (the `g[]` local of
 *                src/denoise.c:465 right after compute_rnn; 0 on silent frames), optional
 */
#ifndef RNNOISE_AMD_H
#define RNNOISE_AMD_H

#include "rnnoise.h"
#include "rn_layout.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct RNNoiseBatch RNNoiseBatch;

/* Number of visible HIP devices (0 if none / runtime unavailable). */
RNNOISE_EXPORT int rnnoise_amd_device_count(void);

/* The reference's x86 tanh / sigmoid divide through `rcpps` (src/vec_avx.h:413,442,484,505), whose low bits depend on
 * the CPU family: its output is a function of the host it runs on.  The library reproduces one family at a time from a
 * 4096-entry table (rnnoise_amd/csrc/rcp_profiles.h):
 *   "host"      (default, alias "auto") the table is captured from the CPU this process runs on, so the bits are those the
 *               reference library produces on this same machine;
 *   "intel"     Intel Xeon (the committed goldens);  "amd-zen5" (alias "amd")  AMD EPYC 9005.
 * Process-wide: $RNNOISE_AMD_RCP_PROFILE at first use, or this call at any time (devices are drained and their table
 * replaced; every model and batch follows).  0, or -1 for an unknown name.  rnnoise_amd_rcp_profile() names the active
 * profile: "intel", "amd-zen5", "host=intel", "host=amd-zen5", "host=captured" (a CPU whose table equals neither). */
RNNOISE_EXPORT int rnnoise_amd_set_rcp_profile(const char *name);
RNNOISE_EXPORT const char *rnnoise_amd_rcp_profile(void);

/* The reference's one libm call on the path is log10 of the band energies in double, rounded to float (src/denoise.c:383): which
 * double comes out is a property of the HOST's libm.  The kernels restate GNU libc's algorithm (>= 2.28, the FMA build every AVX2
 * host selects) operation for operation (rnnoise_amd/csrc/log10_glibc.h) -- bit-identical to that libm for every float band energy
 * (swept exhaustively) -- after checking at first use that this process's libm is that one.  $RNNOISE_AMD_LOG10 = host (default) |
 * glibc-fma | ocml (the device library's log10).  Names the model in use: "host=glibc-fma", "glibc-fma", "ocml", or
 * "host=unknown:ocml" (a different libm: said on stderr once). */
RNNOISE_EXPORT const char *rnnoise_amd_log10_model(void);

/* Create N zero-initialised streams on `device`, all using `model` (must outlive the
 * batch, like rnnoise_create()).  NULL on error (no GPU, bad model, out of memory).
 * model==NULL fails here (the drop-in entry points rnnoise_create / rnnoise_init fall back
 * to $RNNOISE_AMD_DEFAULT_MODEL; the batched API wants the model spelled out). */
RNNOISE_EXPORT RNNoiseBatch *rnnoise_batch_create(RNNModel *model, int n_streams, int device);
RNNOISE_EXPORT void rnnoise_batch_destroy(RNNoiseBatch *b);
RNNOISE_EXPORT int rnnoise_batch_size(const RNNoiseBatch *b);

/* Back to the state rnnoise_init() produces (all zeros). 0 / -1. */
RNNOISE_EXPORT int rnnoise_batch_reset(RNNoiseBatch *b);

/* Host buffers; synchronous.  vad and gains may be NULL.  in may alias out. 0 / -1.
 * Pinned host memory (hipHostMalloc / hipHostRegister) is read and written by DMA in place, frame by frame, through a
 * six-slot ring in HBM beside ONE pipelined multi-frame device call: a call pays one frame's upload before and one frame's
 * download after its kernels whatever its length (use 16 frames or more per call when throughput matters).  Where the runtime
 * offers two free copy engines, uploads and downloads run on two NAMED SDMA engines at once (underneath HIP; DESIGN section 4,
 * $RNNOISE_AMD_HOSTIO_COPY): 65,536 streams 30.9 M frames/s with int16 PCM, 21.2 M with floats.  Pageable
 * memory goes through the library's pinned bounce buffers in ~32 MB chunks, two in flight.  A call that fails part of the way
 * drains the device and resets the batch (every stream back to its initial state) before it returns -1. */
RNNOISE_EXPORT int rnnoise_batch_process(RNNoiseBatch *b, float *out, const float *in, float *vad, float *gains,
                                         int n_frames);

/* Device-resident buffers (same shapes, memory of the batch's device, 16-byte aligned: the
 * kernels read and write them 16 bytes per lane); asynchronous on `hip_stream` (a hipStream_t,
 * NULL = default stream).  This is the throughput path. */
RNNOISE_EXPORT int rnnoise_batch_process_device(RNNoiseBatch *b, float *d_out, const float *d_in, float *d_vad,
                                                float *d_gains, int n_frames, void *hip_stream);

/* The same two calls with 16-bit PCM at both ends: in / out : [n_frames][n_streams][480] int16.  The conversions are those
 * of the reference's only caller (examples/rnnoise_demo.c:56,58: x[i] = tmp[i] going in, tmp[i] = x[i] -- the C float -> short
 * conversion as x86 compiles it, truncation toward zero -- coming out) done inside the first and the last kernel of the step,
 * so a frame moves 2 x 960 bytes instead of 2 x 1,920 over HBM and PCIe.  Bits are those of the float calls followed by
 * that cast.  Device buffers 8-byte aligned.  The float and s16 calls may be mixed on one batch. */
RNNOISE_EXPORT int rnnoise_batch_process_s16(RNNoiseBatch *b, short *out, const short *in, float *vad, float *gains,
                                             int n_frames);
RNNOISE_EXPORT int rnnoise_batch_process_device_s16(RNNoiseBatch *b, short *d_out, const short *d_in, float *d_vad,
                                                    float *d_gains, int n_frames, void *hip_stream);

/* Portable per-stream state: RN_STATE_FLOATS 32-bit words laid out as in rn_layout.h
 * (the 25,128 live bytes of the reference's DenoiseState).  Import requires
 * analysis_mem == the last 480 samples of pitch_buf, which every state produced by the
 * reference or by export satisfies; -1 otherwise. */
RNNOISE_EXPORT int rnnoise_batch_export_state(RNNoiseBatch *b, int stream, float *state);
RNNOISE_EXPORT int rnnoise_batch_import_state(RNNoiseBatch *b, int stream, const float *state);

/* Network implementation: 0 = vector path (v_dot4 / FMA chains), 1 = batched MFMA path -- one kernel per 16-stream tile
 * below 10,240 streams, layer by layer (64 streams per GRU workgroup, five launches) from there up --, 2 = the layer-wise
 * MFMA schedule whatever the batch size (tests, A/B runs).  All produce identical bits and share all state: the path may
 * be switched between calls.  Default: 0 up to 512 streams (there path 0 runs as a latency-oriented kernel, one workgroup per
 * stream, which finishes before a 16-stream MFMA tile does), 1 beyond.
 * Returns the previous value, or -1 if unsupported. */
RNNOISE_EXPORT int rnnoise_batch_set_nn_path(RNNoiseBatch *b, int path);

/* Stream schedule of multi-frame rnnoise_batch_process_device calls: 0 = default (three-stream frame pipeline: high-pass
 * up to two frames ahead, analysis of frame t+1 beside network + synthesis of frame t), 9 = every kernel on the caller's
 * stream (stand-alone kernel timings), 1 = only the high-pass on a side stream.  Same bits in every mode.
 * Returns the previous value, or -1. */
RNNOISE_EXPORT int rnnoise_batch_set_schedule(RNNoiseBatch *b, int schedule);

/* Weight bytes one frame touches (SURVEY 8d "W"): the numerator of the HBM-roofline
 * fraction reported by bench.py. */
RNNOISE_EXPORT long rnnoise_model_weight_bytes(RNNModel *model);

/* GPU-native packed model "RNPK" (SURVEY 8f row f2): the layers already in their device layouts (int8 blocks + column
 * tables, zero-filled MFMA A-fragment order, row sums), behind a header {magic "RNPK", version, architecture dims, W,
 * per-layer offsets}.  rnnoise_model_from_buffer / _file / _filename accept a pack wherever they accept a "DNNw" blob
 * (reference format: src/write_weights.c:46-69); loading one skips the blob walk and the re-layout.
 * Returns the pack size in bytes; writes it only if cap suffices (out == NULL sizes the buffer).  -1 on a bad model.
 * Host-only: no GPU needed.  `python -m rnnoise_amd.blob pack in.blob out.rnpk` is the command-line form. */
RNNOISE_EXPORT long rnnoise_amd_model_pack(RNNModel *model, void *out, long cap);

/* Training-feature extraction (the inner loop of the reference's src/dump_features.c:466-491, a
 * TRAINING=1 build of denoise.c): per frame and stream, Ey from the CLEAN frame, the 65 features
 * from the NOISY frame (no silence short-cut), the 32 band-gain targets and the VAD target passed
 * through: records[n_frames][n_streams][98] = features | gains | vad.  Mixing, filtering and
 * augmentation of the signals stay with the caller, as in dump_features.  lowpass[n_streams] is the
 * first zeroed FFT bin (481 = none, src/denoise.c:340-343), band_lp[n_streams] the last band with a
 * valid target (32 = all), noise_free[n_streams] = (noise_gain==0 && fgnoise_gain==0).
 * The batch's per-stream analysis state tracks the noisy signal; a batch used for extraction
 * should not be mixed with rnnoise_batch_process calls.  Host-buffer and device-buffer variants. */
RNNOISE_EXPORT int rnnoise_batch_train_features(RNNoiseBatch *b, float *records, const float *clean,
                                                const float *noisy, const float *vad, const int *lowpass,
                                                const int *band_lp, const int *noise_free, int n_frames);
RNNOISE_EXPORT int rnnoise_batch_train_features_device(RNNoiseBatch *b, float *d_records, const float *d_clean,
                                                       const float *d_noisy, const float *d_vad, const int *d_lowpass,
                                                       const int *d_band_lp, const int *d_noise_free, int n_frames,
                                                       void *hip_stream);

/* Test taps for the last processed frame step: per-stream feature vectors [N][65],
 * silence flags [N] and final pitch periods [N] (host buffers, any may be NULL). */
RNNOISE_EXPORT int rnnoise_batch_debug_last(RNNoiseBatch *b, float *features, int *silence, int *pitch);

/* Average device time per launch of each kernel over the calls since the last query, in
 * milliseconds, measured with HIP events on the launch stream when timing is enabled.
 * ms[0]=analysis (K1), ms[1]=network (K2), ms[2]=synthesis (K3), ms[3]=high-pass + pitch LPC (K0). */
RNNOISE_EXPORT int rnnoise_batch_enable_timing(RNNoiseBatch *b, int on);
RNNOISE_EXPORT int rnnoise_batch_kernel_ms(RNNoiseBatch *b, double ms[4], long *launches);

#ifdef __cplusplus
}
#endif
#endif
