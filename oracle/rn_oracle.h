/* rn_oracle.h -- CPU restatement of the RNNoise frame path.  TEST INFRASTRUCTURE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the CHECKER.  The product (rnnoise_amd/) never links it.
 *
 * Parity status: PINNED against the compiled reference.  The reference ships no golden
 * vectors; tests/test_oracle_vs_reference.py runs this restatement
 * against oracle/_ref (the reference's own sources compiled by oracle/Makefile) and
 * demands bit-identical features, pitch, gains, VAD, PCM and state; the outputs of
 * that reference build are committed under tests/golden/ so the pin travels.
 *
 * Arithmetic profile restated: the x86 RTCD/AVX2 build the reference's README
 * recommends (README:19-21), with fp contraction off (oracle/Makefile).
 */
#ifndef RN_ORACLE_H
#define RN_ORACLE_H

#include "../include/rn_layout.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct RnoModel RnoModel;

typedef struct {
  float features[RN_NB_FEATURES];
  float gains[RN_NB_BANDS]; /* raw network output (denoise.c:476), 0 on silent frames */
  float vad;
  float pitch_gain;
  int pitch;
  int silence;
} RnoRecord;

/* "DNNw" blob (src/nnet.h:41-62, parse_lpcnet_weights.c:37-78) -> model; NULL on error */
RnoModel *rno_model_from_blob(const void *blob, int len);
void rno_model_free(RnoModel *m);

void rno_state_init(float *state /* RN_STATE_FLOATS */);
float rno_process_frame(const RnoModel *m, float *state, float *out, const float *in, RnoRecord *rec);

/* training-feature extraction step (src/dump_features.c:466-491 with TRAINING=1 semantics) */
void rno_train_frame(float *st_noisy, float *clean_analysis_mem480, const float *clean, const float *noisy, int lowpass,
                     int band_lp, float vad_target, int noise_free, float *rec98);

/* stage-level entry points for known-answer tests */
void rno_fft(const float *in_ri, float *out_ri); /* 960 interleaved complex */
void rno_tables(float *half_window480, float *dct1024, float *twiddles1920, int *bitrev960);
float rno_pitch(const float *pitch_buf1728, int last_period, float last_gain, int *pitch_index_out, float *x_lp864);
float rno_pitch_debug(const float *pitch_buf1728, int last_period, float last_gain, int *pitch_index_out, float *dbg);
void rno_compute_rnn(const RnoModel *m, float *state, float *gains, float *vad, const float *features);
void rno_band_energy(float *bandE, const float *X_ri);
void rno_interp_band_gain(float *g481, const float *bandE);
float rno_rcp(float x);
int rno_set_rcp_profile(const char *name);  /* "intel" (default) | "amd-zen5" | "host"; 0 / -1 */
int rno_rcp_profile_id(void);               /* 0 intel, 1 amd-zen5, 2 other */
float rno_tanh(float x);
float rno_sigmoid(float x);
void rno_quantize_u8(unsigned char *q, const float *x, int n);
void rno_log_energy(float *out, const float *Ex, int n); /* (float)log10(1e-2 + Ex), src/denoise.c:383 */
unsigned rno_log_energy_range_diff(unsigned first_bits, unsigned n, const float *got, unsigned *first_bad);

#ifdef __cplusplus
}
#endif
#endif
