/* Recording harness around the UNMODIFIED reference.  TEST INFRASTRUCTURE.
 *
 * The reference exposes neither the band gains (`g[]` is a local of
 * rnnoise_process_frame, src/denoise.c:465) nor the features nor the pitch period.
 * This translation unit #includes the reference's denoise.c verbatim from
 * /root/reference (REF_DENOISE_C, set by oracle/Makefile) with two of its EXTERNAL
 * callees redirected to recording wrappers:
 *     compute_rnn          (src/rnn.c:44)      -> features in, raw gains + VAD out
 *     rnn_remove_doubling  (src/pitch.c:423)   -> final pitch period + pitch gain
 * The wrappers call the real functions, so the arithmetic executed is the
 * reference's, not ours.  Because denoise.c is in this TU, `struct DenoiseState`
 * (src/denoise.c:68-88) is visible and the live per-stream state can be exported to /
 * imported from the flat layout of oracle/rn_oracle.h (RN_STATE_FLOATS) for
 * teacher-forced tests.
 *
 * No reference code is copied here.
 */
#define compute_rnn refh_wrap_compute_rnn
#define rnn_remove_doubling refh_wrap_remove_doubling
#include REF_DENOISE_C
#undef compute_rnn
#undef rnn_remove_doubling

#include "rn_layout.h"

/* the real ones (src/rnn.h:47, src/pitch.h:44) */
void compute_rnn(const RNNoise *model, RNNState *rnn, float *gains, float *vad, const float *input, int arch);
opus_val16 rnn_remove_doubling(opus_val16 *x, int maxperiod, int minperiod, int N, int *T0,
                               int prev_period, opus_val16 prev_gain);
void rnn_fft_c(const kiss_fft_state *st, const kiss_fft_cpx *fin, kiss_fft_cpx *fout);

typedef struct {
  float features[RN_NB_FEATURES];
  float gains[RN_NB_BANDS]; /* raw network output, before the decay cap */
  float vad;
  float pitch_gain;
  int pitch;   /* period after rnn_remove_doubling */
  int silence; /* 1 = network skipped (denoise.c:389-393,474) */
} RefhRecord;

static RefhRecord *g_rec;

void refh_wrap_compute_rnn(const RNNoise *model, RNNState *rnn, float *gains, float *vad,
                           const float *input, int arch) {
  compute_rnn(model, rnn, gains, vad, input, arch);
  if (g_rec) {
    memcpy(g_rec->features, input, sizeof g_rec->features);
    memcpy(g_rec->gains, gains, sizeof g_rec->gains);
    g_rec->vad = *vad;
    g_rec->silence = 0;
  }
}

opus_val16 refh_wrap_remove_doubling(opus_val16 *x, int maxperiod, int minperiod, int N, int *T0,
                                     int prev_period, opus_val16 prev_gain) {
  opus_val16 pg = rnn_remove_doubling(x, maxperiod, minperiod, N, T0, prev_period, prev_gain);
  if (g_rec) {
    g_rec->pitch = *T0;
    g_rec->pitch_gain = pg;
  }
  return pg;
}

typedef struct {
  DenoiseState *st;
  RNNModel *model;
} Refh;

void *refh_create(const void *blob, int len) {
  Refh *h = calloc(1, sizeof *h);
  if (blob) {
    h->model = rnnoise_model_from_buffer(blob, len);
    h->model->file = NULL; /* the reference leaves this uninitialised (denoise.c:235-242) */
  }
  h->st = rnnoise_create(h->model);
  if (!h->st) {
    if (h->model) rnnoise_model_free(h->model);
    free(h);
    return NULL;
  }
  return h;
}

void refh_destroy(void *hv) {
  Refh *h = hv;
  rnnoise_destroy(h->st);
  if (h->model) rnnoise_model_free(h->model);
  free(h);
}

int refh_arch(void *hv) { return ((Refh *)hv)->st->arch; }
int refh_sizeof_state(void) { return (int)sizeof(DenoiseState); }

float refh_process(void *hv, float *out, const float *in, RefhRecord *rec) {
  Refh *h = hv;
  float vad;
  if (rec) {
    memset(rec, 0, sizeof *rec);
    rec->silence = 1;
  }
  g_rec = rec;
  vad = rnnoise_process_frame(h->st, out, in);
  g_rec = NULL;
  return vad;
}

/* ---- state <-> flat layout (rn_layout.h) ---- */
void refh_get_state(void *hv, float *f) {
  const DenoiseState *st = ((Refh *)hv)->st;
  memcpy(f + RN_OFF_ANALYSIS, st->analysis_mem, sizeof st->analysis_mem);
  memcpy(f + RN_OFF_SYNTHESIS, st->synthesis_mem, sizeof st->synthesis_mem);
  memcpy(f + RN_OFF_PITCH_BUF, st->pitch_buf, sizeof st->pitch_buf);
  f[RN_OFF_LAST_GAIN] = st->last_gain;
  memcpy(f + RN_OFF_LAST_PERIOD, &st->last_period, sizeof(int));
  memcpy(f + RN_OFF_MEM_HP, st->mem_hp_x, sizeof st->mem_hp_x);
  memcpy(f + RN_OFF_LASTG, st->lastg, sizeof st->lastg);
  memcpy(f + RN_OFF_CONV1, st->rnn.conv1_state, sizeof st->rnn.conv1_state);
  memcpy(f + RN_OFF_CONV2, st->rnn.conv2_state, sizeof st->rnn.conv2_state);
  memcpy(f + RN_OFF_GRU1, st->rnn.gru1_state, sizeof st->rnn.gru1_state);
  memcpy(f + RN_OFF_GRU2, st->rnn.gru2_state, sizeof st->rnn.gru2_state);
  memcpy(f + RN_OFF_GRU3, st->rnn.gru3_state, sizeof st->rnn.gru3_state);
  memcpy(f + RN_OFF_DELAYED_X, st->delayed_X, sizeof st->delayed_X);
  memcpy(f + RN_OFF_DELAYED_P, st->delayed_P, sizeof st->delayed_P);
  memcpy(f + RN_OFF_DELAYED_EX, st->delayed_Ex, sizeof st->delayed_Ex);
  memcpy(f + RN_OFF_DELAYED_EP, st->delayed_Ep, sizeof st->delayed_Ep);
  memcpy(f + RN_OFF_DELAYED_EXP, st->delayed_Exp, sizeof st->delayed_Exp);
}

void refh_set_state(void *hv, const float *f) {
  DenoiseState *st = ((Refh *)hv)->st;
  memcpy(st->analysis_mem, f + RN_OFF_ANALYSIS, sizeof st->analysis_mem);
  memcpy(st->synthesis_mem, f + RN_OFF_SYNTHESIS, sizeof st->synthesis_mem);
  memcpy(st->pitch_buf, f + RN_OFF_PITCH_BUF, sizeof st->pitch_buf);
  st->last_gain = f[RN_OFF_LAST_GAIN];
  memcpy(&st->last_period, f + RN_OFF_LAST_PERIOD, sizeof(int));
  memcpy(st->mem_hp_x, f + RN_OFF_MEM_HP, sizeof st->mem_hp_x);
  memcpy(st->lastg, f + RN_OFF_LASTG, sizeof st->lastg);
  memcpy(st->rnn.conv1_state, f + RN_OFF_CONV1, sizeof st->rnn.conv1_state);
  memcpy(st->rnn.conv2_state, f + RN_OFF_CONV2, sizeof st->rnn.conv2_state);
  memcpy(st->rnn.gru1_state, f + RN_OFF_GRU1, sizeof st->rnn.gru1_state);
  memcpy(st->rnn.gru2_state, f + RN_OFF_GRU2, sizeof st->rnn.gru2_state);
  memcpy(st->rnn.gru3_state, f + RN_OFF_GRU3, sizeof st->rnn.gru3_state);
  memcpy(st->delayed_X, f + RN_OFF_DELAYED_X, sizeof st->delayed_X);
  memcpy(st->delayed_P, f + RN_OFF_DELAYED_P, sizeof st->delayed_P);
  memcpy(st->delayed_Ex, f + RN_OFF_DELAYED_EX, sizeof st->delayed_Ex);
  memcpy(st->delayed_Ep, f + RN_OFF_DELAYED_EP, sizeof st->delayed_Ep);
  memcpy(st->delayed_Exp, f + RN_OFF_DELAYED_EXP, sizeof st->delayed_Exp);
}

/* ---- direct access to reference internals for stage-level known-answer tests ---- */
void refh_fft(const float *in_ri, float *out_ri) { /* 960 interleaved complex */
  rnn_fft_c(&rnn_kfft, (const kiss_fft_cpx *)in_ri, (kiss_fft_cpx *)out_ri);
}

void refh_tables(float *half_window480, float *dct1024, float *twiddles1920, int *bitrev960) {
  int i;
  memcpy(half_window480, rnn_half_window, 480 * sizeof(float));
  memcpy(dct1024, rnn_dct_table, 1024 * sizeof(float));
  for (i = 0; i < 960; i++) {
    twiddles1920[2 * i] = rnn_kfft.twiddles[i].r;
    twiddles1920[2 * i + 1] = rnn_kfft.twiddles[i].i;
    bitrev960[i] = rnn_kfft.bitrev[i];
  }
}

/* pitch front end on a caller-supplied 1728-sample buffer (denoise.c:361-368) */
float refh_pitch(const float *pitch_buf1728, int last_period, float last_gain, int *pitch_index_out,
                 float *x_lp864) {
  float buf[PITCH_BUF_SIZE];
  float lp[PITCH_BUF_SIZE >> 1];
  float *pre[1];
  int pitch_index;
  float gain;
  memcpy(buf, pitch_buf1728, sizeof buf);
  pre[0] = buf;
  rnn_pitch_downsample(pre, lp, PITCH_BUF_SIZE, 1);
  rnn_pitch_search(lp + (PITCH_MAX_PERIOD >> 1), lp, PITCH_FRAME_SIZE, PITCH_MAX_PERIOD - 3 * PITCH_MIN_PERIOD,
                   &pitch_index);
  pitch_index = PITCH_MAX_PERIOD - pitch_index;
  gain = rnn_remove_doubling(lp, PITCH_MAX_PERIOD, PITCH_MIN_PERIOD, PITCH_FRAME_SIZE, &pitch_index, last_period,
                             last_gain);
  if (x_lp864) memcpy(x_lp864, lp, sizeof lp);
  *pitch_index_out = pitch_index;
  return gain;
}
