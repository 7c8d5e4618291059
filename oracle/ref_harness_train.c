/* TRAINING=1 recording harness: the reference's feature-extraction inner loop.  TEST INFRASTRUCTURE.
 *
 * src/dump_features.c:466-491 runs rnn_frame_analysis() on the clean frame and
 * rnn_compute_frame_features() on the noisy frame of a -DTRAINING=1 build of src/denoise.c
 * (src/compile.sh) and writes features[65] | band-gain targets[32] | vad.  That loop lives inside
 * dump_features' main() between file I/O and random augmentation, so it cannot be called; this TU
 * #includes the reference's denoise.c with TRAINING=1 (REF_DENOISE_C, oracle/Makefile) and drives
 * the same two reference functions.  The only restated lines are the seven of the gain-target
 * loop (dump_features.c:472-478).
 */
#define TRAINING 1
#include REF_DENOISE_C

#include "rn_layout.h"

int lowpass = FREQ_SIZE; /* the globals dump_features.c:45-46 defines for denoise.c:328-329 */
int band_lp = NB_BANDS;

typedef struct {
  DenoiseState *st, *noisy;
} RefhT;

void *refht_create(void) {
  RefhT *h = calloc(1, sizeof *h);
  h->st = rnnoise_create(NULL);
  h->noisy = rnnoise_create(NULL);
  return h;
}

void refht_destroy(void *hv) {
  RefhT *h = hv;
  rnnoise_destroy(h->st);
  rnnoise_destroy(h->noisy);
  free(h);
}

void refht_frame(void *hv, const float *clean, const float *noisy, int lowpass_, int band_lp_, float vad_target,
                 int noise_free, float *rec98) {
  RefhT *h = hv;
  kiss_fft_cpx X[FREQ_SIZE], Y[FREQ_SIZE], P[FREQ_SIZE];
  float Ex[NB_BANDS], Ey[NB_BANDS], Ep[NB_BANDS], Exp[NB_BANDS], g[NB_BANDS];
  int i, silence;
  lowpass = lowpass_;
  band_lp = band_lp_;
  rnn_frame_analysis(h->st, Y, Ey, clean);
  silence = rnn_compute_frame_features(h->noisy, X, P, Ex, Ep, Exp, rec98, noisy);
  for (i = 0; i < NB_BANDS; i++) { /* dump_features.c:472-478 */
    g[i] = sqrt((Ey[i] + 1e-3) / (Ex[i] + 1e-3));
    if (g[i] > 1) g[i] = 1;
    if (silence || i > band_lp) g[i] = -1;
    if (Ey[i] < 5e-2 && Ex[i] < 5e-2) g[i] = -1;
    if (vad_target == 0 && noise_free) g[i] = -1;
  }
  memcpy(rec98 + NB_FEATURES, g, sizeof g);
  rec98[NB_FEATURES + NB_BANDS] = vad_target;
}
