/* BASELINE configs[0] through the drop-in API from plain C threads (no Python in the loop): T threads, each with its own
 * rnnoise_create() state, F frames each after a warm-up.  Build (no hipcc needed):
 *   gcc -O2 -Iinclude tools/configs0_mt.c -o /tmp/configs0_mt -Lrnnoise_amd -l:librnnoise_amd.so -Wl,-rpath,$PWD/rnnoise_amd -lpthread
 * usage: configs0_mt weights_blob.bin [threads = 4] [frames = 2000] */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include "rnnoise.h"

static RNNModel *model;
static int frames;
static void *worker(void *arg) {
  DenoiseState *st = rnnoise_create(model);
  float x[480];
  unsigned s = 12345u + (unsigned)(size_t)arg;
  for (int t = 0; t < frames + 100; t++) {
    for (int i = 0; i < 480; i++) { s = s * 1664525u + 1013904223u; x[i] = (float)((int)(s >> 18) - 8192); }
    rnnoise_process_frame(st, x, x);
    if (t == 99) *(double *)arg = 0;  /* (warm-up done; the caller times the whole run, warm-up included in both numerator and denominator) */
  }
  rnnoise_destroy(st);
  return NULL;
}
int main(int argc, char **argv) {
  if (argc < 2) return 2;
  const int T = argc > 2 ? atoi(argv[2]) : 4;
  frames = argc > 3 ? atoi(argv[3]) : 2000;
  model = rnnoise_model_from_filename(argv[1]);
  if (!model) { fprintf(stderr, "cannot load %s\n", argv[1]); return 1; }
  { DenoiseState *w = rnnoise_create(model); float x[480] = {0}; for (int i = 0; i < 50; i++) rnnoise_process_frame(w, x, x); rnnoise_destroy(w); }
  pthread_t th[64];
  double slot[64];
  struct timespec a, b;
  clock_gettime(CLOCK_MONOTONIC, &a);
  for (int i = 0; i < T; i++) pthread_create(&th[i], NULL, worker, &slot[i]);
  for (int i = 0; i < T; i++) pthread_join(th[i], NULL);
  clock_gettime(CLOCK_MONOTONIC, &b);
  const double dt = (b.tv_sec - a.tv_sec) + 1e-9 * (b.tv_nsec - a.tv_nsec), n = (double)T * (frames + 100);
  printf("configs[0] C threads: %d x %d frames in %.3f s = %.0f frames/s (%.1f us per frame per thread)\n", T, frames + 100, dt, n / dt, 1e6 * dt / (frames + 100));
  rnnoise_model_free(model);
  return 0;
}
