/* cpu_bench.c -- CPU baseline timer for bench.py's `cpu_baseline` leg.  TEST INFRASTRUCTURE.
 *
 * Times rnnoise_process_frame() of the REFERENCE (oracle/_ref/librnnoise_ref.so, built
 * unmodified from /root/reference by oracle/Makefile; -DUSE_REF, kind "reference") or of
 * our plain-C restatement (liboracle.so; kind "port") on the host cores: `threads`
 * pthreads, each looping over its own independent stream held in memory (no file I/O),
 * for about `seconds` of wall clock.  Prints one JSON line.
 *
 *   cpu_bench <blob> <pcm_s16_file> <threads> <seconds>
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#ifdef USE_REF
#include "rnnoise.h" /* the reference's own public header (-I/root/reference/include) */
#else
#include "rn_oracle.h"
#endif

typedef struct {
  const void *blob;
  int blob_len;
  const float *pcm;
  int n_frames;
  double seconds;
  long frames_done;
  double elapsed;
  int offset;
} Job;

static double now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static void *worker(void *arg) {
  Job *j = arg;
  float out[480];
  long done = 0;
  double t0, t;
#ifdef USE_REF
  RNNModel *m = rnnoise_model_from_buffer(j->blob, j->blob_len);
  DenoiseState *st = rnnoise_create(m);
  if (!st) { fprintf(stderr, "reference rejected the blob\n"); exit(2); }
#else
  RnoModel *m = rno_model_from_blob(j->blob, j->blob_len);
  float *st = calloc(RN_STATE_FLOATS, sizeof(float));
  if (!m) { fprintf(stderr, "oracle rejected the blob\n"); exit(2); }
#endif
  t0 = now();
  do {
    for (int k = 0; k < 50; k++) {
      const float *in = j->pcm + (size_t)((done + j->offset) % j->n_frames) * 480;
#ifdef USE_REF
      rnnoise_process_frame(st, out, in);
#else
      rno_process_frame(m, st, out, in, NULL);
#endif
      done++;
    }
    t = now();
  } while (t - t0 < j->seconds);
  j->frames_done = done;
  j->elapsed = t - t0;
  return NULL;
}

int main(int argc, char **argv) {
  if (argc < 5) { fprintf(stderr, "usage: cpu_bench blob pcm_s16 threads seconds\n"); return 1; }
  FILE *f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 1; }
  fseek(f, 0, SEEK_END);
  long blen = ftell(f);
  fseek(f, 0, SEEK_SET);
  void *blob = malloc(blen);
  if (fread(blob, 1, blen, f) != (size_t)blen) return 1;
  fclose(f);
  f = fopen(argv[2], "rb");
  if (!f) { perror(argv[2]); return 1; }
  fseek(f, 0, SEEK_END);
  long plen = ftell(f);
  fseek(f, 0, SEEK_SET);
  int n_frames = (int)(plen / 2 / 480);
  int16_t *s16 = malloc(plen);
  if (fread(s16, 1, plen, f) != (size_t)plen || n_frames < 1) return 1;
  fclose(f);
  float *pcm = malloc((size_t)n_frames * 480 * sizeof(float));
  for (long i = 0; i < (long)n_frames * 480; i++) pcm[i] = s16[i];
  int threads = atoi(argv[3]);
  double seconds = atof(argv[4]);
  pthread_t *th = malloc(threads * sizeof *th);
  Job *jobs = calloc(threads, sizeof *jobs);
  for (int i = 0; i < threads; i++) {
    jobs[i] = (Job){blob, (int)blen, pcm, n_frames, seconds, 0, 0, i * 37};
    pthread_create(&th[i], NULL, worker, &jobs[i]);
  }
  long total = 0;
  double tmax = 0;
  for (int i = 0; i < threads; i++) {
    pthread_join(th[i], NULL);
    total += jobs[i].frames_done;
    if (jobs[i].elapsed > tmax) tmax = jobs[i].elapsed;
  }
  printf("{\"frames\": %ld, \"seconds\": %.4f, \"frames_per_s\": %.1f, \"threads\": %d}\n", total, tmax, total / tmax,
         threads);
  return 0;
}
