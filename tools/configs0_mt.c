/* BASELINE configs[0] through the drop-in API from plain C threads (no Python in the loop): T threads share S rnnoise_create()
 * states (S >= T; thread i walks over states i, i + T, ... round-robin, a frame at a time: a server with more streams than
 * threads), F frames per STATE after a warm-up.  Reports frames/s and the CPU time the process spent per frame (getrusage:
 * user + system, every thread) -- a synchronous caller that spins while the GPU works burns a core per thread.
 * Build (no hipcc needed):
 *   gcc -O2 -Iinclude tools/configs0_mt.c -o /tmp/configs0_mt -Lrnnoise_amd -l:librnnoise_amd.so -Wl,-rpath,$PWD/rnnoise_amd -lpthread
 * usage: configs0_mt weights_blob.bin [threads = 4] [frames per state = 2000] [states = threads] [check = 0]
 * check = 1: state k is fed the deterministic signal number k % 7 (its own generator, whatever thread calls it) and keeps a checksum
 * of every output sample and VAD value; at the end all states of one signal must hold the same checksum -- a frame lost, repeated,
 * or delivered to the wrong row under concurrency shows as a state that differs from its siblings (exit code 3). */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/resource.h>
#include <time.h>
#include "rnnoise.h"

#define MAXT 256
static RNNModel *model;
static int frames, T, S;
static DenoiseState **states;
static int check;
static unsigned *gen;               /* check mode: per-state generator ... */
static unsigned long long *sum;     /* ... and output checksum */
static void *worker(void *arg) {
  const int me = (int)(size_t)arg;
  float x[480];
  unsigned s = 12345u + (unsigned)me;
  for (int t = 0; t < frames + 100; t++)
    for (int k = me; k < S; k += T) {
      if (!check) {
        for (int i = 0; i < 480; i++) { s = s * 1664525u + 1013904223u; x[i] = (float)((int)(s >> 18) - 8192); }
        rnnoise_process_frame(states[k], x, x);
        continue;
      }
      unsigned g = gen[k];
      for (int i = 0; i < 480; i++) { g = g * 1664525u + 1013904223u; x[i] = (float)((int)(g >> 18) - 8192); }
      gen[k] = g;
      const float vad = rnnoise_process_frame(states[k], x, x);
      unsigned long long h = sum[k];
      for (int i = 0; i < 480; i++) { unsigned u; __builtin_memcpy(&u, &x[i], 4); h = (h ^ u) * 0x100000001b3ull; }
      { unsigned u; __builtin_memcpy(&u, &vad, 4); h = (h ^ u) * 0x100000001b3ull; }
      sum[k] = h;
    }
  return NULL;
}
static double cpu_seconds(void) {
  struct rusage u;
  getrusage(RUSAGE_SELF, &u);
  return u.ru_utime.tv_sec + u.ru_stime.tv_sec + 1e-6 * (u.ru_utime.tv_usec + u.ru_stime.tv_usec);
}
int main(int argc, char **argv) {
  if (argc < 2) return 2;
  T = argc > 2 ? atoi(argv[2]) : 4;
  frames = argc > 3 ? atoi(argv[3]) : 2000;
  S = argc > 4 ? atoi(argv[4]) : T;
  check = argc > 5 ? atoi(argv[5]) : 0;
  if (T < 1 || T > MAXT || S < T) { fprintf(stderr, "threads 1..%d, states >= threads\n", MAXT); return 2; }
  model = rnnoise_model_from_filename(argv[1]);
  if (!model) { fprintf(stderr, "cannot load %s\n", argv[1]); return 1; }
  states = calloc((size_t)S, sizeof *states);
  for (int k = 0; k < S; k++)
    if (!(states[k] = rnnoise_create(model))) { fprintf(stderr, "rnnoise_create failed at state %d\n", k); return 1; }
  if (!check) { float x[480] = {0}; for (int i = 0; i < 50; i++) rnnoise_process_frame(states[0], x, x); }
  gen = calloc((size_t)S, sizeof *gen);
  sum = calloc((size_t)S, sizeof *sum);
  for (int k = 0; k < S; k++) { gen[k] = 977u * (unsigned)(k % 7) + 1u; sum[k] = 0xcbf29ce484222325ull; }
  pthread_t th[MAXT];
  struct timespec a, b;
  const double c0 = cpu_seconds();
  clock_gettime(CLOCK_MONOTONIC, &a);
  for (int i = 0; i < T; i++) pthread_create(&th[i], NULL, worker, (void *)(size_t)i);
  for (int i = 0; i < T; i++) pthread_join(th[i], NULL);
  clock_gettime(CLOCK_MONOTONIC, &b);
  const double c1 = cpu_seconds();
  const double dt = (b.tv_sec - a.tv_sec) + 1e-9 * (b.tv_nsec - a.tv_nsec), n = (double)S * (frames + 100);
  printf("configs[0] C threads: %3d threads, %4d states x %d frames in %.3f s = %7.0f frames/s (%.1f us per frame per thread); CPU %.1f us per frame (%.1f cores busy)\n",
         T, S, frames + 100, dt, n / dt, 1e6 * dt * T / n, 1e6 * (c1 - c0) / n, (c1 - c0) / dt);
  int bad = 0;
  if (check) {
    for (int k = 7; k < S; k++) bad += sum[k] != sum[k % 7];
    printf("check: %d states on 7 signals, %d frames each: %s (%d states differ from the first state of their signal); checksums", S, frames + 100,
           bad ? "MISMATCH" : "every state equal to its siblings", bad);
    for (int k = 0; k < 7 && k < S; k++) printf(" %016llx", sum[k]);
    printf("\n");
  }
  for (int k = 0; k < S; k++) rnnoise_destroy(states[k]);
  free(states);
  rnnoise_model_free(model);
  return bad ? 3 : 0;
}
