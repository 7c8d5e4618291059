/* log10_sweep.c -- test helper: rnnoise_amd/csrc/log10_glibc.h compiled for the host against the running libm.
 *   log10_sweep <mode> <n> <seed>      prints "<n> <double mismatches> <float mismatches> <first bad argument or 0>"
 *   mode 0: x = 1e-2 + (double)Ex, Ex a float with random mantissa and an exponent drawn over the band-energy range (2^-40 .. 2^50)
 *        1: log10 of random positive normal doubles (all exponents)      2: log10 near 1 (both branches of the near-1 test)
 *        3: log() itself on random positive normal doubles               4: specials and subnormals (fixed list; n ignored)
 *        5: x = 1e-2 + (double)Ex for EVERY float Ex in [lo, hi) given as bit patterns in argv[2], argv[3] (exhaustive ranges) */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../rnnoise_amd/csrc/log10_glibc.h"

static const double tab[256] = {RN_LOG_TAB_VALUES};
static uint64_t s[2];
static uint64_t rnd(void) { /* xorshift128+ */
  uint64_t a = s[0], b = s[1];
  s[0] = b;
  a ^= a << 23;
  s[1] = a ^ b ^ (a >> 17) ^ (b >> 26);
  return s[1] + b;
}
static int same(double a, double b) { return rn_log_bits(a) == rn_log_bits(b) || (a != a && b != b); }

int main(int argc, char **argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  unsigned long long n = argc > 2 ? strtoull(argv[2], 0, 0) : 1000000, bad = 0, badf = 0;
  const unsigned long long seed = argc > 3 ? strtoull(argv[3], 0, 0) : 1;
  double first = 0;
  s[0] = 0x9e3779b97f4a7c15ull ^ seed;
  s[1] = 0xbf58476d1ce4e5b9ull + seed * 0x94d049bb133111ebull;
  for (int i = 0; i < 16; i++) rnd();
  if (mode == 4) {
    const double xs[] = {0.0, -0.0, 1.0, -1.0, 0x1p-1074, 0x1p-1023, 0x1.8p-1040, 0x1p-1022, 1.0 / 0.0, -1.0 / 0.0, 0.0 / 0.0, 1e-2, 0.5, 2.0, 10.0, 1e300,
                         0x1.fffffffffffffp1023, 0.9375, 0x1.09p0, 0x1.08fffffffffffp0, 0x1.dffffffffffffp-1};
    n = sizeof(xs) / sizeof(xs[0]);
    for (unsigned i = 0; i < n; i++)
      if (!same(rn_log10_glibc_fma(xs[i], tab), log10(xs[i]))) {
        if (!bad++) first = xs[i];
      }
    printf("%llu %llu %llu %a\n", n, bad, bad, first);
    return 0;
  }
  if (mode == 5) {
    const uint32_t lo = (uint32_t)n, hi = (uint32_t)seed;
    n = 0;
    for (uint32_t u = lo; u < hi; u++, n++) {
      float ex;
      memcpy(&ex, &u, 4);
      const double x = 1e-2 + (double)ex, got = rn_log10_glibc_fma(x, tab), want = log10(x);
      if (!same(got, want)) {
        if (!bad++) first = x;
        if ((float)got != (float)want) badf++;
      }
    }
    printf("%llu %llu %llu %a\n", n, bad, badf, first);
    return 0;
  }
  for (unsigned long long it = 0; it < n; it++) {
    const uint64_t r = rnd();
    double x, got, want;
    if (mode == 0) {
      const uint32_t e = 127 - 40 + (uint32_t)((r >> 32) % 91), u = (e << 23) | (uint32_t)(r & 0x7fffff);
      float ex;
      memcpy(&ex, &u, 4);
      x = 1e-2 + (double)ex;
    } else if (mode == 2) {
      x = rn_log_dbl(0x3fe0000000000000ull + (r % 0x0020000000000000ull));  /* [0.5, 2) */
      if (it & 1) x = rn_log_dbl(0x3fed000000000000ull + (r % 0x0005000000000000ull));  /* [0.906, 1.125): around the near-1 window */
    } else {
      x = rn_log_dbl((1ull << 52) + (r % (0x7ff0000000000000ull - (1ull << 52))));
    }
    if (mode == 3) got = rn_log_glibc_fma(x, tab), want = log(x);
    else got = rn_log10_glibc_fma(x, tab), want = log10(x);
    if (!same(got, want)) {
      if (!bad++) first = x;
      if ((float)got != (float)want) badf++;
    }
  }
  printf("%llu %llu %llu %a\n", n, bad, badf, first);
  return 0;
}
