/* Capture this host's `rcpps` as a 4096-entry table.  TEST/BUILD INFRASTRUCTURE.
 *
 * Why: the reference's x86 tanh/sigmoid use _mm256_rcp_ps (src/vec_avx.h:413,442), an
 * approximate reciprocal whose low bits are micro-architecture specific and whose error
 * (3e-4) is above the 1e-4 parity bar.  On the hosts measured so far it is a
 * pure function of the top mantissa bits, exponent-invariant, with the low 11 result bits zero:
 * 11 index bits on the Intel Xeon build host, 12 on the AMD EPYC 9575F (Zen 5) of the GPU boxes
 * (`--analyze`).  This tool PROVES the 12-bit form over every positive normal float whose
 * reciprocal is normal, then emits the table so that the oracle and the GPU kernels reproduce
 * the host bit-for-bit:
 *
 *     bits(rcp(x)) = (t[(bits(x) >> 11) & 0xfff] << 11) + 0x3f000000 - ((bits(x) & 0x7f800000) - 0x3f800000)
 *
 * The product carries the tables of both families (rnnoise_amd/csrc/rcp_profiles.h) and, by default, re-captures
 * the 4096 entries from the CPU it is running on at load time (tables.cpp: rcp profile "host").
 *
 * Usage: rcp_capture INTEL ../rnnoise_amd/csrc/rcp_profile_intel.h   (exit status 0 only if the model held exhaustively)
 */
#include <immintrin.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static float hw_rcp(float x) {
  float out[8];
  _mm256_storeu_ps(out, _mm256_rcp_ps(_mm256_set1_ps(x)));
  return out[0];
}

/* --analyze FILE: what does this host's rcpps depend on?  Finds the smallest K such that rcp over [1,2) is a function of
 * the top K mantissa bits, checks exponent invariance for that K over all binades, and writes the 2^23 results of the
 * [1,2) binade to FILE (raw u32) for offline modelling when no K <= 16 works. */
static int analyze(const char *path) {
  static uint32_t full[1u << 23];
  for (uint32_t m = 0; m < (1u << 23); m += 8) {
    float x[8], y[8];
    for (int k = 0; k < 8; k++) x[k] = u2f(0x3f800000u | (m + k));
    _mm256_storeu_ps(y, _mm256_rcp_ps(_mm256_loadu_ps(x)));
    for (int k = 0; k < 8; k++) full[m + k] = f2u(y[k]);
  }
  int K;
  for (K = 8; K <= 23; K++) {
    uint32_t step = 1u << (23 - K), bad = 0;
    for (uint32_t m = 0; m < (1u << 23) && !bad; m++) bad = full[m] != full[m & ~(step - 1)];
    if (!bad) break;
  }
  printf("[rcp_analyze] [1,2): result is a function of the top %d mantissa bits\n", K);
  uint32_t distinct = 1, lowbits = 0;
  for (uint32_t m = 1; m < (1u << 23); m++) { distinct += full[m] != full[m - 1]; lowbits |= full[m]; }
  printf("[rcp_analyze] [1,2): %u distinct consecutive values, OR of result bits %08x (trailing zero bits = %d)\n", distinct, lowbits, __builtin_ctz(lowbits));
  uint64_t bad = 0;
  for (uint32_t e = 2; e <= 252; e++)
    for (uint32_t m = 0; m < (1u << 23); m += 8) {
      uint32_t b[8]; float x[8], y[8];
      for (int k = 0; k < 8; k++) { b[k] = (e << 23) | (m + k); x[k] = u2f(b[k]); }
      _mm256_storeu_ps(y, _mm256_rcp_ps(_mm256_loadu_ps(x)));
      for (int k = 0; k < 8; k++) bad += f2u(y[k]) != full[b[k] & 0x7fffff] - ((b[k] & 0x7f800000u) - 0x3f800000u);
    }
  printf("[rcp_analyze] exponent invariance over binades 2..252: %llu mismatches\n", (unsigned long long)bad);
  /* SSE twin (vec_avx.h:484,505 use _mm_rcp_ps) and negative inputs */
  uint64_t bad_sse = 0, bad_neg = 0;
  for (uint32_t m = 0; m < (1u << 23); m += 4) {
    float x[4], y[4], z[4];
    for (int k = 0; k < 4; k++) x[k] = u2f(0x3f800000u | (m + k));
    _mm_storeu_ps(y, _mm_rcp_ps(_mm_loadu_ps(x)));
    for (int k = 0; k < 4; k++) { bad_sse += f2u(y[k]) != full[m + k]; x[k] = -x[k]; }
    _mm_storeu_ps(z, _mm_rcp_ps(_mm_loadu_ps(x)));
    for (int k = 0; k < 4; k++) bad_neg += f2u(z[k]) != (full[m + k] | 0x80000000u);
  }
  printf("[rcp_analyze] _mm_rcp_ps vs _mm256_rcp_ps: %llu mismatches; rcp(-x) vs -rcp(x): %llu\n", (unsigned long long)bad_sse, (unsigned long long)bad_neg);
  if (path) { FILE *f = fopen(path, "wb"); fwrite(full, 4, 1u << 23, f); fclose(f); }
  return 0;
}

/* usage:  rcp_capture NAME out.h     capture + exhaustive proof + table `RN_RCP16_<NAME>`
 *         rcp_capture --analyze [raw]  what the instruction depends on (any host) */
int main(int argc, char **argv) {
  if (argc > 1 && !strcmp(argv[1], "--analyze")) return analyze(argc > 2 ? argv[2] : NULL);
  static uint32_t lut[4096];
  uint64_t checked = 0, bad = 0;
  for (int i = 0; i < 4096; i++) lut[i] = f2u(hw_rcp(u2f(0x3f800000u | ((uint32_t)i << 11))));
  for (int i = 0; i < 4096; i++)
    if ((lut[i] & 0x7ff) || lut[i] < 0x3f000000u || lut[i] > 0x3f800000u) { fprintf(stderr, "entry %d = %08x does not fit the 16-bit form\n", i, lut[i]); return 1; }

  /* exhaustive check, 8 lanes at a time: exponents 2..252 keep 1/x normal */
  for (uint32_t e = 2; e <= 252; e++) {
    for (uint32_t m = 0; m < (1u << 23); m += 8) {
      uint32_t b[8]; float x[8], y[8];
      for (int k = 0; k < 8; k++) { b[k] = (e << 23) | (m + k); x[k] = u2f(b[k]); }
      _mm256_storeu_ps(y, _mm256_rcp_ps(_mm256_loadu_ps(x)));
      for (int k = 0; k < 8; k++) {
        uint32_t want = lut[(b[k] >> 11) & 0xfff] - ((b[k] & 0x7f800000u) - 0x3f800000u);
        checked++;
        if (f2u(y[k]) != want) {
          if (bad < 5) fprintf(stderr, "mismatch x=%08x hw=%08x model=%08x\n", b[k], f2u(y[k]), want);
          bad++;
        }
      }
    }
  }
  fprintf(stderr, "[rcp_capture] checked %llu inputs, %llu mismatches\n",
          (unsigned long long)checked, (unsigned long long)bad);
  if (bad) return 1;
  if (argc > 2) {
    FILE *f = fopen(argv[2], "w");
    char cpu[256] = "unknown";
    FILE *ci = fopen("/proc/cpuinfo", "r");
    if (ci) {
      char line[512];
      while (fgets(line, sizeof line, ci))
        if (!strncmp(line, "model name", 10)) {
          char *c = strchr(line, ':');
          if (c) { strncpy(cpu, c + 2, sizeof cpu - 1); cpu[strcspn(cpu, "\n")] = 0; }
          break;
        }
      fclose(ci);
    }
    fprintf(f, "/* GENERATED by oracle/rcp_capture.c -- do not edit.\n"
               " * rcpps of the capture host (%s): entry i = (bits(rcp(1 + i/4096)) - 0x3f000000) >> 11.\n"
               " * Verified exhaustively (%llu inputs) against the hardware instruction:\n"
               " *   bits(rcp(x)) = (entry[(bits>>11)&0xfff] << 11) + 0x3f000000 - ((bits&0x7f800000) - 0x3f800000)\n"
               " * Stands in for _mm256_rcp_ps / _mm_rcp_ps of the reference (src/vec_avx.h:413,442,484,505). */\n"
               "static const unsigned short RN_RCP16_%s[4096] = {\n", cpu, (unsigned long long)checked, argv[1]);
    for (int i = 0; i < 4096; i++) fprintf(f, "%u,%c", (lut[i] - 0x3f000000u) >> 11, (i % 16 == 15) ? '\n' : ' ');
    fprintf(f, "};\n");
    fclose(f);
  }
  return 0;
}
