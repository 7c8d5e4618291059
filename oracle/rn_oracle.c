/* rn_oracle.c -- plain-C restatement of rnnoise_process_frame().  TEST INFRASTRUCTURE.
 *
 * See rn_oracle.h for status and rules of use.  Every function cites the reference code
 * (paths relative to /root/reference) whose arithmetic it restates.  The goal is
 * bit-identity with the pinned reference build, so the ORDER of every floating-point
 * operation, every float<->double promotion the C language performs on the reference's
 * expressions, and every use of a fused multiply-add (explicit fmaf() below; the file is
 * compiled with -ffp-contract=off) is part of the specification.  The HIP kernels
 * follow this file operation by operation.
 *
 * This is a restatement, not a copy: the structure (flat state, index-based
 * butterflies, one loop nest per accumulator) is ours.
 */
#include "rn_oracle.h"

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "rcp_profiles.h"

#define NB RN_NB_BANDS
#define NFREQ RN_FREQ_SIZE
#define NWIN RN_WINDOW_SIZE
#define NFRAME RN_FRAME_SIZE

typedef struct { float r, i; } cpx;

/* band edges in 50 Hz bins: src/denoise.c:63-65 */
static const int EBAND[NB + 2] = {0,  2,  4,  6,  8,  10, 12, 15, 18,  21,  24,  28,  32,  36,  41,  47,  53,
                                  60, 68, 77, 87, 98, 110, 124, 140, 157, 176, 198, 223, 251, 282, 317, 356, 400};

/* ------------------------------------------------------------------------------------------
 * Static tables, by formula (src/dump_rnnoise_tables.c:54,84-96; src/kiss_fft.c:352-421).
 * tests/test_oracle_vs_reference.py checks them bit-for-bit against src/rnnoise_tables.c.
 * ---------------------------------------------------------------------------------------- */
static float T_WINDOW[NFRAME];
static float T_DCT[NB * NB];
static cpx T_TW[NWIN];
static int T_BITREV[NWIN];
static float T_FFT_SCALE;
static int tables_ready;

static void tables_init(void) {
  int i, j;
  if (tables_ready) return;
  for (i = 0; i < NFRAME; i++) { /* dump_rnnoise_tables.c:84 */
    double a = .5 * M_PI * (i + .5) / NFRAME;
    T_WINDOW[i] = (float)sin(.5 * M_PI * sin(a) * sin(a));
  }
  for (i = 0; i < NB; i++) /* dump_rnnoise_tables.c:91-96 */
    for (j = 0; j < NB; j++) {
      float v = (float)cos((i + .5) * j * M_PI / NB);
      if (j == 0) v = (float)(v * sqrt(.5));
      T_DCT[i * NB + j] = v;
    }
  for (i = 0; i < NWIN; i++) { /* kiss_fft.c:412-419 */
    const double pi = 3.14159265358979323846264338327;
    double phase = (-2 * pi / NWIN) * i;
    T_TW[i].r = (float)cos(phase);
    T_TW[i].i = (float)sin(phase);
  }
  /* digit reversal for radices 5,3,4,4,4 (kiss_fft.c:314-346 applied to factors
     {5,192, 3,64, 4,16, 4,4, 4,1}) */
  for (i = 0; i < NWIN; i++) {
    int j0 = i % 5, j1 = (i / 5) % 3, j2 = (i / 15) % 4, j3 = (i / 60) % 4, j4 = i / 240;
    T_BITREV[i] = j0 * 192 + j1 * 64 + j2 * 16 + j3 * 4 + j4;
  }
  T_FFT_SCALE = 1.f / NWIN; /* kiss_fft.c:452 */
  tables_ready = 1;
}

void rno_tables(float *w, float *dct, float *tw, int *br) {
  int i;
  tables_init();
  memcpy(w, T_WINDOW, sizeof T_WINDOW);
  memcpy(dct, T_DCT, sizeof T_DCT);
  for (i = 0; i < NWIN; i++) {
    tw[2 * i] = T_TW[i].r;
    tw[2 * i + 1] = T_TW[i].i;
    br[i] = T_BITREV[i];
  }
}

/* ------------------------------------------------------------------------------------------
 * FFT: src/kiss_fft.c:566-586 (scale + digit-reverse scatter), :518-564 (stage schedule),
 * butterflies :101-168 (radix 4), :173-228 (radix 3), :232-306 (radix 5).
 * complex multiply: src/_kiss_fft_guts.h:101-103.
 * ---------------------------------------------------------------------------------------- */
static inline cpx cmul(cpx a, cpx b) {
  cpx m;
  m.r = a.r * b.r - a.i * b.i;
  m.i = a.r * b.i + a.i * b.r;
  return m;
}
static inline cpx cadd(cpx a, cpx b) { cpx m = {a.r + b.r, a.i + b.i}; return m; }
static inline cpx csub(cpx a, cpx b) { cpx m = {a.r - b.r, a.i - b.i}; return m; }

/* radix-4, twiddle-free first pass: groups of 4 consecutive points (kiss_fft.c:112-131) */
static void bfly4_first(cpx *F) {
  cpx s0 = csub(F[0], F[2]);
  cpx f0 = cadd(F[0], F[2]);
  cpx s1 = cadd(F[1], F[3]);
  cpx f2 = csub(f0, s1);
  cpx d;
  f0 = cadd(f0, s1);
  d = csub(F[1], F[3]);
  F[0] = f0;
  F[2] = f2;
  F[1].r = s0.r + d.i;
  F[1].i = s0.i - d.r;
  F[3].r = s0.r - d.i;
  F[3].i = s0.i + d.r;
}

/* one radix-4 butterfly on F[0], F[m], F[2m], F[3m] with twiddle index step ts*j (kiss_fft.c:141-165) */
static void bfly4_one(cpx *F, int m, int tw) {
  cpx s0 = cmul(F[m], T_TW[tw]);
  cpx s1 = cmul(F[2 * m], T_TW[2 * tw]);
  cpx s2 = cmul(F[3 * m], T_TW[3 * tw]);
  cpx s5 = csub(F[0], s1);
  cpx f0 = cadd(F[0], s1);
  cpx s3 = cadd(s0, s2);
  cpx s4 = csub(s0, s2);
  F[2 * m] = csub(f0, s3);
  F[0] = cadd(f0, s3);
  F[m].r = s5.r + s4.i;
  F[m].i = s5.i - s4.r;
  F[3 * m].r = s5.r - s4.i;
  F[3 * m].i = s5.i + s4.r;
}

/* one radix-3 butterfly (kiss_fft.c:201-225); epi3i = twiddles[fstride*m].i */
static void bfly3_one(cpx *F, int m, int tw, float epi3i) {
  cpx s1 = cmul(F[m], T_TW[tw]);
  cpx s2 = cmul(F[2 * m], T_TW[2 * tw]);
  cpx s3 = cadd(s1, s2);
  cpx s0 = csub(s1, s2);
  cpx fm;
  fm.r = F[0].r - s3.r * .5f;
  fm.i = F[0].i - s3.i * .5f;
  s0.r *= epi3i;
  s0.i *= epi3i;
  F[0] = cadd(F[0], s3);
  F[2 * m].r = fm.r + s0.i;
  F[2 * m].i = fm.i - s0.r;
  F[m].r = fm.r - s0.i;
  F[m].i = fm.i + s0.r;
}

/* one radix-5 butterfly (kiss_fft.c:269-302) */
static void bfly5_one(cpx *F, int m, int tw, cpx ya, cpx yb) {
  cpx s0 = F[0];
  cpx s1 = cmul(F[m], T_TW[tw]);
  cpx s2 = cmul(F[2 * m], T_TW[2 * tw]);
  cpx s3 = cmul(F[3 * m], T_TW[3 * tw]);
  cpx s4 = cmul(F[4 * m], T_TW[4 * tw]);
  cpx s7 = cadd(s1, s4), s10 = csub(s1, s4);
  cpx s8 = cadd(s2, s3), s9 = csub(s2, s3);
  cpx s5, s6, s11, s12;
  F[0].r = F[0].r + (s7.r + s8.r);
  F[0].i = F[0].i + (s7.i + s8.i);
  s5.r = s0.r + (s7.r * ya.r + s8.r * yb.r);
  s5.i = s0.i + (s7.i * ya.r + s8.i * yb.r);
  s6.r = s10.i * ya.i + s9.i * yb.i;
  s6.i = -(s10.r * ya.i + s9.r * yb.i);
  F[m] = csub(s5, s6);
  F[4 * m] = cadd(s5, s6);
  s11.r = s0.r + (s7.r * yb.r + s8.r * ya.r);
  s11.i = s0.i + (s7.i * yb.r + s8.i * ya.r);
  s12.r = s9.i * ya.i - s10.i * yb.i;
  s12.i = s10.r * yb.i - s9.r * ya.i;
  F[2 * m] = cadd(s11, s12);
  F[3 * m] = csub(s11, s12);
}

static void fft960(const cpx *in, cpx *F) {
  int i, j;
  tables_init();
  for (i = 0; i < NWIN; i++) { /* kiss_fft.c:577-582 */
    F[T_BITREV[i]].r = T_FFT_SCALE * in[i].r;
    F[T_BITREV[i]].i = T_FFT_SCALE * in[i].i;
  }
  /* stage schedule for factors {5,192,3,64,4,16,4,4,4,1}, last factor first (kiss_fft.c:539-563) */
  for (i = 0; i < 240; i++) bfly4_first(F + 4 * i);                              /* m=1            */
  for (i = 0; i < 60; i++) for (j = 0; j < 4; j++) bfly4_one(F + 16 * i + j, 4, 60 * j);   /* m=4,  fstride 60 */
  for (i = 0; i < 15; i++) for (j = 0; j < 16; j++) bfly4_one(F + 64 * i + j, 16, 15 * j); /* m=16, fstride 15 */
  for (i = 0; i < 5; i++) for (j = 0; j < 64; j++) bfly3_one(F + 192 * i + j, 64, 5 * j, T_TW[5 * 64].i);
  for (j = 0; j < 192; j++) bfly5_one(F + j, 192, j, T_TW[192], T_TW[384]);
}

void rno_fft(const float *in_ri, float *out_ri) { fft960((const cpx *)in_ri, (cpx *)out_ri); }

/* src/denoise.c:219-225 */
static void apply_window(float *x) {
  int i;
  for (i = 0; i < NFRAME; i++) {
    x[i] *= T_WINDOW[i];
    x[NWIN - 1 - i] *= T_WINDOW[i];
  }
}

/* real input -> bins 0..480 (src/denoise.c:186-198) */
static void forward_transform(cpx *out, const float *in) {
  cpx x[NWIN], y[NWIN];
  int i;
  for (i = 0; i < NWIN; i++) { x[i].r = in[i]; x[i].i = 0; }
  fft960(x, y);
  memcpy(out, y, NFREQ * sizeof(cpx));
}

/* src/denoise.c:200-217: Hermitian extension, FORWARD transform, index-reversed read-out */
static void inverse_transform(float *out, const cpx *in) {
  cpx x[NWIN], y[NWIN];
  int i;
  for (i = 0; i < NFREQ; i++) x[i] = in[i];
  for (; i < NWIN; i++) { x[i].r = x[NWIN - i].r; x[i].i = -x[NWIN - i].i; }
  fft960(x, y);
  out[0] = NWIN * y[0].r;
  for (i = 1; i < NWIN; i++) out[i] = NWIN * y[NWIN - i].r;
}

/* ------------------------------------------------------------------------------------------
 * Band energies / correlations: src/denoise.c:90-138.  Written per accumulator: sum[k]
 * receives band k-1's `frac` parts in bin order, then band k's `1-frac` parts in bin order --
 * exactly the sequence the reference's interleaved loop produces for each sum[].
 * ---------------------------------------------------------------------------------------- */
static void band_accumulate(float *bandE, const cpx *X, const cpx *P) {
  float sum[NB + 2];
  int k, j;
  for (k = 0; k < NB + 2; k++) {
    float s = 0;
    if (k >= 1) {
      int bs = EBAND[k] - EBAND[k - 1];
      for (j = 0; j < bs; j++) {
        float frac = (float)j / bs;
        cpx a = X[EBAND[k - 1] + j], b = P[EBAND[k - 1] + j];
        float tmp = a.r * b.r;
        tmp += a.i * b.i;
        s += frac * tmp;
      }
    }
    if (k <= NB) {
      int bs = EBAND[k + 1] - EBAND[k];
      for (j = 0; j < bs; j++) {
        float frac = (float)j / bs;
        cpx a = X[EBAND[k] + j], b = P[EBAND[k] + j];
        float tmp = a.r * b.r;
        tmp += a.i * b.i;
        s += (1 - frac) * tmp;
      }
    }
    sum[k] = s;
  }
  sum[1] = (sum[0] + sum[1]) * 2 / 3;
  sum[NB] = (sum[NB] + sum[NB + 1]) * 2 / 3;
  for (k = 0; k < NB; k++) bandE[k] = sum[k + 1];
}
static void compute_band_energy(float *bandE, const cpx *X) { band_accumulate(bandE, X, X); }
static void compute_band_corr(float *bandE, const cpx *X, const cpx *P) { band_accumulate(bandE, X, P); }
void rno_band_energy(float *bandE, const float *X_ri) { compute_band_energy(bandE, (const cpx *)X_ri); }

/* src/denoise.c:140-154; bins 400..480 stay 0 for every caller */
static void interp_band_gain(float *g, const float *bandE) {
  int i, j;
  for (j = 0; j < NFREQ; j++) g[j] = 0;
  for (i = 1; i < NB; i++) {
    int bs = EBAND[i + 1] - EBAND[i];
    for (j = 0; j < bs; j++) {
      float frac = (float)j / bs;
      g[EBAND[i] + j] = (1 - frac) * bandE[i - 1] + frac * bandE[i];
    }
  }
  for (j = 0; j < EBAND[1]; j++) g[j] = bandE[0];
  for (j = EBAND[NB]; j < EBAND[NB + 1]; j++) g[j] = bandE[NB - 1];
}
void rno_interp_band_gain(float *g, const float *bandE) { interp_band_gain(g, bandE); }

/* src/denoise.c:160-170: float accumulate, final scale in double */
static void dct(float *out, const float *in) {
  int i, j;
  for (i = 0; i < NB; i++) {
    float sum = 0;
    for (j = 0; j < NB; j++) sum += in[j] * T_DCT[j * NB + i];
    out[i] = (float)(sum * sqrt(2. / 22));
  }
}

/* ------------------------------------------------------------------------------------------
 * Pitch analysis: src/pitch.c, src/pitch.h, src/celt_lpc.c (float build: src/arch.h:150-251,
 * MAC16_16(c,a,b) = c + a*b unfused, HALF32(x) = .5f*x).
 * Every dot product is one serial chain over the sample index (pitch.h:51-142).
 * ---------------------------------------------------------------------------------------- */
/* optional stage taps for tests (layout: include/rn_layout.h RN_DBG_*) */
static float *g_dbg;
#define DBG(off, val) do { if (g_dbg) g_dbg[(off)] = (float)(val); } while (0)

static float inner_prod(const float *x, const float *y, int n) {
  float s = 0;
  int i;
  for (i = 0; i < n; i++) s = s + x[i] * y[i];
  return s;
}

/* src/pitch.c:146-214 with C==1; autocorr src/celt_lpc.c:92-174; LPC src/celt_lpc.c:38-89;
   FIR src/pitch.c:104-143 */
static void pitch_downsample(const float *x, float *x_lp) {
  const int n = RN_PITCH_BUF_SIZE >> 1; /* 864 */
  float ac[5], lpc[4], lpc2[5];
  float tmp = 1.f, c1 = .8f;
  int i, j, k;
  for (i = 1; i < n; i++) x_lp[i] = .5f * (.5f * (x[2 * i - 1] + x[2 * i + 1]) + x[2 * i]);
  x_lp[0] = .5f * (.5f * (x[1]) + x[0]);

  /* rnn_autocorr(x_lp, ac, NULL, 0, 4, 864): lags 0..4 over the first 860 samples, then tails */
  for (k = 0; k <= 4; k++) {
    float d = 0;
    ac[k] = inner_prod(x_lp, x_lp + k, n - 4);
    for (i = k + n - 4; i < n; i++) d = d + x_lp[i] * x_lp[i - k];
    ac[k] += d;
  }
  ac[0] *= 1.0001f;
  for (i = 1; i <= 4; i++) ac[i] -= ac[i] * (.008f * i) * (.008f * i);

  /* Levinson-Durbin, order 4 (celt_lpc.c:38-89) */
  for (i = 0; i < 4; i++) lpc[i] = 0;
  if (ac[0] != 0) {
    float error = ac[0];
    for (i = 0; i < 4; i++) {
      float rr = 0, r;
      for (j = 0; j < i; j++) rr += lpc[j] * ac[i - j];
      rr += ac[i + 1];
      r = -rr / error;
      lpc[i] = r;
      for (j = 0; j < (i + 1) >> 1; j++) {
        float t1 = lpc[j], t2 = lpc[i - 1 - j];
        lpc[j] = t1 + r * t2;
        lpc[i - 1 - j] = t2 + r * t1;
      }
      error = error - (r * r) * error;
      if (error < .001f * ac[0]) break;
    }
  }
  for (i = 0; i < 4; i++) {
    tmp = .9f * tmp;
    lpc[i] = lpc[i] * tmp;
  }
  lpc2[0] = lpc[0] + .8f;
  lpc2[1] = lpc[1] + c1 * lpc[0];
  lpc2[2] = lpc[2] + c1 * lpc[1];
  lpc2[3] = lpc[3] + c1 * lpc[2];
  lpc2[4] = c1 * lpc[3];
  for (i = 0; i < 5; i++) { DBG(RN_DBG_AC + i, ac[i]); DBG(RN_DBG_LPC + i, lpc2[i]); }
  { /* celt_fir5, in place, zero initial memory */
    float m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0;
    for (i = 0; i < n; i++) {
      float xi = x_lp[i];
      float sum = xi;
      sum = sum + lpc2[0] * m0;
      sum = sum + lpc2[1] * m1;
      sum = sum + lpc2[2] * m2;
      sum = sum + lpc2[3] * m3;
      sum = sum + lpc2[4] * m4;
      m4 = m3; m3 = m2; m2 = m1; m1 = m0; m0 = xi;
      x_lp[i] = sum;
    }
  }
}

/* src/pitch.c:44-102 (float build) */
static void find_best_pitch(const float *xcorr, const float *y, int len, int max_pitch, int *best_pitch) {
  float Syy = 1;
  float best_num[2] = {-1, -1}, best_den[2] = {0, 0};
  int i, j;
  best_pitch[0] = 0;
  best_pitch[1] = 1;
  for (j = 0; j < len; j++) Syy = Syy + y[j] * y[j];
  for (i = 0; i < max_pitch; i++) {
    if (xcorr[i] > 0) {
      float x16 = xcorr[i];
      float num;
      x16 *= 1e-12f;
      num = x16 * x16;
      if (num * best_den[1] > best_num[1] * Syy) {
        if (num * best_den[0] > best_num[0] * Syy) {
          best_num[1] = best_num[0];
          best_den[1] = best_den[0];
          best_pitch[1] = best_pitch[0];
          best_num[0] = num;
          best_den[0] = Syy;
          best_pitch[0] = i;
        } else {
          best_num[1] = num;
          best_den[1] = Syy;
          best_pitch[1] = i;
        }
      }
    }
    Syy += y[i + len] * y[i + len] - y[i] * y[i];
    Syy = (1 > Syy) ? 1 : Syy;
  }
}

/* src/pitch.c:281-385 with len=960, max_pitch=588 (called from denoise.c:363) */
static int pitch_search(const float *x_lp, const float *y) {
  enum { LEN = RN_PITCH_FRAME_SIZE, MAXP = RN_PITCH_MAX_PERIOD - 3 * RN_PITCH_MIN_PERIOD };
  float x4[LEN >> 2], y4[(LEN + MAXP) >> 2], xcorr[MAXP >> 1];
  int best[2] = {0, 0};
  int i, j, offset;
  for (j = 0; j < LEN >> 2; j++) x4[j] = x_lp[2 * j];
  for (j = 0; j < (LEN + MAXP) >> 2; j++) y4[j] = y[2 * j];
  for (i = 0; i < MAXP >> 2; i++) xcorr[i] = inner_prod(x4, y4 + i, LEN >> 2);
  find_best_pitch(xcorr, y4, LEN >> 2, MAXP >> 2, best);
  for (i = 0; i < MAXP >> 2; i++) DBG(RN_DBG_XC_COARSE + i, xcorr[i]);
  DBG(RN_DBG_BEST + 0, best[0]); DBG(RN_DBG_BEST + 1, best[1]);
  for (i = 0; i < MAXP >> 1; i++) {
    float sum;
    xcorr[i] = 0;
    if (abs(i - 2 * best[0]) > 2 && abs(i - 2 * best[1]) > 2) continue;
    sum = inner_prod(x_lp, y + i, LEN >> 1);
    xcorr[i] = (-1 > sum) ? -1 : sum;
  }
  find_best_pitch(xcorr, y, LEN >> 1, MAXP >> 1, best);
  for (i = 0; i < MAXP >> 1; i++) DBG(RN_DBG_XC_FINE + i, xcorr[i]);
  DBG(RN_DBG_BEST + 2, best[0]); DBG(RN_DBG_BEST + 3, best[1]);
  if (best[0] > 0 && best[0] < (MAXP >> 1) - 1) {
    float a = xcorr[best[0] - 1], b = xcorr[best[0]], c = xcorr[best[0] + 1];
    if ((c - a) > .7f * (b - a)) offset = 1;
    else if ((a - c) > .7f * (b - c)) offset = -1;
    else offset = 0;
  } else {
    offset = 0;
  }
  DBG(RN_DBG_BEST + 4, offset);
  return 2 * best[0] - offset;
}

/* src/pitch.c:416-419: product and +1 in float, sqrt and division in double */
static float pitch_gain(float xy, float xx, float yy) { return (float)(xy / sqrt(1 + xx * yy)); }

/* src/pitch.c:422-528 with maxperiod=768, minperiod=60, N=960 (denoise.c:367) */
static float remove_doubling(const float *xbuf, int *T0_, int prev_period, float prev_gain) {
  static const int second_check[16] = {0, 0, 3, 2, 3, 2, 5, 2, 3, 2, 3, 2, 5, 2, 3, 2};
  const int minperiod0 = RN_PITCH_MIN_PERIOD;
  const int maxperiod = RN_PITCH_MAX_PERIOD / 2, minperiod = RN_PITCH_MIN_PERIOD / 2, N = RN_PITCH_FRAME_SIZE / 2;
  const float *x = xbuf + maxperiod;
  float yy_lookup[RN_PITCH_MAX_PERIOD / 2 + 1];
  float xx, xy, xy2, yy, g, g0, pg, best_xy, best_yy, xc[3];
  int k, i, T, T0, offset;
  *T0_ /= 2;
  prev_period /= 2;
  if (*T0_ >= maxperiod) *T0_ = maxperiod - 1;
  T = T0 = *T0_;
  xx = 0; xy = 0;
  for (i = 0; i < N; i++) { /* dual_inner_prod(x, x, x-T0) */
    xx = xx + x[i] * x[i];
    xy = xy + x[i] * x[i - T0];
  }
  yy_lookup[0] = xx;
  yy = xx;
  for (i = 1; i <= maxperiod; i++) {
    yy = yy + x[-i] * x[-i] - x[N - i] * x[N - i];
    yy_lookup[i] = (0 > yy) ? 0 : yy;
  }
  yy = yy_lookup[T0];
  DBG(RN_DBG_DOTS + 0, xx); DBG(RN_DBG_DOTS + 1, xy); DBG(RN_DBG_DOTS + 2, yy);
  best_xy = xy;
  best_yy = yy;
  g = g0 = pitch_gain(xy, xx, yy);
  for (k = 2; k <= 15; k++) {
    int T1, T1b;
    float g1, cont, thresh;
    T1 = (2 * T0 + k) / (2 * k);
    if (T1 < minperiod) break;
    if (k == 2) {
      if (T1 + T0 > maxperiod) T1b = T0;
      else T1b = T0 + T1;
    } else {
      T1b = (2 * second_check[k] * T0 + k) / (2 * k);
    }
    xy = 0; xy2 = 0;
    for (i = 0; i < N; i++) {
      xy = xy + x[i] * x[i - T1];
      xy2 = xy2 + x[i] * x[i - T1b];
    }
    xy = .5f * (xy + xy2);
    yy = .5f * (yy_lookup[T1] + yy_lookup[T1b]);
    g1 = pitch_gain(xy, xx, yy);
    if (abs(T1 - prev_period) <= 1) cont = prev_gain;
    else if (abs(T1 - prev_period) <= 2 && 5 * k * k < T0) cont = .5f * prev_gain;
    else cont = 0;
    thresh = (.3f > .7f * g0 - cont) ? .3f : .7f * g0 - cont;
    if (T1 < 3 * minperiod) thresh = (.4f > .85f * g0 - cont) ? .4f : .85f * g0 - cont;
    else if (T1 < 2 * minperiod) thresh = (.5f > .9f * g0 - cont) ? .5f : .9f * g0 - cont;
    if (g1 > thresh) {
      best_xy = xy;
      best_yy = yy;
      T = T1;
      g = g1;
    }
  }
  best_xy = (0 > best_xy) ? 0 : best_xy;
  if (best_yy <= best_xy) pg = 1.f;
  else pg = best_xy / (best_yy + 1);
  for (k = 0; k < 3; k++) xc[k] = inner_prod(x, x - (T + k - 1), N);
  if ((xc[2] - xc[0]) > .7f * (xc[1] - xc[0])) offset = 1;
  else if ((xc[0] - xc[2]) > .7f * (xc[1] - xc[2])) offset = -1;
  else offset = 0;
  if (pg > g) pg = g;
  DBG(RN_DBG_DOTS + 3, T); DBG(RN_DBG_DOTS + 4, xc[0]); DBG(RN_DBG_DOTS + 5, xc[1]); DBG(RN_DBG_DOTS + 6, xc[2]);
  *T0_ = 2 * T + offset;
  if (*T0_ < minperiod0) *T0_ = minperiod0;
  return pg;
}

/* the pitch front end as driven by src/denoise.c:361-368 */
float rno_pitch(const float *pitch_buf, int last_period, float last_gain, int *pitch_index_out, float *x_lp_out) {
  float lp[RN_PITCH_BUF_SIZE >> 1];
  int pitch_index;
  float gain;
  tables_init();
  pitch_downsample(pitch_buf, lp);
  if (g_dbg) memcpy(g_dbg + RN_DBG_XLP, lp, sizeof lp);
  pitch_index = pitch_search(lp + (RN_PITCH_MAX_PERIOD >> 1), lp);
  pitch_index = RN_PITCH_MAX_PERIOD - pitch_index;
  DBG(RN_DBG_BEST + 5, pitch_index);
  gain = remove_doubling(lp, &pitch_index, last_period, last_gain);
  if (x_lp_out) memcpy(x_lp_out, lp, sizeof lp);
  *pitch_index_out = pitch_index;
  return gain;
}

/* same as rno_pitch, additionally filling the RN_DBG_FLOATS stage-tap record */
float rno_pitch_debug(const float *pitch_buf, int last_period, float last_gain, int *pitch_index_out, float *dbg) {
  float r;
  memset(dbg, 0, RN_DBG_FLOATS * sizeof(float));
  g_dbg = dbg;
  r = rno_pitch(pitch_buf, last_period, last_gain, pitch_index_out, NULL);
  g_dbg = NULL;
  return r;
}

/* ------------------------------------------------------------------------------------------
 * Network.  Model container: src/nnet.h:65-75; blob: src/nnet.h:41-62,
 * src/parse_lpcnet_weights.c:37-176, record names from the exporter
 * (torch/weight-exchange/wexchange/c_export/common.py:194-258,328-364).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  const float *bias, *subias, *fw, *diag, *scale;
  const int8_t *w;
  const int *idx;
  int nin, nout;
} Lin;

struct RnoModel {
  void *blob;
  Lin conv1, conv2, gru_in[3], gru_rec[3], dense_out, vad_dense;
};

typedef struct { char head[4]; int version, type, size, block_size; char name[44]; } BlobHead;

static const void *blob_find(const unsigned char *b, int len, const char *name, int want_size) {
  int off = 0;
  while (len - off >= 64) {
    const BlobHead *h = (const BlobHead *)(b + off);
    if (h->block_size < h->size || h->size < 0 || h->block_size > len - off - 64) return NULL;
    if (h->name[43] != 0) return NULL;
    if (!strcmp(h->name, name)) return (want_size < 0 || h->size == want_size) ? b + off + 64 : NULL;
    off += 64 + h->block_size;
  }
  return NULL;
}

static int blob_size(const unsigned char *b, int len, const char *name) {
  int off = 0;
  while (len - off >= 64) {
    const BlobHead *h = (const BlobHead *)(b + off);
    if (h->block_size < h->size || h->size < 0 || h->block_size > len - off - 64) return -1;
    if (!strcmp(h->name, name)) return h->size;
    off += 64 + h->block_size;
  }
  return -1;
}

/* mirrors the checks of linear_init (parse_lpcnet_weights.c:123-176) for the layer kinds we have */
static int lin_init(Lin *l, const unsigned char *b, int len, const char *layer, int nin, int nout, int kind) {
  char nm[64];
  memset(l, 0, sizeof *l);
  l->nin = nin;
  l->nout = nout;
#define GET(field, suffix, size) \
  do { strcpy(nm, layer); strcat(nm, suffix); l->field = blob_find(b, len, nm, size); if (!l->field) return 1; } while (0)
  GET(bias, "_bias", nout * 4);
  if (kind == 0) { /* float dense */
    GET(fw, "_weights_float", nin * nout * 4);
    return 0;
  }
  GET(subias, "_subias", nout * 4);
  GET(scale, "_scale", nout * 4);
  if (kind == 1) { /* dense int8 */
    GET(w, "_weights_int8", nin * nout);
    return 0;
  }
  { /* block-sparse int8: validate the index stream (parse_lpcnet_weights.c:98-121) */
    int isz, remain, total = 0, rows = nout;
    const int *idx;
    strcpy(nm, layer); strcat(nm, "_weights_idx");
    isz = blob_size(b, len, nm);
    if (isz < 0) return 1;
    l->idx = blob_find(b, len, nm, -1);
    idx = l->idx;
    remain = isz / 4;
    while (remain > 0) {
      int nb = *idx++, i;
      if (remain < nb + 1) return 1;
      for (i = 0; i < nb; i++) {
        int pos = *idx++;
        if (pos + 3 >= nin || (pos & 3)) return 1;
      }
      rows -= 8;
      remain -= nb + 1;
      total += nb;
    }
    if (rows != 0) return 1;
    GET(w, "_weights_int8", 32 * total);
  }
  if (kind == 3) GET(diag, "_weights_diag", nout * 4);
#undef GET
  return 0;
}

RnoModel *rno_model_from_blob(const void *blob, int len) {
  RnoModel *m = calloc(1, sizeof *m);
  const unsigned char *b;
  int k, err = 0;
  m->blob = malloc(len);
  memcpy(m->blob, blob, len);
  b = m->blob;
  err |= lin_init(&m->conv1, b, len, "conv1", RN_CONV1_K, RN_CONV1_OUT, 0);
  err |= lin_init(&m->conv2, b, len, "conv2", RN_CONV2_K, RN_CONV2_OUT, 1);
  for (k = 0; k < 3 && !err; k++) {
    char nm[32];
    strcpy(nm, "gruX_input"); nm[3] = '1' + k;
    err |= lin_init(&m->gru_in[k], b, len, nm, RN_GRU, RN_GRU3, 2);
    strcpy(nm, "gruX_recurrent"); nm[3] = '1' + k;
    err |= lin_init(&m->gru_rec[k], b, len, nm, RN_GRU, RN_GRU3, 3);
  }
  err |= lin_init(&m->dense_out, b, len, "dense_out", RN_CAT, RN_NB_BANDS, 0);
  err |= lin_init(&m->vad_dense, b, len, "vad_dense", RN_CAT, 1, 0);
  if (err) { rno_model_free(m); return NULL; }
  tables_init();
  return m;
}

void rno_model_free(RnoModel *m) {
  if (!m) return;
  free(m->blob);
  free(m);
}

/* rcpps stand-in, see rcp_profiles.h (src/vec_avx.h:413,442 use _mm256_rcp_ps; :484,505 _mm_rcp_ps).  The profile is
   process-wide: "intel" (default here: the committed goldens were made on the Intel build host), "amd-zen5", or "host"
   = captured from the CPU the oracle runs on, which is what a live comparison against oracle/_ref needs. */
static unsigned short rcp_host_table[RN_RCP_ENTRIES];
static const unsigned short *rcp_table = RN_RCP16_INTEL;
int rno_set_rcp_profile(const char *name) {
  if (!strcmp(name, "intel")) rcp_table = RN_RCP16_INTEL;
  else if (!strcmp(name, "amd-zen5")) rcp_table = RN_RCP16_AMD_ZEN5;
  else if (!strcmp(name, "host")) {
    if (rn_rcp_capture_host(rcp_host_table)) return -1;
    rcp_table = rcp_host_table;
  } else return -1;
  return 0;
}
/* 0 = intel, 1 = amd-zen5, 2 = another table */
int rno_rcp_profile_id(void) {
  if (!memcmp(rcp_table, RN_RCP16_INTEL, sizeof rcp_host_table)) return 0;
  if (!memcmp(rcp_table, RN_RCP16_AMD_ZEN5, sizeof rcp_host_table)) return 1;
  return 2;
}
float rno_rcp(float x) {
  uint32_t b, r;
  float f;
  memcpy(&b, &x, 4);
  r = rn_rcp_bits_from(rcp_table, b);
  memcpy(&f, &r, 4);
  return f;
}

/* src/vec_avx.h:398-416 (tanh8_approx): FMA Horner, rcp, clamp */
float rno_tanh(float x) {
  const float N0 = 952.52801514f, N1 = 96.39235687f, N2 = 0.60863042f;
  const float D0 = 952.72399902f, D1 = 413.36801147f, D2 = 11.88600922f;
  float x2 = x * x;
  float num = fmaf(fmaf(N2, x2, N1), x2, N0);
  float den = fmaf(fmaf(D2, x2, D1), x2, D0);
  num = num * x;
  den = rno_rcp(den);
  num = num * den;
  num = (1.f < num) ? 1.f : num;  /* _mm256_min_ps(max_out, num) */
  return (-1.f > num) ? -1.f : num; /* _mm256_max_ps(min_out, .) */
}

/* src/vec_avx.h:426-445 (sigmoid8_approx) */
float rno_sigmoid(float x) {
  const float N0 = 238.13200378f, N1 = 6.02452230f, N2 = 0.00950985f;
  const float D0 = 952.72399902f, D1 = 103.34200287f, D2 = 0.74287558f;
  float x2 = x * x;
  float num = fmaf(fmaf(N2, x2, N1), x2, N0);
  float den = fmaf(fmaf(D2, x2, D1), x2, D0);
  num = num * x;
  den = rno_rcp(den);
  num = fmaf(num, den, .5f);
  num = (1.f < num) ? 1.f : num;
  return (0.f > num) ? 0.f : num;
}

/* src/vec_avx.h:326-341 (AVX2 vector_ps_to_epi8): fused x*127+127, round to nearest even,
   then packus_epi32 (signed 32 -> unsigned 16 saturate) and packus_epi16 (the 16-bit value
   re-read as SIGNED -> unsigned 8 saturate) */
void rno_quantize_u8(unsigned char *q, const float *x, int n) {
  int i;
  for (i = 0; i < n; i++) {
    float xf = fmaf(x[i], 127.f, 127.f);
    int32_t xi;
    int32_t u16;
    int16_t s16;
    if (!(xf >= -2147483648.f && xf < 2147483648.f)) xi = INT32_MIN; /* cvtps2dq "indefinite" */
    else xi = (int32_t)lrintf(xf);
    u16 = xi < 0 ? 0 : (xi > 65535 ? 65535 : xi);
    s16 = (int16_t)(uint16_t)u16;
    q[i] = (unsigned char)(s16 < 0 ? 0 : (s16 > 255 ? 255 : s16));
  }
}

/* float matvec with AVX2 numerics: one FMA chain per output over the input index
   (src/vec_avx.h:672-730, weights column-major W[j*N+i]); bias added after
   (src/nnet_arch.h:149-151).  `fused`=0 is the scalar tail used when N%4 != 0
   (vec_avx.h:732-736: plain `out += w*x`, unfused in the pinned build) */
static void lin_float(const Lin *l, float *out, const float *x, int fused) {
  int i, j;
  for (i = 0; i < l->nout; i++) {
    float acc = 0;
    if (fused) for (j = 0; j < l->nin; j++) acc = fmaf(l->fw[j * l->nout + i], x[j], acc);
    else for (j = 0; j < l->nin; j++) acc = acc + l->fw[j * l->nout + i] * x[j];
    out[i] = acc + l->bias[i];
  }
}

/* int8 matvec, unsigned activations (src/vec_avx.h:778-877): exact integer accumulation of
   s8 weight x u8 activation, then float(acc)*scale, + subias (src/nnet_arch.h:145-151).
   Dense layout: w[((i/8)*(M/4) + j/4)*32 + (i%8)*4 + (j%4)]; sparse: per 8-row group
   idx = [nblocks, col0, col1, ...], blocks of 8x4 stored consecutively. */
static void lin_int8(const Lin *l, float *out, const float *x) {
  unsigned char q[1024];
  const int8_t *w = l->w;
  const int *idx = l->idx;
  int i, j, r, c;
  rno_quantize_u8(q, x, l->nin);
  for (i = 0; i < l->nout; i += 8) {
    int32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int nblk = idx ? *idx++ : l->nin / 4;
    for (j = 0; j < nblk; j++) {
      int col = idx ? *idx++ : 4 * j;
      for (r = 0; r < 8; r++)
        for (c = 0; c < 4; c++) acc[r] += (int32_t)w[r * 4 + c] * (int32_t)q[col + c];
      w += 32;
    }
    for (r = 0; r < 8; r++) out[i + r] = (float)acc[r] * l->scale[i + r] + l->subias[i + r];
  }
  if (l->diag) { /* recurrent diagonal, float, unfused (src/nnet_arch.h:153-161) */
    int M = l->nin;
    for (i = 0; i < M; i++) {
      out[i] += l->diag[i] * x[i];
      out[i + M] += l->diag[i + M] * x[i];
      out[i + 2 * M] += l->diag[i + 2 * M] * x[i];
    }
  }
}

/* src/nnet.c:65-94 */
static void gru_step(const Lin *in_w, const Lin *rec_w, float *state, const float *in) {
  enum { N = RN_GRU };
  float zrh[3 * N], recur[3 * N];
  float *z = zrh, *r = zrh + N, *h = zrh + 2 * N;
  int i;
  lin_int8(in_w, zrh, in);
  lin_int8(rec_w, recur, state);
  for (i = 0; i < 2 * N; i++) zrh[i] += recur[i];
  for (i = 0; i < 2 * N; i++) zrh[i] = rno_sigmoid(zrh[i]);
  for (i = 0; i < N; i++) h[i] += recur[2 * N + i] * r[i];
  for (i = 0; i < N; i++) h[i] = rno_tanh(h[i]);
  for (i = 0; i < N; i++) h[i] = z[i] * state[i] + (1 - z[i]) * h[i];
  for (i = 0; i < N; i++) state[i] = h[i];
}

/* src/rnn.c:44-60 + src/nnet.c:57-61,113-123 */
void rno_compute_rnn(const RnoModel *m, float *st, float *gains, float *vad, const float *features) {
  float tmp1[RN_CONV1_K], c1[RN_CONV1_OUT], tmp2[RN_CONV2_K], cat[RN_CAT];
  int i;
  /* conv1: [t-2 | t-1 | t] -> 128, tanh */
  memcpy(tmp1, st + RN_OFF_CONV1, 130 * sizeof(float));
  memcpy(tmp1 + 130, features, RN_CONV1_IN * sizeof(float));
  lin_float(&m->conv1, c1, tmp1, 1);
  for (i = 0; i < RN_CONV1_OUT; i++) c1[i] = rno_tanh(c1[i]);
  memcpy(st + RN_OFF_CONV1, tmp1 + RN_CONV1_IN, 130 * sizeof(float));
  /* conv2 (int8) -> 384, tanh */
  memcpy(tmp2, st + RN_OFF_CONV2, 256 * sizeof(float));
  memcpy(tmp2 + 256, c1, RN_CONV2_IN * sizeof(float));
  lin_int8(&m->conv2, cat, tmp2);
  for (i = 0; i < RN_CONV2_OUT; i++) cat[i] = rno_tanh(cat[i]);
  memcpy(st + RN_OFF_CONV2, tmp2 + RN_CONV2_IN, 256 * sizeof(float));
  gru_step(&m->gru_in[0], &m->gru_rec[0], st + RN_OFF_GRU1, cat);
  gru_step(&m->gru_in[1], &m->gru_rec[1], st + RN_OFF_GRU2, st + RN_OFF_GRU1);
  gru_step(&m->gru_in[2], &m->gru_rec[2], st + RN_OFF_GRU3, st + RN_OFF_GRU2);
  memcpy(cat + RN_GRU, st + RN_OFF_GRU1, 3 * RN_GRU * sizeof(float)); /* gru1|gru2|gru3 are contiguous */
  lin_float(&m->dense_out, gains, cat, 1);
  for (i = 0; i < NB; i++) gains[i] = rno_sigmoid(gains[i]);
  lin_float(&m->vad_dense, vad, cat, 0);
  *vad = rno_sigmoid(*vad);
}

/* ------------------------------------------------------------------------------------------
 * Frame driver: src/denoise.c:457-504 and its callees in the same file.
 * ---------------------------------------------------------------------------------------- */

/* src/denoise.c:409-419, coefficients :469-470 */
static void biquad_hp(float *y, float *mem, const float *x) {
  const float a0 = -1.99599f, a1 = 0.99600f, b0 = -2.f, b1 = 1.f;
  int i;
  for (i = 0; i < NFRAME; i++) {
    float xi = x[i];
    float yi = x[i] + mem[0];
    mem[0] = (float)(mem[1] + (b0 * (double)xi - a0 * (double)yi));
    mem[1] = (float)(b1 * (double)xi - a1 * (double)yi);
    y[i] = yi;
  }
}

/* src/denoise.c:421-455 */
static void pitch_filter(cpx *X, const cpx *P, const float *Ex, const float *Ep, const float *Exp, const float *g) {
  float r[NB], rf[NFREQ], newE[NB], norm[NB], normf[NFREQ];
  int i;
  for (i = 0; i < NB; i++) {
    float t;
    if (Exp[i] > g[i]) r[i] = 1;
    else r[i] = (float)((Exp[i] * Exp[i]) * (1 - (g[i] * g[i])) / (.001 + (g[i] * g[i]) * (1 - (Exp[i] * Exp[i]))));
    t = (0 > r[i]) ? 0 : r[i];
    t = (1 < t) ? 1 : t;
    r[i] = (float)sqrt(t);
    r[i] = (float)(r[i] * sqrt(Ex[i] / (1e-8 + Ep[i])));
  }
  interp_band_gain(rf, r);
  for (i = 0; i < NFREQ; i++) {
    X[i].r += rf[i] * P[i].r;
    X[i].i += rf[i] * P[i].i;
  }
  compute_band_energy(newE, X);
  for (i = 0; i < NB; i++) norm[i] = (float)sqrt(Ex[i] / (1e-8 + newE[i]));
  interp_band_gain(normf, norm);
  for (i = 0; i < NFREQ; i++) {
    X[i].r *= normf[i];
    X[i].i *= normf[i];
  }
}

/* the one libm call of the feature path: (float)log10(1e-2 + (double)Ex), src/denoise.c:383, with the HOST's libm as the
 * reference has it -- exposed so that a test can sweep the device's restatement of that libm against it */
void rno_log_energy(float *out, const float *Ex, int n) {
  for (int i = 0; i < n; i++) out[i] = (float)log10(1e-2 + Ex[i]);
}
/* the same for the n floats whose bit patterns are first_bits, first_bits + 1, ..., compared with got[]: returns how many
 * differ bit for bit (first_bad: the first such pattern).  Exhaustive sweeps: no input array, no output array. */
unsigned rno_log_energy_range_diff(unsigned first_bits, unsigned n, const float *got, unsigned *first_bad) {
  unsigned bad = 0;
  for (unsigned i = 0; i < n; i++) {
    const unsigned u = first_bits + i;
    float ex, want;
    memcpy(&ex, &u, 4);
    want = (float)log10(1e-2 + ex);
    if (memcmp(&want, &got[i], 4)) {
      if (!bad++ && first_bad) *first_bad = u;
    }
  }
  return bad;
}

void rno_state_init(float *st) { memset(st, 0, RN_STATE_FLOATS * sizeof(float)); }

/* rnn_frame_analysis (src/denoise.c:332-345) on an explicit 480-sample analysis memory.
   lowpass < NFREQ applies the TRAINING-mode band limit (denoise.c:340-343). */
static void frame_analysis(float *analysis_mem, cpx *X, float *Ex, const float *in, int lowpass) {
  float xw[NWIN];
  int i;
  memcpy(xw, analysis_mem, NFRAME * sizeof(float));
  memcpy(xw + NFRAME, in, NFRAME * sizeof(float));
  memcpy(analysis_mem, in, NFRAME * sizeof(float));
  apply_window(xw);
  forward_transform(X, xw);
  for (i = lowpass; i < NFREQ; i++) X[i].r = X[i].i = 0;
  compute_band_energy(Ex, X);
}

/* Experiment hook (tools/log10_flip_effect.py; never set by a test of the product): the `rno_flip_countdown`-th next call of
   frame_features moves Ly[rno_flip_band] to the adjacent float before the follower uses it -- what a log10 that rounds a
   double-rounding tie the other way would do; band -2: every band of every frame from then on -- so that its effect on the gains downstream can be measured. */
int rno_flip_countdown = 0, rno_flip_band = -1;
void rno_flip_log_energy(int countdown, int band) {
  rno_flip_countdown = countdown;
  rno_flip_band = band;
}

/* rnn_compute_frame_features (src/denoise.c:347-398) on the flat state; `training` selects the
   TRAINING=1 build semantics (no silence short-cut; return value E < 0.1, denoise.c:389,397). */
static int frame_features(float *st, cpx *X, cpx *P, float *Ex, float *Ep, float *Exp, float *features,
                          const float *x, int training, int lowpass, int *pitch_out, float *gain_out) {
  float p[NWIN], lp[RN_PITCH_BUF_SIZE >> 1], Ly[NB];
  float E = 0, gain, follow, logMax;
  float *pitch_buf = st + RN_OFF_PITCH_BUF;
  int i, pitch_index, last_period;
  frame_analysis(st + RN_OFF_ANALYSIS, X, Ex, x, lowpass);
  memmove(pitch_buf, pitch_buf + NFRAME, (RN_PITCH_BUF_SIZE - NFRAME) * sizeof(float));
  memcpy(pitch_buf + RN_PITCH_BUF_SIZE - NFRAME, x, NFRAME * sizeof(float));
  pitch_downsample(pitch_buf, lp);
  pitch_index = pitch_search(lp + (RN_PITCH_MAX_PERIOD >> 1), lp);
  pitch_index = RN_PITCH_MAX_PERIOD - pitch_index;
  memcpy(&last_period, st + RN_OFF_LAST_PERIOD, sizeof(int));
  gain = remove_doubling(lp, &pitch_index, last_period, st[RN_OFF_LAST_GAIN]);
  memcpy(st + RN_OFF_LAST_PERIOD, &pitch_index, sizeof(int));
  st[RN_OFF_LAST_GAIN] = gain;
  for (i = 0; i < NWIN; i++) p[i] = pitch_buf[RN_PITCH_BUF_SIZE - NWIN - pitch_index + i];
  apply_window(p);
  forward_transform(P, p);
  compute_band_energy(Ep, P);
  compute_band_corr(Exp, X, P);
  for (i = 0; i < NB; i++) Exp[i] = (float)(Exp[i] / sqrt(.001 + Ex[i] * Ep[i]));
  dct(&features[NB], Exp);
  features[2 * NB] = (float)(.01 * (pitch_index - 300));
  logMax = -2;
  follow = -2;
  if (rno_flip_countdown > 0) rno_flip_countdown--;
  for (i = 0; i < NB; i++) {
    double t;
    Ly[i] = (float)log10(1e-2 + Ex[i]);
    if (rno_flip_band == i && rno_flip_countdown == 0) {
      Ly[i] = nextafterf(Ly[i], 1e30f);
      rno_flip_band = -1;
    } else if (rno_flip_band == -2 && rno_flip_countdown == 0) {  /* every band of every frame from here on */
      Ly[i] = nextafterf(Ly[i], 1e30f);
    }
    t = (follow - 1.5 > Ly[i]) ? follow - 1.5 : Ly[i];
    Ly[i] = (float)((logMax - 7 > t) ? logMax - 7 : t);
    logMax = (logMax > Ly[i]) ? logMax : Ly[i];
    follow = (float)((follow - 1.5 > Ly[i]) ? follow - 1.5 : Ly[i]);
    E += Ex[i];
  }
  *pitch_out = pitch_index;
  *gain_out = gain;
  if (!training && E < 0.04) {
    memset(features, 0, RN_NB_FEATURES * sizeof(float));
    return 1;
  }
  dct(features, Ly);
  features[0] -= 12;
  features[1] -= 4;
  return training && E < 0.1;
}

/* One step of the training-feature extraction loop (src/dump_features.c:466-491, a TRAINING=1
   build): Ey from the clean frame, features from the noisy frame, band-gain targets, VAD passed
   through.  rec98 = features[65] | g[32] | vad.  `noise_free` = (noise_gain==0 && fgnoise_gain==0). */
void rno_train_frame(float *st_noisy, float *clean_analysis_mem, const float *clean, const float *noisy, int lowpass,
                     int band_lp, float vad_target, int noise_free, float *rec98) {
  cpx X[NFREQ], P[NFREQ], Y[NFREQ];
  float Ex[NB], Ep[NB], Exp[NB], Ey[NB];
  float pg;
  int i, pitch, silence;
  tables_init();
  frame_analysis(clean_analysis_mem, Y, Ey, clean, lowpass);
  silence = frame_features(st_noisy, X, P, Ex, Ep, Exp, rec98, noisy, 1, lowpass, &pitch, &pg);
  for (i = 0; i < NB; i++) { /* dump_features.c:472-478 */
    float g = (float)sqrt((Ey[i] + 1e-3) / (Ex[i] + 1e-3));
    if (g > 1) g = 1;
    if (silence || i > band_lp) g = -1;
    if (Ey[i] < 5e-2 && Ex[i] < 5e-2) g = -1;
    if (vad_target == 0 && noise_free) g = -1;
    rec98[RN_NB_FEATURES + i] = g;
  }
  rec98[RN_NB_FEATURES + NB] = vad_target;
}

float rno_process_frame(const RnoModel *m, float *st, float *out, const float *in, RnoRecord *rec) {
  cpx X[NFREQ], P[NFREQ];
  float x[NFRAME], xw[NWIN];
  float Ex[NB], Ep[NB], Exp[NB], features[RN_NB_FEATURES], g[NB], gf[NFREQ];
  float vad_prob = 0, gain;
  cpx *dX = (cpx *)(st + RN_OFF_DELAYED_X);
  int i, pitch_index, silence;
  tables_init();
  if (rec) memset(rec, 0, sizeof *rec);

  biquad_hp(x, st + RN_OFF_MEM_HP, in);
  silence = frame_features(st, X, P, Ex, Ep, Exp, features, x, 0, NFREQ, &pitch_index, &gain);
  if (rec) {
    rec->pitch = pitch_index;
    rec->pitch_gain = gain;
    rec->silence = silence;
  }

  if (!silence) {
    rno_compute_rnn(m, st, g, &vad_prob, features);
    if (rec) {
      memcpy(rec->features, features, sizeof features);
      memcpy(rec->gains, g, sizeof g);
      rec->vad = vad_prob;
    }
    pitch_filter(dX, (const cpx *)(st + RN_OFF_DELAYED_P), st + RN_OFF_DELAYED_EX, st + RN_OFF_DELAYED_EP,
                 st + RN_OFF_DELAYED_EXP, g);
    for (i = 0; i < NB; i++) { /* denoise.c:479-487 */
      float alpha = .6f;
      float *lastg = st + RN_OFF_LASTG;
      double q;
      g[i] = (g[i] > alpha * lastg[i]) ? g[i] : alpha * lastg[i];
      q = g[i] * (st[RN_OFF_DELAYED_EX + i] + 1e-3) / (Ex[i] + 1e-3);
      lastg[i] = (float)((1.f < q) ? 1.f : q);
    }
    interp_band_gain(gf, g);
    for (i = 0; i < NFREQ; i++) {
      dX[i].r *= gf[i];
      dX[i].i *= gf[i];
    }
  }

  /* frame_synthesis, denoise.c:400-407 */
  inverse_transform(xw, dX);
  apply_window(xw);
  for (i = 0; i < NFRAME; i++) out[i] = xw[i] + st[RN_OFF_SYNTHESIS + i];
  memcpy(st + RN_OFF_SYNTHESIS, xw + NFRAME, NFRAME * sizeof(float));

  memcpy(dX, X, sizeof X);
  memcpy(st + RN_OFF_DELAYED_P, P, sizeof P);
  memcpy(st + RN_OFF_DELAYED_EX, Ex, sizeof Ex);
  memcpy(st + RN_OFF_DELAYED_EP, Ep, sizeof Ep);
  memcpy(st + RN_OFF_DELAYED_EXP, Exp, sizeof Exp);
  return vad_prob;
}
