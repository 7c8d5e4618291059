/* rnnoise_amd_debug.h -- test and measurement taps.  NOT part of the product library: these entry points exist only in the
 * instrumented build (rnnoise_amd/librnnoise_amd_instr.so: the same sources compiled with -DRN_INSTRUMENT=1, which also
 * compiles the stage taps and shader-clock probes into the kernels, plus the probe kernels of csrc/fft_probe.hip).
 * librnnoise_amd.so / librnnoise.so.0 export the symbols of rnnoise.h and rnnoise_amd.h and nothing else. */
#ifndef RNNOISE_AMD_DEBUG_H
#define RNNOISE_AMD_DEBUG_H

#include "rnnoise_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Pitch-analysis stage taps ([N][RN_DBG_FLOATS], layout rn_layout.h RN_DBG_*).  The first
 * call arms the taps (dst may be NULL); later calls copy the last step's record. Tests only. */
RNNOISE_EXPORT int rnnoise_batch_debug_pitch(RNNoiseBatch *b, float *dst);

/* Test tap: n independent 960-point transforms through the register-resident FFT of the analysis / synthesis kernels
 * (rnnoise_amd/csrc/fft_reg.h; reference: rnn_fft_c, src/kiss_fft.c:566-586), `reps` passes each with the spectrum fed
 * back as the next input.  in/out: [n][960][2] host floats, natural order.  variant 0/1 = exchange implementation.
 * clocks[n] (optional): shader clocks per wave; xlane[2][6][64] (optional): source lanes of the exchange primitives. */
RNNOISE_EXPORT int rnnoise_amd_debug_fft(int device, int variant, float *out, const float *in, int n, int reps,
                                         unsigned long long *clocks, int *xlane);

/* Test tap: out[i] = (float)log10(1e-2 + (double)ex[i]) evaluated on `device` by the feature stage's own function (src/denoise.c:383
 * is the one libm call of the path: the kernels restate the host libm's algorithm, rnnoise_amd/csrc/log10_glibc.h). Host buffers. 0 / -1.
 * _range: ex == NULL sweeps the n floats whose bit patterns are first_bits, first_bits + 1, ... (exhaustive sweeps);
 * model 0 = what this process's kernels use (rnnoise_amd_log10_model()), 1 = the device library's log10. */
RNNOISE_EXPORT int rnnoise_amd_debug_log_energy(int device, float *out, const float *ex, int n);
RNNOISE_EXPORT int rnnoise_amd_debug_log_energy_range(int device, float *out, const float *ex, unsigned first_bits, unsigned n, int model);

/* Race hunt (tools/gru_race.py): the log of the GRU layer kernel's checking instantiations ($RNNOISE_AMD_GRU_VARIANT=w4chk ...),
 * 484 words: [0] h_old vectors that differed from HBM, [1] of them stale (= the previous unit tile's), [2] vectors checked,
 * then 40 records of 12 words.  Reading clears it.  0 / -1. */
RNNOISE_EXPORT int rnnoise_amd_debug_gru_race(int device, unsigned *log, int words);

#ifdef __cplusplus
}
#endif
#endif
