/* rn_layout.h -- constants of the RNNoise frame path and the portable per-stream state.
 *
 * Public, plain C.  Shared by the HIP library (rnnoise_amd/csrc), the additive batched
 * API (include/rnnoise_amd.h: state export/import) and the test oracle (oracle/).
 *
 * Sizes follow the reference: src/denoise.h:31-41 (frame geometry), the generated
 * rnnoise_data.h of the default architecture (cond 128, hidden 384;
 * torch/rnnoise/train_rnnoise.py:48-49) and struct DenoiseState src/denoise.c:68-88.
 */
#ifndef RN_LAYOUT_H
#define RN_LAYOUT_H

#define RN_FRAME_SIZE 480
#define RN_WINDOW_SIZE 960
#define RN_FREQ_SIZE 481
#define RN_NB_BANDS 32
#define RN_NB_FEATURES 65
#define RN_PITCH_MIN_PERIOD 60
#define RN_PITCH_MAX_PERIOD 768
#define RN_PITCH_FRAME_SIZE 960
#define RN_PITCH_BUF_SIZE 1728

#define RN_CONV1_IN 65
#define RN_CONV1_OUT 128
#define RN_CONV1_K (3 * RN_CONV1_IN)  /* 195 inputs to the conv1 matvec */
#define RN_CONV2_IN 128
#define RN_CONV2_OUT 384
#define RN_CONV2_K (3 * RN_CONV2_IN)  /* 384 */
#define RN_GRU 384
#define RN_GRU3 (3 * RN_GRU)          /* 1152 gate rows: z, r, h */
#define RN_CAT (4 * RN_GRU)           /* 1536 = conv2 | gru1 | gru2 | gru3 */

/* Portable per-stream state ("flat state"): the 25,128 live bytes of DenoiseState
 * (SURVEY 8a row S), as 6282 32-bit words.  All words are IEEE float except
 * RN_OFF_LAST_PERIOD, which holds an int32 bit pattern.  Complex spectra are
 * interleaved (re, im). */
#define RN_OFF_ANALYSIS 0                                   /* analysis_mem[480]  */
#define RN_OFF_SYNTHESIS (RN_OFF_ANALYSIS + 480)            /* synthesis_mem[480] */
#define RN_OFF_PITCH_BUF (RN_OFF_SYNTHESIS + 480)           /* pitch_buf[1728]    */
#define RN_OFF_LAST_GAIN (RN_OFF_PITCH_BUF + 1728)
#define RN_OFF_LAST_PERIOD (RN_OFF_LAST_GAIN + 1)           /* int32 */
#define RN_OFF_MEM_HP (RN_OFF_LAST_PERIOD + 1)              /* mem_hp_x[2]        */
#define RN_OFF_LASTG (RN_OFF_MEM_HP + 2)                    /* lastg[32]          */
#define RN_OFF_CONV1 (RN_OFF_LASTG + 32)                    /* conv1_state[130]   */
#define RN_OFF_CONV2 (RN_OFF_CONV1 + 130)                   /* conv2_state[256]   */
#define RN_OFF_GRU1 (RN_OFF_CONV2 + 256)
#define RN_OFF_GRU2 (RN_OFF_GRU1 + 384)
#define RN_OFF_GRU3 (RN_OFF_GRU2 + 384)
#define RN_OFF_DELAYED_X (RN_OFF_GRU3 + 384)                /* 481 complex        */
#define RN_OFF_DELAYED_P (RN_OFF_DELAYED_X + 962)
#define RN_OFF_DELAYED_EX (RN_OFF_DELAYED_P + 962)
#define RN_OFF_DELAYED_EP (RN_OFF_DELAYED_EX + 32)
#define RN_OFF_DELAYED_EXP (RN_OFF_DELAYED_EP + 32)
#define RN_STATE_FLOATS (RN_OFF_DELAYED_EXP + 32)           /* 6282 words = 25,128 B */

/* Stage-tap record of the pitch analysis (tests only; rnnoise_batch_debug_pitch) */
#define RN_DBG_XLP 0          /* [864] decimated + whitened signal (src/pitch.c:146-214)  */
#define RN_DBG_AC 864         /* [5]   lag-windowed autocorrelation                       */
#define RN_DBG_LPC 869        /* [5]   FIR taps lpc2[]                                    */
#define RN_DBG_XC_COARSE 880  /* [147] coarse xcorr (src/pitch.c:332)                     */
#define RN_DBG_BEST 1030      /* coarse best0,best1, fine best0,best1, offset, 768-pitch  */
#define RN_DBG_XC_FINE 1040   /* [294] fine xcorr (src/pitch.c:344-361)                   */
#define RN_DBG_DOTS 1340      /* xx, xy, yy, T, xcorr[3] of rnn_remove_doubling           */
#define RN_DBG_CLK 1348       /* [12] shader-clock deltas of the analysis kernel's sections (profiling) */
#define RN_DBG_CLK2 1360      /* [16] shader-clock deltas of the MFMA network kernel phases (tile leader streams) */
#define RN_DBG_FLOATS 1400

#endif /* RN_LAYOUT_H */
