/* rnnoise.h -- drop-in public API of the MI355X back end.
 *
 * Same ten entry points, types and calling conventions as xiph/rnnoise's
 * include/rnnoise.h (cited per function as rnnoise.h:<line> of the reference), so an
 * application written against the reference recompiles and relinks unchanged.  The
 * implementation behind them is HIP (rnnoise_amd/csrc); the throughput API for many
 * concurrent streams is the additive include/rnnoise_amd.h.
 *
 * Frames are 480 samples of 48 kHz mono, float, int16-scaled (the demo feeds samples in
 * the +-32768 range, examples/rnnoise_demo.c:56).
 *
 * Attribution: the ten prototypes below are the public interface of xiph/rnnoise
 * (include/rnnoise.h, Copyright (c) 2018 Gregor Richards, Copyright (c) 2017 Mozilla, BSD
 * 3-clause); they are reproduced because they ARE the contract.  Comments and everything
 * behind the interface are this project's.
 */
#ifndef RNNOISE_H
#define RNNOISE_H 1

#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef RNNOISE_EXPORT
#if defined(__GNUC__)
#define RNNOISE_EXPORT __attribute__((visibility("default")))
#else
#define RNNOISE_EXPORT
#endif
#endif

/* Opaque, as in the reference (rnnoise.h:51-52). */
typedef struct DenoiseState DenoiseState;
typedef struct RNNModel RNNModel;

/* rnnoise.h:57 -- bytes a caller must provide to rnnoise_init(). */
RNNOISE_EXPORT int rnnoise_get_size(void);

/* rnnoise.h:62 -- samples per frame (480). */
RNNOISE_EXPORT int rnnoise_get_frame_size(void);

/* rnnoise.h:71 -- initialise caller-allocated storage of rnnoise_get_size() bytes.
 * Returns 0, or -1 if the model is rejected.  A state initialised this way owns no
 * library resources and needs no destructor, exactly like the reference.
 * model==NULL selects the built-in model in the reference (rnnoise.h:64-76); the upstream
 * weights are a separate download and are not compiled in here either, so NULL loads the blob
 * named by $RNNOISE_AMD_DEFAULT_MODEL, else weights_blob.bin beside the library (once per
 * process); -1 if there is neither.
 * Each rnnoise_process_frame() on such a state stages it through the GPU with one copy each way. */
RNNOISE_EXPORT int rnnoise_init(DenoiseState *st, RNNModel *model);

/* rnnoise.h:80 -- allocate and initialise; free with rnnoise_destroy(). NULL on failure.
 * The state itself lives in GPU memory (a row of a per-model pool) for its whole life; the
 * returned handle carries its stream and lock.  model==NULL as for rnnoise_init(). */
RNNOISE_EXPORT DenoiseState *rnnoise_create(RNNModel *model);

/* rnnoise.h:87 -- free a state from rnnoise_create(); the model is freed separately, after. */
RNNOISE_EXPORT void rnnoise_destroy(DenoiseState *st);

/* rnnoise.h:94 -- denoise one frame; returns the voice-activity probability (0 on silent
 * frames).  `in` and `out` hold at least 480 floats and may be the same buffer.  Distinct
 * states may be driven from different threads concurrently; one state is not re-entrant.
 * On a GPU failure the frame is zeroed, 0 is returned and the reason goes to stderr. */
RNNOISE_EXPORT float rnnoise_process_frame(DenoiseState *st, float *out, const float *in);

/* rnnoise.h:102 -- model from a "DNNw" weight blob (or an "RNPK" pack, rnnoise_amd.h) in
 * memory; the buffer is borrowed and must stay valid until rnnoise_model_free(). */
RNNOISE_EXPORT RNNModel *rnnoise_model_from_buffer(const void *ptr, int len);

/* rnnoise.h:111 -- model from an open file (read eagerly and completely). */
RNNOISE_EXPORT RNNModel *rnnoise_model_from_file(FILE *f);

/* rnnoise.h:118 -- model from a path; NULL if the file cannot be opened. */
RNNOISE_EXPORT RNNModel *rnnoise_model_from_filename(const char *filename);

/* rnnoise.h:125 -- free a model after every state/batch created from it. */
RNNOISE_EXPORT void rnnoise_model_free(RNNModel *model);

#ifdef __cplusplus
}
#endif
#endif
