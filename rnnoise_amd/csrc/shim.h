// shim.h -- what the host-side translation units of librnnoise_amd.so share: the kernel launchers (one per .hip file), the
// device guard and error macros, the host-side types behind the opaque handles of include/rnnoise.h and include/rnnoise_amd.h,
// and the few functions that cross file boundaries.  Internal: nothing here is exported (-fvisibility=hidden, exports.map).
//
//   model.cpp    "DNNw" blob reader, "RNPK" pack, GPU re-layout of the model, rnnoise_model_* entry points
//   tables.cpp   static tables by formula, the rcpps profile
//   batch.cpp    rnnoise_batch_*: N streams on one GPU, the frame pipeline over HIP streams, state export / import, timing
//   host_io.cpp  host-fed calls: the pinned frame ring, the bounce chunks of pageable callers
//   dropin.cpp   the reference's own API (include/rnnoise.h): state pools, the combiner of concurrent one-frame calls
#pragma once
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rnnoise_amd.h"
#include "rn_dev.h"
#if RN_INSTRUMENT
#include "../../include/rnnoise_amd_debug.h"
#endif
#include "rcp_profiles.h"

extern "C" hipError_t rn_launch_hp(const RnGroupDev *, const void *, int in_s16, int, hipStream_t, hipEvent_t, hipEvent_t);
extern "C" hipError_t rn_launch_analysis(const RnGroupDev *, const RnTablesDev *, int, int, hipStream_t, hipEvent_t, hipEvent_t);
extern "C" hipError_t rn_launch_synthesis(const RnGroupDev *, const RnTablesDev *, void *, int out_s16, int, int, hipStream_t,
                                          hipEvent_t, hipEvent_t);
extern "C" hipError_t rn_launch_train_features(const RnGroupDev *, const RnTablesDev *, const float *, int, int,
                                               const RnTrainArgs *, hipStream_t);
extern "C" hipError_t rn_launch_nn_vector(const RnGroupDev *, const RnModelDev *, const RnTablesDev *, hipStream_t, hipEvent_t,
                                          hipEvent_t);
extern "C" hipError_t rn_launch_nn_one(const RnGroupDev *, const RnModelDev *, const RnTablesDev *, hipStream_t, hipEvent_t, hipEvent_t);
extern "C" hipError_t rn_launch_nn_mfma(const RnGroupDev *, const RnModelDev *, const RnTablesDev *, hipStream_t, hipEvent_t,
                                        hipEvent_t, int alone);  // alone: no other kernel of the call runs beside it
extern "C" hipError_t rn_launch_nn_layers(const RnGroupDev *, const RnModelDev *, const RnTablesDev *, hipStream_t, hipEvent_t[5][2]);
extern "C" int rn_nn_layers_launches(void);
extern "C" hipError_t rn_launch_nn_requant(const RnGroupDev *, hipStream_t);
extern "C" int rn_nn_mfma_available(void);
extern "C" hipError_t rn_launch_release_store(void *, long long, hipStream_t);
#if RN_INSTRUMENT
extern "C" hipError_t rn_launch_analysis_synth(const RnGroupDev *, const RnGroupDev *, const RnTablesDev *, int, int, void *, int, int, int, hipStream_t,
                                               hipEvent_t, hipEvent_t);
extern "C" hipError_t rn_launch_log_energy(const float *, unsigned, float *, unsigned, const double *, hipStream_t);
extern "C" hipError_t rn_launch_fft_probe(int, const float *, float *, unsigned long long *, int, int, const RnTablesDev *, hipStream_t);
extern "C" hipError_t rn_launch_xlane_probe(int *, hipStream_t);
#endif
extern "C" hipError_t rn_launch_state_gather(const RnGroupDev *, float *, int, int, hipStream_t);
extern "C" hipError_t rn_launch_state_scatter(const RnGroupDev *, const float *, int, int, hipStream_t);
extern "C" hipError_t rn_launch_copy_to_host(void *, const void *, size_t, int, hipStream_t);


extern "C" hipError_t rn_launch_hp_rows(const RnGroupDev *, const RnRows *, hipStream_t);
extern "C" hipError_t rn_launch_analysis_rows(const RnGroupDev *, const RnTablesDev *, const RnRows *, hipStream_t);
extern "C" hipError_t rn_launch_nn_rows(const RnGroupDev *, const RnModelDev *, const RnTablesDev *, const RnRows *, hipStream_t);
extern "C" hipError_t rn_launch_synthesis_rows(const RnGroupDev *, const RnTablesDev *, const RnRows *, hipStream_t);

// Every entry point works on the batch's device and leaves the calling thread's current device as it found it
// (a host thread may be driving another GPU: torch on cuda:0 beside a batch on device 1).
struct DeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    ok = (prev == device) || hipSetDevice(device) == hipSuccess;
    if (prev == device) prev = -1;  // nothing to restore
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};
#define ON_DEVICE(dev)                                                                                 \
  DeviceGuard guard_(dev);                                                                             \
  if (!guard_.ok) {                                                                                    \
    fprintf(stderr, "[rnnoise_amd] cannot select HIP device %d (%s:%d)\n", (dev), __FILE__, __LINE__);  \
    return -1;                                                                                         \
  }

#define HIP_OK(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess) {                                                                            \
      fprintf(stderr, "[rnnoise_amd] %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return -1;                                                                                       \
    }                                                                                                  \
  } while (0)

// Host view of one layer (pointers alias the blob, like the reference's LinearLayer)
struct HostLinear {
  const float *bias = nullptr, *subias = nullptr, *fw = nullptr, *diag = nullptr, *scale = nullptr;
  const int8_t *w = nullptr;
  const int32_t *idx = nullptr;
  int idx_words = 0, nblocks = 0, nin = 0, nout = 0;
  bool is_int8() const { return w != nullptr; }
};

struct HostModel {
  HostLinear conv1, conv2, gru_in[3], gru_rec[3], dense_out, vad_dense;
};

// device arena: one allocation, 256-byte aligned carve-outs
struct Staging {
  std::vector<uint8_t> bytes;
  size_t add(const void *src, size_t n) {
    size_t off = (bytes.size() + 255) & ~size_t(255);
    bytes.resize(off + n);
    if (src) memcpy(bytes.data() + off, src, n);
    else memset(bytes.data() + off, 0, n);
    return off;
  }
};

struct DevLinearOffsets {
  size_t bias = 0, fw = 0, scale = 0, diag = 0, w = 0, wmf = 0, rowsum = 0, grp = 0, cols = 0;  // (float layers: wmf = MFMA-ordered copy)
  bool has_fw = false, has_diag = false, has_cols = false, is_int8 = false;
};

struct DeviceModel {
  int device = -1;
  void *mem = nullptr, *mem_rows = nullptr;  // the staged model; the row-major int8 copies of the vector path
  RnModelDev dev{};
};

// =============================================================================================
// public types
// =============================================================================================
struct StatePool;
struct StagedModel;  // model.cpp
struct RNNModel {
  const void *const_blob = nullptr;  // borrowed (rnnoise_model_from_buffer)
  void *blob = nullptr;              // owned (rnnoise_model_from_file)
  int blob_len = 0;
  FILE *file = nullptr;
  std::mutex mu;
  int parsed = 0;  // 0 not yet, 1 ok, -1 rejected
  HostModel host;                  // layer views into a "DNNw" blob (unused for a packed model)
  StagedModel *staged = nullptr;   // device layout of every layer, built from the blob or taken from an "RNPK" pack
  long weight_bytes = 0;           // SURVEY 8d "W"
  std::vector<DeviceModel> dev;
  std::vector<StatePool *> pools;  // device-resident one-stream states behind rnnoise_create / rnnoise_process_frame
  const void *bytes() const { return blob ? blob : const_blob; }
};

struct RNNoiseBatch {
  RNNModel *model = nullptr;
  int device = 0, n = 0, nn_path = 0;
  bool img_valid = false;  // g.act_q[1..3] mirror gru_state (rn_dev.h); cleared by whatever else writes the state
  int schedule = 0;  // 0: default (3-stream frame pipeline in multi-frame calls); 9: one stream; 1: only the high-pass aside
  int parity = 0;  // spectra slot (mod RN_SPEC_SLOTS) the next frame writes; the previous one holds the delayed spectra
  long frame_no = 0;  // selects the per-step scratch copy (features / silence / pitch are double-buffered)
  float *features2[2] = {nullptr, nullptr};
  int *silence2[2] = {nullptr, nullptr}, *pitch2[2] = {nullptr, nullptr};
  int ring_slot = 0;  // pitch-ring slot the next frame is written to
  // side stream + events: in multi-frame calls the (latency-bound, 1 lane per stream) high-pass of frame
  // f+1 runs beside analysis/network/synthesis of frame f
  hipStream_t side = nullptr, side_hp = nullptr;
  // ordering events of the pipelined schedule: own_* are the batch's persistent events, cur_* the handle that marks the
  // completion of hp / analysis / synthesis of frame f & 7 (an own_* event, or the stop event of a timed launch)
  hipEvent_t ev_begin = nullptr, own_hp[8] = {}, own_k1[8] = {}, own_k3[8] = {};
  hipEvent_t cur_hp[8] = {}, cur_k1[8] = {}, cur_k3[8] = {};
  void *arena = nullptr;
  size_t arena_bytes = 0;
  RnGroupDev g{};
  RnModelDev m{};
  RnTablesDev tb{};
  float *scratch_gains = nullptr, *scratch_vad = nullptr;
  float *debug_buf = nullptr;
  float *state_stage = nullptr;  // one flat state in HBM: export / import go through the gather / scatter kernels
  // host-fed path (rnnoise_batch_process): two chunks in flight -- H2D of chunk i+1 and D2H of chunk i-1 overlap the
  // kernels of chunk i; pinned bounce buffers are used only when the caller's memory is pageable
  struct HostIo {
    hipStream_t up = nullptr, run = nullptr, down = nullptr;
    hipEvent_t up_done[2] = {}, run_done[2] = {}, down_done[2] = {};
    float *d_in[2] = {}, *d_out[2] = {}, *d_vad[2] = {}, *d_gains[2] = {};
    float *h_in[2] = {}, *h_out[2] = {}, *h_vad[2] = {}, *h_gains[2] = {};
    int chunk_frames = 0;
    size_t pcm_floats = 0;  // capacity of d_in / d_out (and h_in / h_out) of one chunk
    // pinned callers: a ring of RING frame slots, filled and drained frame by frame beside the kernels
    static constexpr int RING = 6;
    char *ring_mem = nullptr;
    hipEvent_t r_k3[RING] = {}, r_down[RING] = {}, r_up[RING] = {}, r_hp[RING] = {};
    struct Sdma *sdma = nullptr;  // copy mode "sdma" (host_io.cpp): explicit copy engines underneath HIP, created on first use
  } io;
  // timing
  bool timing = false;
  struct Ev { hipEvent_t a, b; int kind; };
  std::vector<Ev> pending, pool;
  double ms_sum[4] = {0, 0, 0, 0};  // analysis, network, synthesis, high-pass
  long launches = 0;
};

struct PooledRef;
// The combiner of a pool (dropin.cpp): concurrent rnnoise_process_frame calls on states of one pool are gathered into launch
// groups -- ONE set of the four latency kernels over a row list (rn_dev.h: RnRows) per group, a few groups in flight on streams
// of their own.
struct CombMember {
  PooledRef *ref;  // dereferenced only while the request is queued or being launched (its caller is blocked then); afterwards an identity
  int slot;        // the state's row: its request / sleeping / done words are the pool's (StatePool::req ...), valid whatever the state does
  uint32_t seq;    // the request of `ref` this entry stands for (a state that has its frame may be back with the next one
                   // before the group's owner has retired the old entry)
};
struct Combiner {
  static constexpr int MAXG = 8;
  std::mutex mu;
  int n_streams = 0;                       // streams created so far (at most $RNNOISE_AMD_COMBINE_STREAMS)
  hipStream_t stream[MAXG] = {};
  bool busy[MAXG] = {};                    // a group is in flight on the stream
  std::vector<CombMember> members[MAXG];   // ... these requests
  std::vector<CombMember> queue;           // submitted, not launched yet
  bool gathering = false;                  // a caller holds a free stream back for the callers that have just got their frames
  std::atomic<int> pending_returns{0};     // callers that have just got their frame and have not come back with the next one yet
  std::atomic<uint64_t> t_complete_ns{0};  // ... when the last of them left
  std::atomic<int> active{0};              // threads inside rnnoise_process_frame on this pool right now (spin or sleep?)
  std::atomic<uint64_t> group_ns{0};       // running estimate of a group's launch -> frames-out time (followers sleep through most of it)
  bool no_nn_one = false;                  // the latency network kernel cannot run here (LDS opt-in refused, $RNNOISE_AMD_NN_ONE_MAX=0):
                                           // frames go one state at a time through pool_step instead of through launch groups
};

// A pool of device-resident one-stream states of one model on one device: the arrays of a POOL_SLOTS-stream batch, of
// which every rnnoise_create() owns one row, and one block of pinned host memory per row through which the row's frames
// travel (the kernels address it directly: no copy commands).
struct StatePool {
  // rows of a pool: $RNNOISE_AMD_POOL_ROWS (default and maximum RN_POOL_ROWS_MAX = 1024, at least 64; 27 KB of HBM and 3.9 KB of
  // pinned host memory per row).  Round 4 had 64 = one launch group's worth, so the 65th state opened a second pool with a combiner
  // and three streams of its own -- 64 threads over 256 / 1,024 states: 152 k / 30 k frames/s against 304 k / 132 k with 256 rows.
  int rows = 0;
  RNNoiseBatch *batch = nullptr;   // owns the arena; never processed as a whole
  static constexpr int FLAT_IO = RN_STATE_FLOATS + 2;           // frame offset inside a staging block (16-byte aligned)
  static constexpr int FLAT_BLK = FLAT_IO + RN_FRAME_SIZE + 4;  // state | pad | frame (in, then out in place) | vad | pad
  static constexpr int FLAT_ROWS = 64;                          // staging blocks (rnnoise_init states borrow rows 0 .. 63 only)
  float *h_io = nullptr;           // pinned [rows][RN_ROW_IO]: in[480] | pad[4] | out[480] | vad | pad[2] | done (rn_dev.h: RnRows)
  float *d_flat = nullptr;         // [FLAT_ROWS][FLAT_BLK] staging for self-contained states (rnnoise_init path)
  std::mutex mu;
  std::vector<unsigned long long> used;  // bit per row
  // combiner words of the rows' requests.  They live HERE, not in the state (PooledRef), because a group's owner touches a
  // member's words after the member may have taken its frame and been destroyed: pool memory outlives every state of the pool
  int *req = nullptr;        // [rows] (sequence number << 4 | state) of the row's request in flight (atomic access; futex word)
  int *sleeping = nullptr;   // [rows] the row's caller sleeps on req[row] (atomic access)
  Combiner comb;
};

// What a DenoiseState holds when it came from rnnoise_create(): a row of a StatePool plus the host-side frame
// bookkeeping of that row.  (A member of a union with the self-contained state: plain data only.)
struct PooledRef {
  StatePool *pool;
  int slot;
  int parity, ring_slot;
  long frame_no;
  float *h_io;        // the row's block of the pool's pinned frame memory
  std::mutex *mu;     // one frame at a time per state (the reference's states are not re-entrant either)
  hipStream_t stream; // only when the combiner is switched off ($RNNOISE_AMD_COMBINE=0): the state's own stream
  int grp;            // combiner: stream slot of the group the request went into
  uint32_t seq;       // combiner: sequence number of the request (never 0); the last kernel stores it into the row's `done` word
  uint32_t req_no;    // combiner: requests made so far, failed ones included (the sequence numbers come from here, so that a retried
                      //           frame is never mistaken for the failed request)
  int poisoned;       // a launch group this state was in failed part of the way: its pitch ring may already hold the frame.  The
                      //           next call restarts the state from zero (as rnnoise_init would) instead of running one slot off
};

struct DenoiseState {
  uint32_t magic;
  uint32_t pad;
  RNNModel *model;
  union {
    float state[RN_STATE_FLOATS];  // rnnoise_init() on caller memory: self-contained POD, no library-owned resource (SURVEY 8b "Types")
    PooledRef ref;                 // rnnoise_create(): device-resident, released by rnnoise_destroy()
  };
};
static const uint32_t kStateMagic = 0x524e4e41u;   // "RNNA": self-contained state
static const uint32_t kPooledMagic = 0x524e4e50u;  // "RNNP": row of a StatePool

// Hooks of the host-fed path: the frame buffers of a call are then a ring of `ring` frame slots in HBM (frame f lives in
// slot f % ring) that uploads fill and downloads drain while the kernels run; each hook is called on the host right where
// the named kernel of frame f is enqueued, with the stream it goes to.
struct FrameIoHooks {
  int ring = 0;
  std::function<int(int, hipStream_t)> before_hp, after_hp, before_nn, after_k3;
};

// ---- functions that cross file boundaries ----
int tables_for_device(int device, RnTablesDev &out);                 // tables.cpp
int model_parse_locked(RNNModel *m);                                 // model.cpp (caller holds m->mu)
int model_on_device(RNNModel *m, int device, RnModelDev &out);       // model.cpp
RNNModel *default_model();                                           // model.cpp: the blob behind model == NULL
void pools_free(RNNModel *model);                                    // dropin.cpp: the state pools of a model (rnnoise_model_free)
RnGroupDev group_view(const RnGroupDev &g, int first, int count);    // batch.cpp
int nn_one_max_streams();                                            // batch.cpp
int batch_process_device_impl(RNNoiseBatch *b, void *d_out, const void *d_in, float *d_vad, float *d_gains, int n_frames,
                              void *hip_stream, bool s16, const FrameIoHooks *hk = nullptr);  // batch.cpp
void host_io_release(RNNoiseBatch *b);                               // host_io.cpp
