// dropin.cpp -- the reference's own API (include/rnnoise.h; reference implementation src/denoise.c:227-325,457-504) on
// device-resident state pools, and the combiner that turns concurrent one-frame calls into shared launches.
#include "shim.h"
#include "device_choice.h"

#include <linux/futex.h>
#include <sched.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

namespace {

// ---------------------------------------------------------------------------------------------
// device-resident one-stream states (rnnoise_create / rnnoise_destroy)
// ---------------------------------------------------------------------------------------------
int pool_rows() {
  static const int n = [] {
    const char *e = getenv("RNNOISE_AMD_POOL_ROWS");
    int v = e && *e ? atoi(e) : RN_POOL_ROWS_MAX;
    v = std::max(64, std::min(RN_POOL_ROWS_MAX, v));
    return (v + 63) & ~63;
  }();
  return n;
}

StatePool *pool_new(RNNModel *model, int device) {
  StatePool *p = new StatePool();
  p->rows = pool_rows();
  p->batch = rnnoise_batch_create(model, p->rows, device);
  if (!p->batch) {
    delete p;
    return nullptr;
  }
  DeviceGuard guard(device);
  if (!guard.ok ||
      hipHostMalloc((void **)&p->h_io, (size_t)p->rows * RN_ROW_IO * sizeof(float), hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess ||
      hipMalloc((void **)&p->d_flat, (size_t)StatePool::FLAT_ROWS * StatePool::FLAT_BLK * sizeof(float)) != hipSuccess) {
    if (p->h_io) hipHostFree(p->h_io);
    rnnoise_batch_destroy(p->batch);
    delete p;
    return nullptr;
  }
  memset(p->h_io, 0, (size_t)p->rows * RN_ROW_IO * sizeof(float));
  p->used.assign(p->rows / 64, 0ull);
  p->req = static_cast<int *>(calloc(p->rows, sizeof(int)));
  p->sleeping = static_cast<int *>(calloc(p->rows, sizeof(int)));
  if (!p->req || !p->sleeping) {
    free(p->req);
    free(p->sleeping);
    hipHostFree(p->h_io);
    hipFree(p->d_flat);
    rnnoise_batch_destroy(p->batch);
    delete p;
    return nullptr;
  }
  // launch groups run the latency network kernel (rn_nn_one_kernel, 125 KB of LDS by opt-in): where it cannot run, pooled frames go
  // through pool_step one state at a time, which has the vector kernel to fall back on
  p->comb.no_nn_one = nn_one_max_streams() < 1;
  return p;
}

// Which device a NEW pool goes to: device_choice.h (rn_pool_device), with $RNNOISE_AMD_DEVICE = <index> as the pin
int pool_device_for(size_t n_pools_so_far) {
  static const int pinned = [] {
    const char *e = getenv("RNNOISE_AMD_DEVICE");
    return e && *e ? atoi(e) : -1;
  }();
  return rn_pool_device(n_pools_so_far, pinned, rnnoise_amd_device_count());
}

// a free row of one of the model's pools (a new pool when all are full); zeroed like rnnoise_init().  max_row: rows below it only
// (self-contained states stage through d_flat, which has FLAT_ROWS blocks).
int pool_acquire(RNNModel *model, StatePool *&pool, int &slot, int max_row = RN_POOL_ROWS_MAX) {
  std::unique_lock<std::mutex> lk(model->mu);
  for (int pass = 0; pass < 2; pass++) {
    for (StatePool *p : model->pools) {
      std::lock_guard<std::mutex> pl(p->mu);
      // rnnoise_create() states take rows from the top, borrowed rows (max_row = FLAT_ROWS) from the bottom: the two kinds do not
      // crowd each other out of a pool
      const bool from_top = max_row >= p->rows;
      const int words = std::min((int)p->used.size(), (max_row + 63) / 64);
      for (int i = 0; i < words; i++) {
        const int w = from_top ? words - 1 - i : i;
        if (~p->used[w]) {
          const int bit = from_top ? 63 - __builtin_clzll(~p->used[w]) : __builtin_ctzll(~p->used[w]);
          if (w * 64 + bit >= max_row) break;
          p->used[w] |= 1ull << bit;
          slot = w * 64 + bit;
          pool = p;
          return 0;
        }
      }
    }
    if (pass == 0) {
      const int device = pool_device_for(model->pools.size());
      lk.unlock();  // rnnoise_batch_create takes the model lock itself
      StatePool *p = pool_new(model, device);
      lk.lock();
      if (!p) return -1;
      model->pools.push_back(p);
    }
  }
  return -1;
}

void pool_release(StatePool *p, int slot) {
  std::lock_guard<std::mutex> pl(p->mu);
  p->used[slot >> 6] &= ~(1ull << (slot & 63));
}

// the stateful arrays of one row back to all-zero (what rnnoise_init does to a DenoiseState, src/denoise.c:286)
int pool_zero_row(StatePool *p, int slot, hipStream_t st) {
  const RnGroupDev v = group_view(p->batch->g, slot, 1);
  const size_t N = p->batch->n;
  HIP_OK(hipMemsetAsync(v.mem_hp, 0, 2 * 4, st));
  HIP_OK(hipMemsetAsync(v.pitch_ring, 0, RN_RING_SIZE * 4, st));
  HIP_OK(hipMemsetAsync(v.xlp_ring, 0, RN_XRING_SIZE * 4, st));
  HIP_OK(hipMemsetAsync(v.synth_mem, 0, RN_FRAME_SIZE * 4, st));
  HIP_OK(hipMemsetAsync(v.last_gain, 0, 4, st));
  HIP_OK(hipMemsetAsync(v.last_period, 0, 4, st));
  HIP_OK(hipMemsetAsync(v.lastg, 0, RN_NB_BANDS * 4, st));
  HIP_OK(hipMemsetAsync(v.conv1_state, 0, 130 * 4, st));
  HIP_OK(hipMemsetAsync(v.conv2_state, 0, 256 * 4, st));
  for (int k = 0; k < 3; k++) HIP_OK(hipMemsetAsync(v.gru_state + k * N * RN_GRU, 0, RN_GRU * 4, st));
  for (int k = 0; k < RN_SPEC_SLOTS; k++) {
    HIP_OK(hipMemsetAsync(v.spec_X[k], 0, RN_SPEC_STRIDE * 4, st));
    HIP_OK(hipMemsetAsync(v.spec_P[k], 0, RN_SPEC_STRIDE * 4, st));
    HIP_OK(hipMemsetAsync(v.spec_E[k], 0, 96 * 4, st));
  }
  return 0;
}

// one frame of one row: the four kernels of a step on a one-stream view, on `st` (no side streams: a single frame has
// nothing to overlap with).  The row's frame bookkeeping (parity, ring slot, scratch copy) is the caller's.
int pool_step(StatePool *p, int slot, int parity, int ring_slot, long frame_no, float *d_out, const float *d_in, float *d_vad,
              hipStream_t st) {
  RNNoiseBatch *b = p->batch;
  RnGroupDev g = group_view(b->g, slot, 1);
  if (frame_no & 1) {
    g.features = g.features_b;
    g.silence = g.silence_b;
    g.pitch = g.pitch_b;
  }
  g.vad = d_vad;
  const int prev = (parity + RN_SPEC_SLOTS - 1) % RN_SPEC_SLOTS;
  HIP_OK(rn_launch_hp(&g, d_in, 0, ring_slot, st, nullptr, nullptr));
  HIP_OK(rn_launch_analysis(&g, &b->tb, ring_slot, parity, st, nullptr, nullptr));
  if (nn_one_max_streams() >= 1) HIP_OK(rn_launch_nn_one(&g, &b->m, &b->tb, st, nullptr, nullptr));
  else HIP_OK(rn_launch_nn_vector(&g, &b->m, &b->tb, st, nullptr, nullptr));
  HIP_OK(rn_launch_synthesis(&g, &b->tb, d_out, 0, parity, prev, st, nullptr, nullptr));
  return 0;
}

// ---------------------------------------------------------------------------------------------
// The combiner.  rnnoise_process_frame is one frame of one stream, synchronous (include/rnnoise.h:94); a frame is four
// dependent kernels of one workgroup each, ~110 us of GPU latency during which the GPU is all but idle, and HIP streams are
// multiplexed onto four hardware queues -- so T threads with a stream per state got the throughput of four (round 3: 24.9 k
// frames/s from sixteen threads).  Here the states of a pool share launches instead: a caller queues its request (row, frame
// phase; the frame itself is already in the row's pinned block) and
//   * if a stream is free, takes everything queued as ONE group: the four latency kernels over the row list (rn_dev.h:
//     RnRows, carried in the kernel arguments), waits for the stream, marks every member done and wakes the sleepers;
//   * otherwise waits on its request's state word -- spinning while there are cores to spin on, a futex after that;
//   * the caller that completes a group launches the next one for whoever queued up meanwhile and names one of ITS members
//     to wait for it, so the GPU never waits for a sleeping thread to be scheduled;
//   * the threads of a group that has just completed come back within microseconds of each other: the first one back holds
//     its launch for at most $RNNOISE_AMD_COMBINE_GATHER_US (15) until the others have queued too -- without that, T threads
//     in lock-step decay into T one-row launches and one big group that waits a whole frame time for a stream.
// One thread alone takes the first branch at once, every time: its latency is the four kernels', as before.
// ---------------------------------------------------------------------------------------------
enum : int { REQ_IDLE = 0, REQ_QUEUED, REQ_INFLIGHT, REQ_SYNCER, REQ_DONE_OK, REQ_DONE_FAIL };
inline int req_word(uint32_t seq, int state) { return (int)(seq << 4) | state; }

uint64_t now_ns() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}
inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#endif
}
int env_int(const char *name, int dflt) {
  const char *e = getenv(name);
  return e && *e ? atoi(e) : dflt;
}
// CPUs this process may actually run on: the affinity mask, cut down to the cgroup's CPU quota (a container with 16 CPUs of
// quota on a 128-thread host can keep 16 threads spinning, not 128)
int effective_cpus() {
  static const int n = [] {
    int cpus = (int)sysconf(_SC_NPROCESSORS_ONLN);
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) cpus = std::min(cpus, CPU_COUNT(&set));
    long quota = -1, period = -1;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
      char q[32] = "";
      if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max")) quota = atol(q);
      fclose(f);
    } else if (FILE *f1 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {  // cgroup v1
      if (fscanf(f1, "%ld", &quota) != 1) quota = -1;
      fclose(f1);
      if (FILE *f2 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
        if (fscanf(f2, "%ld", &period) != 1) period = -1;
        fclose(f2);
      }
    }
    if (quota > 0 && period > 0) cpus = (int)std::min<long>(cpus, std::max<long>(1, (quota + period - 1) / period));
    return std::max(1, cpus);
  }();
  return n;
}
long futex(int *addr, int op, int val) { return syscall(SYS_futex, addr, op, val, nullptr, nullptr, 0); }
inline volatile uint32_t *done_word(StatePool *p, int slot) { return reinterpret_cast<volatile uint32_t *>(p->h_io + (size_t)slot * RN_ROW_IO + RN_ROW_IO - 1); }

// Retire the entries of a group (combiner lock held): an entry's request goes INFLIGHT / SYNCER -> `state` unless its
// caller has already seen its frame come out and left (it may be back with a new request under a new sequence number:
// the compare-and-swap then fails and the new request is left alone).  Returns the callers asleep on their word: they are
// woken after the lock is dropped, by address only.
void comb_retire(StatePool *p, const std::vector<CombMember> &grp, int state, PooledRef *self, std::vector<int *> &wake) {
  for (const CombMember &e : grp) {
    if (e.ref == self) continue;
    for (int from : {REQ_INFLIGHT, REQ_SYNCER}) {
      int expect = req_word(e.seq, from);
      if (__atomic_compare_exchange_n(&p->req[e.slot], &expect, req_word(e.seq, state), false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {
        // (the row's words are the pool's: reading them is safe even if the state has just taken its frame and been destroyed)
        if (__atomic_load_n(&p->sleeping[e.slot], __ATOMIC_SEQ_CST)) wake.push_back(&p->req[e.slot]);
        break;
      }
    }
  }
}
void comb_wake(const std::vector<int *> &wake) {
  for (int *w : wake) futex(w, FUTEX_WAKE_PRIVATE, 1);
}

int comb_free_stream(StatePool *p) {  // (lock held) a stream without a group in flight, created on demand; -1: none
  Combiner &c = p->comb;
  static const int max_streams = std::max(1, std::min((int)Combiner::MAXG, env_int("RNNOISE_AMD_COMBINE_STREAMS", 3)));
  for (int k = 0; k < c.n_streams; k++)
    if (!c.busy[k]) return k;
  if (c.n_streams < max_streams) {
    if (hipStreamCreateWithFlags(&c.stream[c.n_streams], hipStreamNonBlocking) != hipSuccess) {
      (void)hipGetLastError();
      return -1;
    }
    return c.n_streams++;
  }
  return -1;
}

// (lock held) everything queued becomes the group in flight on stream k
void comb_take_queue(StatePool *p, int k, std::vector<CombMember> &grp) {
  Combiner &c = p->comb;
  if (c.queue.size() <= (size_t)RN_ROWS_MAX) {
    grp.swap(c.queue);
    c.queue.clear();
  } else {  // (a pool has more rows than a row list has entries: the oldest RN_ROWS_MAX requests go, the rest stay queued)
    grp.assign(c.queue.begin(), c.queue.begin() + RN_ROWS_MAX);
    c.queue.erase(c.queue.begin(), c.queue.begin() + RN_ROWS_MAX);
  }
  c.busy[k] = true;
  c.members[k] = grp;
  for (const CombMember &e : grp) {
    e.ref->grp = k;
    __atomic_store_n(&p->req[e.slot], req_word(e.seq, REQ_INFLIGHT), __ATOMIC_SEQ_CST);  // (a queued request's caller is waiting on it)
  }
}

// the four kernels of one frame step over the rows of `grp`, on stream k of the pool's combiner
int comb_launch(StatePool *p, int k, const std::vector<CombMember> &grp) {
  RNNoiseBatch *b = p->batch;
  RnRows rows;
  // (mapped, coherent pinned memory: the device's alias of the block -- the same address under unified addressing, asked for all the same)
  void *d_io = nullptr;
  HIP_OK(hipHostGetDevicePointer(&d_io, p->h_io, 0));
  rows.io = static_cast<float *>(d_io);
  rows.n = (int)grp.size();
  for (int i = 0; i < rows.n; i++) {
    const PooledRef *m = grp[i].ref;
    rows.e[i] = RN_ROW_ENTRY(m->slot, m->ring_slot, m->parity, grp[i].seq);
  }
  for (int i = rows.n; i < RN_ROWS_MAX; i++) rows.e[i] = 0;
  hipStream_t st = p->comb.stream[k];
  HIP_OK(rn_launch_hp_rows(&b->g, &rows, st));
#if RN_INSTRUMENT
  // $RNNOISE_AMD_TEST_FAIL_GROUP=<n>[,<m>...] (fault injection, instrumented library only): the n-th (m-th ...) launch group of the process
  // "fails" here -- AFTER its high-pass has been queued, which is the case that leaves a row's pitch ring one frame ahead of the
  // host-side slot counters
  static const std::vector<long> fail_at = [] {
    std::vector<long> v;
    if (const char *e = RN_LAB_ENV("TEST_FAIL_GROUP"))
      for (const char *q = e; *q;) {
        char *end = nullptr;
        v.push_back(strtol(q, &end, 10));
        if (end == q) break;
        q = *end ? end + 1 : end;
      }
    return v;
  }();
  static std::atomic<long> n_groups{0};
  const long this_group = n_groups.fetch_add(1);
  for (long f : fail_at)
    if (f == this_group) return -1;
#endif
  HIP_OK(rn_launch_analysis_rows(&b->g, &b->tb, &rows, st));
  HIP_OK(rn_launch_nn_rows(&b->g, &b->m, &b->tb, &rows, st));
  HIP_OK(rn_launch_synthesis_rows(&b->g, &b->tb, &rows, st));
  return 0;
}

bool comb_complete(StatePool *p, int k, PooledRef *self);
int comb_launch(StatePool *p, int k, const std::vector<CombMember> &grp);

// a launch failed part of the way: nothing of the group may outlive its frames; every member fails.
// A pool has more rows than a launch group has entries, so requests may still be QUEUED behind the failed group.  Somebody has to
// adopt them: a group in flight on another stream does when it completes, a caller inside its gather window does when the window
// closes -- and when there is neither, this thread does, here (before round 6 it did not, and those callers slept for ever).
void comb_abort(StatePool *p, int k, std::vector<CombMember> grp, PooledRef *self) {
  Combiner &c = p->comb;
  for (;;) {
    (void)hipStreamSynchronize(c.stream[k]);  // (whatever of the group was launched is off the rows before anybody zeroes them)
    (void)hipGetLastError();
    std::vector<int *> wake;
    std::vector<CombMember> next;
    {
      std::lock_guard<std::mutex> lk(c.mu);
      // the high-pass of the group may have run before the failing launch: the members' pitch rings may already hold this frame,
      // which their host-side ring slot will not count.  Every member restarts from zero at its next call (its caller is still
      // blocked on its request word: the entry's state is alive here).
      for (const CombMember &e : grp) __atomic_store_n(&e.ref->poisoned, 1, __ATOMIC_SEQ_CST);
      c.members[k].clear();
      bool others = c.gathering;
      for (int i = 0; i < c.n_streams; i++) others |= i != k && c.busy[i];
      if (!c.queue.empty() && !others) comb_take_queue(p, k, next);
      else c.busy[k] = false;
      comb_retire(p, grp, REQ_DONE_FAIL, self, wake);
    }
    comb_wake(wake);
    if (next.empty()) return;
    if (comb_launch(p, k, next)) {  // this one too: its members fail, and whatever is queued behind it gets the same treatment
      grp.swap(next);
      self = nullptr;
      continue;
    }
    // one of the adopted group's own members waits for it: the first one that is still waiting for its frame
    for (const CombMember &e : next) {
      int expect = req_word(e.seq, REQ_INFLIGHT);
      if (__atomic_compare_exchange_n(&p->req[e.slot], &expect, req_word(e.seq, REQ_SYNCER), false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {
        if (__atomic_load_n(&p->sleeping[e.slot], __ATOMIC_SEQ_CST)) futex(&p->req[e.slot], FUTEX_WAKE_PRIVATE, 1);
        return;
      }
    }
    (void)comb_complete(p, k, nullptr);  // (every member has already taken its frame and gone)
    return;
  }
}

// The caller `self` owns the wait for the group on stream k (it launched it, or was named its syncer): wait, hand the stream
// to whatever queued up meanwhile, retire the group.  Returns self's own result.
bool comb_complete(StatePool *p, int k, PooledRef *self) {
  Combiner &c = p->comb;
  // The group is complete when the last kernel has stored every member's sequence number into the member's `done` word
  // (pinned, coherent host memory): polling that is microseconds faster than waking up from hipStreamSynchronize, and each
  // member sees its own word without waiting for this thread.  $RNNOISE_AMD_COMBINE_POLL=0, or no word after 2 ms: the
  // stream is synchronised instead (which is also where a GPU fault would surface).
  static const bool poll = env_int("RNNOISE_AMD_COMBINE_POLL", 1) != 0;
  bool self_ok = true;
  const uint64_t t_begin = now_ns();
  for (bool first = true;; first = false) {
    std::vector<CombMember> done, next;
    {
      std::lock_guard<std::mutex> lk(c.mu);
      done = c.members[k];
    }
    bool flagged = poll;
    if (poll) {
      const uint64_t give_up = now_ns() + 2000000ull;
      for (const CombMember &e : done)
        // (a member that has left and come back has cleared its word for its next request: the old frame is out)
        for (unsigned it = 0; flagged && *done_word(p, e.slot) != e.seq && (__atomic_load_n(&p->req[e.slot], __ATOMIC_SEQ_CST) >> 4) == (int)e.seq; it++) {
          cpu_relax();
          if ((it & 255) == 255 && now_ns() > give_up) flagged = false;
        }
    }
    if (first && flagged) {  // how long a group takes from its launch to its frames: the followers' sleep is sized by it
      const uint64_t dt = now_ns() - t_begin, old = c.group_ns.load(std::memory_order_relaxed);
      c.group_ns.store(old ? (3 * old + dt) / 4 : dt, std::memory_order_relaxed);
    }
    bool ok = true;
    if (!flagged) {
      ok = hipStreamSynchronize(c.stream[k]) == hipSuccess;
      if (!ok) (void)hipGetLastError();
    }
    if (first) self_ok = ok;
    std::vector<int *> wake;
    {
      std::lock_guard<std::mutex> lk(c.mu);
      c.members[k].clear();
      if (!c.queue.empty() && !c.gathering) comb_take_queue(p, k, next);
      else c.busy[k] = false;
      comb_retire(p, done, ok ? REQ_DONE_OK : REQ_DONE_FAIL, first ? self : nullptr, wake);
    }
    comb_wake(wake);
    if (next.empty()) return self_ok;
    if (comb_launch(p, k, next)) {
      comb_abort(p, k, next, nullptr);
      return self_ok;
    }
    // one of the new group's own members waits for it: the first one that is still waiting for its frame
    for (const CombMember &e : next) {
      int expect = req_word(e.seq, REQ_INFLIGHT);
      if (__atomic_compare_exchange_n(&p->req[e.slot], &expect, req_word(e.seq, REQ_SYNCER), false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {
        if (__atomic_load_n(&p->sleeping[e.slot], __ATOMIC_SEQ_CST)) futex(&p->req[e.slot], FUTEX_WAKE_PRIVATE, 1);
        return self_ok;
      }
    }
    // (every member has already taken its frame and gone: this thread retires that group as well)
  }
}

// one frame of a pooled state through the combiner: r->h_io holds the input; true when out / vad are in place
bool comb_submit(StatePool *p, PooledRef *r) {
  Combiner &c = p->comb;
  static const uint64_t gather_ns = (uint64_t)std::max(0, env_int("RNNOISE_AMD_COMBINE_GATHER_US", 15)) * 1000ull;
  // $RNNOISE_AMD_COMBINE_LINGER_US (10): with other groups in flight (the GPU is busy anyway), a caller that could launch waits this
  // long for more requests to join its group -- fewer, larger groups, so that a stream is free more often when a request arrives
  static const uint64_t linger_ns = (uint64_t)std::max(0, env_int("RNNOISE_AMD_COMBINE_LINGER_US", 10)) * 1000ull;
  static const int spin_us = env_int("RNNOISE_AMD_COMBINE_SPIN_US", 400);
  // A follower of a group has nothing to do until the group's frames come out, ~100-150 us later, and round 4 had it spin the whole
  // time: at 16 threads that was 16 cores burning 2-3 x the CPU time the reference spends COMPUTING a frame.  Now it sleeps first --
  // a timed futex wait on its request word for (the running estimate of a group's duration) minus $RNNOISE_AMD_COMBINE_WAKE_EARLY_US
  // (40; measured 70 / 40 / 0 = spin from the start, 16 threads: 64 / 40 / 147 us of CPU per frame at 106 k / 109 k / 102 k frames/s,
  // profiles/r5_configs0_cthreads.txt) -- and spins only for the rest.  A wake-up by the group's owner ends the sleep early.
  static const int wake_early_us = env_int("RNNOISE_AMD_COMBINE_WAKE_EARLY_US", 40);
  struct Active {
    std::atomic<int> &a;
    explicit Active(std::atomic<int> &a_) : a(a_) { a.fetch_add(1, std::memory_order_relaxed); }
    ~Active() { a.fetch_sub(1, std::memory_order_relaxed); }
  } active(c.active);
  r->seq = (uint32_t)(r->req_no++ & 0x7fff) + 1u;  // (from the request counter, not the frame number: a retried frame is a new request)
  int *const req = &p->req[r->slot], *const sleeping = &p->sleeping[r->slot];
  *done_word(p, r->slot) = 0;
  const uint32_t seq = r->seq;
  // every way out with a frame: this caller is now "on its way back" (see the gather window below)
  auto leave = [&](bool ok) {
    if (ok) {
      c.t_complete_ns.store(now_ns(), std::memory_order_relaxed);
      c.pending_returns.fetch_add(1, std::memory_order_relaxed);
    }
    return ok;
  };
  int lead = -1;
  std::vector<CombMember> grp;
  // (lock held, stream k free, this request still QUEUED) the queue -- its oldest RN_ROWS_MAX requests -- becomes the group on stream k.
  // Returns k when this request is in it (the caller then leads: lead_group), -1 when more than a row list's worth was queued in front
  // of it: that group is launched here and handed to one of its members, and this thread goes on waiting like a follower.
  auto take_and_maybe_lead = [&](std::unique_lock<std::mutex> &lk, int k) -> int {
    comb_take_queue(p, k, grp);
    bool mine = false;
    for (const CombMember &e : grp) mine |= e.ref == r;
    if (mine) return k;
    lk.unlock();
    if (comb_launch(p, k, grp)) comb_abort(p, k, grp, nullptr);
    else {
      bool named = false;
      for (const CombMember &e : grp) {
        int expect = req_word(e.seq, REQ_INFLIGHT);
        if (__atomic_compare_exchange_n(&p->req[e.slot], &expect, req_word(e.seq, REQ_SYNCER), false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {
          if (__atomic_load_n(&p->sleeping[e.slot], __ATOMIC_SEQ_CST)) futex(&p->req[e.slot], FUTEX_WAKE_PRIVATE, 1);
          named = true;
          break;
        }
      }
      if (!named) (void)comb_complete(p, k, nullptr);
    }
    grp.clear();
    lk.lock();
    return -1;
  };
  auto lead_group = [&](int k) -> bool {
    if (comb_launch(p, k, grp)) {
      comb_abort(p, k, grp, r);
      return false;
    }
    return leave(comb_complete(p, k, r));
  };
  {
    std::unique_lock<std::mutex> lk(c.mu);
    if (now_ns() - c.t_complete_ns.load(std::memory_order_relaxed) > gather_ns + 5000) c.pending_returns.store(0, std::memory_order_relaxed);  // stale
    else if (c.pending_returns.load(std::memory_order_relaxed) > 0) c.pending_returns.fetch_sub(1, std::memory_order_relaxed);
    __atomic_store_n(req, req_word(seq, REQ_QUEUED), __ATOMIC_SEQ_CST);
    c.queue.push_back(CombMember{r, r->slot, seq});
    int k = c.gathering ? -1 : comb_free_stream(p);
    if (k < 0 && c.n_streams == 0) {
      // not a single stream could be created: no group is in flight that could adopt this request, nobody would ever wake it
      c.queue.pop_back();
      __atomic_store_n(req, req_word(seq, REQ_IDLE), __ATOMIC_SEQ_CST);
      return false;
    }
    if (k >= 0 && c.pending_returns.load(std::memory_order_relaxed) > 0 && gather_ns) {
      // the other callers that have just got their frames are on their way back: hold the stream for them
      c.gathering = true;
      lk.unlock();
      while (c.pending_returns.load(std::memory_order_relaxed) > 0 && now_ns() - c.t_complete_ns.load(std::memory_order_relaxed) < gather_ns)
        cpu_relax();
      lk.lock();
      c.gathering = false;
      k = comb_free_stream(p);
    }
    if (k >= 0 && linger_ns && !c.gathering) {
      bool busy = false;
      for (int i = 0; i < c.n_streams; i++) busy |= c.busy[i];
      if (busy) {
        const uint64_t t0 = now_ns();
        c.gathering = true;
        lk.unlock();
        while (now_ns() - t0 < linger_ns) cpu_relax();
        lk.lock();
        c.gathering = false;
        k = comb_free_stream(p);
      }
    }
    if (k >= 0 && __atomic_load_n(req, __ATOMIC_SEQ_CST) == req_word(seq, REQ_QUEUED)) lead = take_and_maybe_lead(lk, k);
  }
  if (lead >= 0) return lead_group(lead);
  // wait: for the frame (the row's `done` word, or the group's owner saying so), or to be named the owner of the group's wait
  const bool may_spin = c.active.load(std::memory_order_relaxed) <= effective_cpus();
  if (wake_early_us > 0) {
    const uint64_t est = c.group_ns.load(std::memory_order_relaxed);
    if (est > (uint64_t)(wake_early_us + 20) * 1000ull) {
      const uint64_t ns = est - (uint64_t)wake_early_us * 1000ull;
      const int w = __atomic_load_n(req, __ATOMIC_SEQ_CST);
      if ((w & 15) == REQ_QUEUED || (w & 15) == REQ_INFLIGHT) {
        const timespec ts{(time_t)(ns / 1000000000ull), (long)(ns % 1000000000ull)};
        __atomic_store_n(sleeping, 1, __ATOMIC_SEQ_CST);
        if (__atomic_load_n(req, __ATOMIC_SEQ_CST) == w) syscall(SYS_futex, req, FUTEX_WAIT_PRIVATE, w, &ts, nullptr, 0);
        __atomic_store_n(sleeping, 0, __ATOMIC_SEQ_CST);
      }
    }
  }
  const uint64_t spin_until = now_ns() + (uint64_t)(may_spin ? spin_us : 20) * 1000ull;
  for (unsigned it = 0;; it++) {
    const int w = __atomic_load_n(req, __ATOMIC_SEQ_CST), s = w & 15;
    if (s == REQ_SYNCER) return leave(comb_complete(p, r->grp, r));
    if (s == REQ_DONE_OK) return leave(true);
    if (s == REQ_DONE_FAIL) return false;
    if (s == REQ_INFLIGHT && *done_word(p, r->slot) == seq) {
      // the frame is out: say so (nobody may name this request its group's syncer any more) and go
      int expect = w;
      if (__atomic_compare_exchange_n(req, &expect, req_word(seq, REQ_DONE_OK), false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) return leave(true);
      continue;  // (named syncer, or retired, in between: look again)
    }
    if ((it & 63) != 63 || now_ns() < spin_until) {
      cpu_relax();
      continue;
    }
    // (timed: a request still QUEUED when the wait runs out, with no group in flight and nobody gathering, has nobody left to adopt it
    //  -- comb_abort and comb_complete see to it that this does not happen; if it ever does, its caller leads it instead of sleeping on)
    const timespec ts{0, 20 * 1000 * 1000};
    __atomic_store_n(sleeping, 1, __ATOMIC_SEQ_CST);
    if (__atomic_load_n(req, __ATOMIC_SEQ_CST) == w) syscall(SYS_futex, req, FUTEX_WAIT_PRIVATE, w, &ts, nullptr, 0);
    __atomic_store_n(sleeping, 0, __ATOMIC_SEQ_CST);
    if (s == REQ_QUEUED && __atomic_load_n(req, __ATOMIC_SEQ_CST) == w) {
      std::unique_lock<std::mutex> lk(c.mu);
      bool adopters = c.gathering;
      for (int i = 0; i < c.n_streams; i++) adopters |= c.busy[i];
      if (!adopters && __atomic_load_n(req, __ATOMIC_SEQ_CST) == w) {
        const int k = comb_free_stream(p);
        if (k >= 0) lead = take_and_maybe_lead(lk, k);
      }
      lk.unlock();
      if (lead >= 0) return lead_group(lead);
    }
  }
}

// a state that is being destroyed must not be listed in a group any more (its caller may have seen the frame come out before
// the group's owner retired the entry)
void comb_forget(StatePool *p, PooledRef *r) {
  Combiner &c = p->comb;
  for (;;) {
    {
      std::lock_guard<std::mutex> lk(c.mu);
      bool listed = false;
      for (int k = 0; k < c.n_streams; k++)
        for (const CombMember &e : c.members[k]) listed |= e.ref == r;
      if (!listed) return;
    }
    sched_yield();
  }
}

}  // namespace

void pools_free(RNNModel *model) {
  for (StatePool *p : model->pools) {
    {
      DeviceGuard guard(p->batch->device);
      for (int k = 0; k < p->comb.n_streams; k++) {
        hipStreamSynchronize(p->comb.stream[k]);
        hipStreamDestroy(p->comb.stream[k]);
      }
      hipHostFree(p->h_io);
      hipFree(p->d_flat);
    }
    free(p->req);
    free(p->sleeping);
    rnnoise_batch_destroy(p->batch);
    delete p;
  }
  model->pools.clear();
}

extern "C" int rnnoise_get_size(void) { return (int)sizeof(DenoiseState); }
extern "C" int rnnoise_get_frame_size(void) { return RN_FRAME_SIZE; }

// rnnoise_init() on caller-owned memory (include/rnnoise.h:57,71): there is no rnnoise_uninit, so such a state must not
// hold library resources -- it stays a self-contained POD and is staged to a pool row for every frame.
extern "C" int rnnoise_init(DenoiseState *st, RNNModel *model) {
  if (!st) return -1;
  memset(st, 0, sizeof *st);
  if (!model && !(model = default_model())) return -1;
  {
    std::lock_guard<std::mutex> lk(model->mu);
    if (model_parse_locked(model)) return -1;
  }
  if (rnnoise_amd_device_count() < 1) {
    fprintf(stderr, "[rnnoise_amd] no HIP device visible; this library has no CPU path\n");
    return -1;
  }
  st->magic = kStateMagic;
  st->model = model;
  return 0;
}

// rnnoise_create(): the state lives in HBM (a row of a StatePool) until rnnoise_destroy().
extern "C" DenoiseState *rnnoise_create(RNNModel *model) {
  if (!model && !(model = default_model())) return nullptr;
  {
    std::lock_guard<std::mutex> lk(model->mu);
    if (model_parse_locked(model)) return nullptr;
  }
  if (rnnoise_amd_device_count() < 1) {
    fprintf(stderr, "[rnnoise_amd] no HIP device visible; this library has no CPU path\n");
    return nullptr;
  }
  DenoiseState *st = static_cast<DenoiseState *>(calloc(1, sizeof(DenoiseState)));
  if (!st) return nullptr;
  PooledRef &r = st->ref;
  if (pool_acquire(model, r.pool, r.slot)) {
    free(st);
    return nullptr;
  }
  DeviceGuard guard(r.pool->batch->device);
  r.mu = new std::mutex();
  r.h_io = r.pool->h_io + (size_t)r.slot * RN_ROW_IO;
  // (the row is cleared on the legacy default stream, which the pool's non-blocking streams do not synchronise with: the other
  //  rows' frames are not held up)
  if (!guard.ok || pool_zero_row(r.pool, r.slot, nullptr) || hipStreamSynchronize(nullptr) != hipSuccess) {
    delete r.mu;
    pool_release(r.pool, r.slot);
    free(st);
    return nullptr;
  }
  st->magic = kPooledMagic;
  st->model = model;
  return st;
}

extern "C" void rnnoise_destroy(DenoiseState *st) {
  if (!st) return;
  if (st->magic == kPooledMagic) {
    PooledRef &r = st->ref;
    comb_forget(r.pool, &r);
    if (r.stream) {
      DeviceGuard guard(r.pool->batch->device);
      hipStreamSynchronize(r.stream);
      hipStreamDestroy(r.stream);
    }
    delete r.mu;
    pool_release(r.pool, r.slot);
  }
  free(st);
}

// One 480-sample frame of one stream (include/rnnoise.h:94).  There is no error channel in this signature: on a GPU
// failure the frame comes back zeroed with VAD 0 and the reason on stderr (the host process is never aborted).
static float frame_failed(float *out, const char *why) {
  fprintf(stderr, "[rnnoise_amd] rnnoise_process_frame: %s; returning a zeroed frame\n", why);
  if (out) memset(out, 0, RN_FRAME_SIZE * sizeof(float));
  return 0.f;
}

extern "C" float rnnoise_process_frame(DenoiseState *st, float *out, const float *in) {
  if (!st || (st->magic != kStateMagic && st->magic != kPooledMagic) || !st->model || !out || !in)
    return frame_failed(out, "uninitialised state or NULL buffer");
  if (st->magic == kPooledMagic) {
    // device-resident state.  The frame travels through the row's block of the pool's pinned memory, which the kernels address
    // directly (host memory mapped into the device's address space: the first kernel reads its 1,920 bytes over PCIe, the
    // last ones write frame and VAD back): no copy commands.  The launches are shared with whoever else is calling on this
    // pool right now (the combiner above); $RNNOISE_AMD_COMBINE=0: four launches on a stream of the state's own, as in
    // round 3 (A/B runs).
    PooledRef &r = st->ref;
    std::lock_guard<std::mutex> lk(*r.mu);
    DeviceGuard guard(r.pool->batch->device);
    if (!guard.ok) return frame_failed(out, "cannot select the HIP device");
    static const bool combine = [] { const char *e = getenv("RNNOISE_AMD_COMBINE"); return !e || atoi(e) != 0; }();
    float *h_in = r.h_io, *h_out = r.h_io + RN_FRAME_SIZE + 4;
    if (__atomic_load_n(&r.poisoned, __ATOMIC_SEQ_CST)) {
      // a launch group this state was part of failed after its high-pass may have run: the row's pitch ring and the host-side slot
      // counters no longer agree.  Restart the stream from zero (what rnnoise_init leaves) rather than run it one ring slot off.
      fprintf(stderr, "[rnnoise_amd] rnnoise_process_frame: the previous frame of this state failed on the GPU; its state restarts from zero\n");
      if (pool_zero_row(r.pool, r.slot, nullptr) || hipStreamSynchronize(nullptr) != hipSuccess) return frame_failed(out, "cannot reset the state's row");
      r.parity = r.ring_slot = 0;
      r.frame_no = 0;
      __atomic_store_n(&r.poisoned, 0, __ATOMIC_SEQ_CST);
    }
    memcpy(h_in, in, RN_FRAME_SIZE * sizeof(float));
    bool ok;
    if (combine && !r.pool->comb.no_nn_one) {
      ok = comb_submit(r.pool, &r);
    } else {
      ok = (r.stream || hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking) == hipSuccess) &&
           pool_step(r.pool, r.slot, r.parity, r.ring_slot, r.frame_no, h_out, h_in, h_out + RN_FRAME_SIZE, r.stream) == 0 &&
           hipStreamSynchronize(r.stream) == hipSuccess;
    }
    if (!ok) {
      __atomic_store_n(&r.poisoned, 1, __ATOMIC_SEQ_CST);  // (whichever path failed: the frame may be half in the row's state)
      return frame_failed(out, "GPU step failed");
    }
    r.parity = (r.parity + 1) % RN_SPEC_SLOTS;
    r.ring_slot = (r.ring_slot + 1) % RN_RING_SLOTS;
    r.frame_no++;
    memcpy(out, h_out, RN_FRAME_SIZE * sizeof(float));
    return h_out[RN_FRAME_SIZE];
  }
  // self-contained state: borrow a pool row for the duration of the call -- state + frame up in one copy, scatter,
  // the four kernels, gather, state + frame + VAD down in one copy.  Rows are per call, so states on different threads
  // proceed concurrently.
  RNNModel *m = st->model;
  StatePool *pool = nullptr;
  int slot = -1;
  if (pool_acquire(m, pool, slot, StatePool::FLAT_ROWS)) return frame_failed(out, "no GPU state row available (no CPU fallback)");
  struct Scratch {  // per-thread: a stream and a pinned staging block, created on first use, released at thread exit
    hipStream_t stream = nullptr;
    float *h = nullptr;
    int device = -1;
    bool ready(int dev) {  // both resources or neither: a half-built scratch is torn down and retried at the next call
      if (stream && h && device == dev) return true;
      release();
      if (hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) { stream = nullptr; return false; }
      if (hipHostMalloc((void **)&h, 2 * StatePool::FLAT_BLK * sizeof(float), hipHostMallocDefault) != hipSuccess) {
        h = nullptr;
        release();
        return false;
      }
      device = dev;
      return true;
    }
    void release() {
      if (stream || h) {
        DeviceGuard guard(device >= 0 ? device : 0);
        if (stream) (void)hipStreamDestroy(stream);
        if (h) (void)hipHostFree(h);
      }
      stream = nullptr;
      h = nullptr;
      device = -1;
    }
    ~Scratch() { release(); }
  };
  static thread_local Scratch sc;
  float vad = 0.f;
  bool ok = false;
  {
    DeviceGuard guard(pool->batch->device);
    constexpr size_t IO = StatePool::FLAT_IO, UP = IO + RN_FRAME_SIZE, DOWN = UP + 1;
    if (guard.ok && sc.ready(pool->batch->device)) {
      float *d_blk = pool->d_flat + (size_t)slot * StatePool::FLAT_BLK, *d_io = d_blk + IO;  // frame processed in place
      const RnGroupDev v = group_view(pool->batch->g, slot, 1);
      // conventions of a freshly scattered row: its newest frame sits in ring slot 5 and spectra slot 2, so the next frame
      // goes to ring slot 0 / spectra slot 0 and leaves its own "delayed" spectra in slot 0
      float *hu = sc.h, *hd = sc.h + StatePool::FLAT_BLK;
      memcpy(hu, st->state, RN_STATE_FLOATS * sizeof(float));
      memcpy(hu + IO, in, RN_FRAME_SIZE * sizeof(float));
      ok = hipMemcpyAsync(d_blk, hu, UP * sizeof(float), hipMemcpyHostToDevice, sc.stream) == hipSuccess &&
           rn_launch_state_scatter(&v, d_blk, RN_RING_SLOTS - 1, RN_SPEC_SLOTS - 1, sc.stream) == hipSuccess &&
           pool_step(pool, slot, 0, 0, 0, d_io, d_io, d_io + RN_FRAME_SIZE, sc.stream) == 0 &&
           rn_launch_state_gather(&v, d_blk, 0, 0, sc.stream) == hipSuccess &&
           hipMemcpyAsync(hd, d_blk, DOWN * sizeof(float), hipMemcpyDeviceToHost, sc.stream) == hipSuccess &&
           hipStreamSynchronize(sc.stream) == hipSuccess;
      if (ok) {
        memcpy(st->state, hd, RN_STATE_FLOATS * sizeof(float));
        memcpy(out, hd + IO, RN_FRAME_SIZE * sizeof(float));
        vad = hd[IO + RN_FRAME_SIZE];
      }
    }
  }
  pool_release(pool, slot);
  if (!ok) return frame_failed(out, "GPU step failed");
  return vad;
}

