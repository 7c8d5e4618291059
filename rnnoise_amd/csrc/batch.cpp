// batch.cpp -- rnnoise_batch_*: N streams resident on one GPU; the frame step as a pipeline over HIP streams.
#include "shim.h"
#include "device_choice.h"

namespace {
template <typename T>
T *carve(uint8_t *&p, size_t count) {
  T *r = reinterpret_cast<T *>(p);
  p += (count * sizeof(T) + 255) & ~size_t(255);
  return r;
}

}  // namespace

// Batches from this size up run the network layer by layer (nn_layers.hip: 64 streams per GRU workgroup); below it the five launches
// and the smaller grids cost more than the weight reuse gains.  The tile kernel holds out while a CU has at most two tiles (8,192
// streams on 256 CUs: 25.9 against 24.2 M frames/s); with a third its K2 jumps (0.139 -> 0.193 ms at 10,240 streams) and the layer-wise
// network is ahead -- 27.5 against 25.8 M frames/s at 10,240, 29.5 against 26.2 at 12,288, one frame per call 24.8 against 23.0 M at
// 10,240 (profiles/r6_late_ab.txt; rounds 3-6 had the switch at 16,384).  $RNNOISE_AMD_NN_LAYERS_MIN overrides (A/B runs).
int nn_layers_min_streams() {
  static const int v = [] {
    const char *e = getenv("RNNOISE_AMD_NN_LAYERS_MIN");
    return e ? atoi(e) : 10240;
  }();
  return v;
}

// Up to this many streams the vector-path network runs as the latency-oriented kernel (nn_kernels.hip: rn_nn_one_kernel, one
// 14-wave workgroup with 125 KB of LDS per stream -- one per CU, two rounds at 512 streams); $RNNOISE_AMD_NN_ONE_MAX overrides (A/B runs, 0 = never).
int nn_one_max_streams() {
  static const int v = [] {
    const char *e = getenv("RNNOISE_AMD_NN_ONE_MAX");
    return e ? atoi(e) : 512;
  }();
  return v;
}
namespace {
size_t batch_layout(RnGroupDev &g, uint8_t *base, int n) {
  uint8_t *p = base;
  size_t N = n;
  g.n_streams = n;
  g.n_stride = n;
  g.mem_hp = carve<float>(p, 2 * N);
  g.pitch_ring = carve<float>(p, RN_RING_SIZE * N);
  g.xlp_ring = carve<float>(p, RN_XRING_SIZE * N);
  g.synth_mem = carve<float>(p, RN_FRAME_SIZE * N);
  g.last_gain = carve<float>(p, N);
  g.last_period = carve<int>(p, N);
  g.lastg = carve<float>(p, RN_NB_BANDS * N);
  g.conv1_state = carve<float>(p, 130 * N);
  g.conv2_state = carve<float>(p, 256 * N);
  g.gru_state = carve<float>(p, 3 * RN_GRU * N);
  for (int k = 0; k < RN_SPEC_SLOTS; k++) {
    g.spec_X[k] = carve<float>(p, RN_SPEC_STRIDE * N);
    g.spec_P[k] = carve<float>(p, RN_SPEC_STRIDE * N);
    g.spec_E[k] = carve<float>(p, 96 * N);
  }
  g.features = carve<float>(p, 68 * N);
  g.silence = carve<int>(p, N);
  g.pitch = carve<int>(p, N);
  g.features_b = carve<float>(p, 68 * N);
  g.silence_b = carve<int>(p, N);
  g.pitch_b = carve<int>(p, N);
  g.gains = carve<float>(p, RN_NB_BANDS * N);
  g.vad = carve<float>(p, N);
  g.nn_act = carve<float>(p, RN_GRU * N);
  for (int k = 0; k < 4; k++) g.act_q[k] = carve<int8_t>(p, (N + 15) / 16 * 6144);
  g.lpc2 = carve<float>(p, 8 * N * RN_RING_SLOTS);
  g.train_clean_mem = carve<float>(p, RN_FRAME_SIZE * N);
  return (size_t)(p - base);
}

}  // namespace

// rows [first, first + count) of a batch as a group of their own (rn_dev.h: n_stride keeps the plane strides)
RnGroupDev group_view(const RnGroupDev &g, int first, int count) {
  RnGroupDev v = g;
  const size_t f = first;
  v.n_streams = count;
  v.mem_hp += 2 * f;
  v.pitch_ring += RN_RING_SIZE * f;
  v.xlp_ring += RN_XRING_SIZE * f;
  v.synth_mem += RN_FRAME_SIZE * f;
  v.last_gain += f;
  v.last_period += f;
  v.lastg += RN_NB_BANDS * f;
  v.conv1_state += 130 * f;
  v.conv2_state += 256 * f;
  v.gru_state += RN_GRU * f;
  for (int k = 0; k < RN_SPEC_SLOTS; k++) {
    v.spec_X[k] += RN_SPEC_STRIDE * f;
    v.spec_P[k] += RN_SPEC_STRIDE * f;
    v.spec_E[k] += 96 * f;
  }
  v.features += 68 * f;
  v.silence += f;
  v.pitch += f;
  v.features_b += 68 * f;
  v.silence_b += f;
  v.pitch_b += f;
  v.gains += RN_NB_BANDS * f;
  v.vad += f;
  v.nn_act += RN_GRU * f;
  v.lpc2 += 8 * f;
  v.train_clean_mem += RN_FRAME_SIZE * f;
  if (v.debug) v.debug += RN_DBG_FLOATS * f;
  return v;
}

namespace {
int batch_flush_timing(RNNoiseBatch *b) {
  for (auto &e : b->pending) {
    float ms = 0;
    HIP_OK(hipEventSynchronize(e.b));
    HIP_OK(hipEventElapsedTime(&ms, e.a, e.b));
    b->ms_sum[e.kind] += ms;
    b->pool.push_back(e);
  }
  b->pending.clear();
  return 0;
}

// A (start, stop) event pair for one kernel launch while timing is enabled; the launch helper hands it to the
// dispatch packet (hipExtLaunchKernel), the pair is read back in rnnoise_batch_kernel_ms.
struct TimedLaunch {
  RNNoiseBatch *b;
  RNNoiseBatch::Ev ev{};
  bool on;
  TimedLaunch(RNNoiseBatch *b_, int kind) : b(b_), on(b_->timing) {
    if (!on) return;
    if (!b->pool.empty()) {
      ev = b->pool.back();
      b->pool.pop_back();
    } else {
      // timing only: no cache writeback / invalidation at the event
      if (hipEventCreateWithFlags(&ev.a, hipEventDisableSystemFence) != hipSuccess ||
          hipEventCreateWithFlags(&ev.b, hipEventDisableSystemFence) != hipSuccess) {
        fprintf(stderr, "[rnnoise_amd] cannot create timing events; this launch is not timed\n");
        if (ev.a) hipEventDestroy(ev.a);
        ev.a = ev.b = nullptr;
        on = false;
        return;
      }
    }
    ev.kind = kind;
  }
  hipEvent_t start() const { return on ? ev.a : nullptr; }
  hipEvent_t stop() const { return on ? ev.b : nullptr; }
  ~TimedLaunch() {
    if (on) b->pending.push_back(ev);
  }
};

}  // namespace

// =============================================================================================
// batched API
// =============================================================================================
extern "C" int rnnoise_amd_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" RNNoiseBatch *rnnoise_batch_create(RNNModel *model, int n_streams, int device) {
  if (!model || n_streams <= 0) {
    fprintf(stderr, "[rnnoise_amd] rnnoise_batch_create: a model blob is required (no compiled-in weights)\n");
    return nullptr;
  }
  if (!rn_device_index_ok(device, rnnoise_amd_device_count())) {
    fprintf(stderr, "[rnnoise_amd] no HIP device %d (visible devices: %d); there is no CPU fallback\n", device,
            rnnoise_amd_device_count());
    return nullptr;
  }
  RNNoiseBatch *b = new RNNoiseBatch();
  b->model = model;
  b->device = device;
  b->n = n_streams;
  // same bits either way.  Up to 512 streams the latency-oriented vector kernel (one 14-wave workgroup per stream, one per CU)
  // finishes first -- measured K2 at 64 / 256 / 512 / 768 streams: 36 / 42 / 83 / 120 us against 82 / 101 / 105 / 105 us for MFMA
  // tiles of 16 streams; beyond that the MFMA paths do
  b->nn_path = (n_streams > nn_one_max_streams() && n_streams >= 16 && rn_nn_mfma_available()) ? 1 : 0;
  if (model_on_device(model, device, b->m) || tables_for_device(device, b->tb)) {
    delete b;
    return nullptr;
  }
  RnGroupDev probe{};
  b->arena_bytes = batch_layout(probe, nullptr, n_streams);
  DeviceGuard guard(device);
  if (!guard.ok || hipMalloc(&b->arena, b->arena_bytes) != hipSuccess) {
    fprintf(stderr, "[rnnoise_amd] cannot allocate %zu bytes of HBM for %d streams\n", b->arena_bytes, n_streams);
    delete b;
    return nullptr;
  }
  batch_layout(b->g, static_cast<uint8_t *>(b->arena), n_streams);
  b->scratch_gains = b->g.gains;
  b->scratch_vad = b->g.vad;
  b->features2[0] = b->g.features;
  b->silence2[0] = b->g.silence;
  b->pitch2[0] = b->g.pitch;
  b->features2[1] = b->g.features_b;
  b->silence2[1] = b->g.silence_b;
  b->pitch2[1] = b->g.pitch_b;
  if (rnnoise_batch_reset(b)) {
    rnnoise_batch_destroy(b);
    return nullptr;
  }
  return b;
}

extern "C" void rnnoise_batch_destroy(RNNoiseBatch *b) {
  if (!b) return;
  DeviceGuard guard(b->device);
  hipDeviceSynchronize();
  for (auto &e : b->pending) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
  for (auto &e : b->pool) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
  host_io_release(b);
  if (b->state_stage) hipFree(b->state_stage);
  if (b->arena) hipFree(b->arena);
  if (b->debug_buf) hipFree(b->debug_buf);
  if (b->side) hipStreamDestroy(b->side);
  if (b->side_hp) {
    hipStreamDestroy(b->side_hp);
    hipEventDestroy(b->ev_begin);
    for (int k = 0; k < 8; k++) { hipEventDestroy(b->own_hp[k]); hipEventDestroy(b->own_k1[k]); hipEventDestroy(b->own_k3[k]); }
  }
  delete b;
}

extern "C" int rnnoise_batch_size(const RNNoiseBatch *b) { return b ? b->n : -1; }

extern "C" int rnnoise_batch_reset(RNNoiseBatch *b) {
  if (!b) return -1;
  ON_DEVICE(b->device);
  // a control operation, synchronous like state export / import: whatever the batch (or anybody else) still has in flight on
  // this device is drained first, and the cleared state is in place when the call returns
  HIP_OK(hipDeviceSynchronize());
  HIP_OK(hipMemset(b->arena, 0, b->arena_bytes));
  HIP_OK(hipDeviceSynchronize());
  b->img_valid = false;
  b->parity = 0;
  b->ring_slot = 0;
  b->frame_no = 0;
  return 0;
}

extern "C" int rnnoise_batch_set_schedule(RNNoiseBatch *b, int schedule) {
  if (!b || (schedule != 0 && schedule != 1 && schedule != 9)) return -1;
  const int old = b->schedule;
  b->schedule = schedule;
  return old;
}

extern "C" int rnnoise_batch_set_nn_path(RNNoiseBatch *b, int path) {
  if (!b || path < 0 || path > 2) return -1;  // 0 vector, 1 MFMA (layer-wise from nn_layers_min_streams() up), 2 layer-wise
  if (path >= 1 && !rn_nn_mfma_available()) return -1;
  int old = b->nn_path;
  b->nn_path = path;
  return old;
}

// PCM frames are float (the reference API's sample type) or, with s16 set, int16 converted at the two ends of the step as the
// reference's only caller does (examples/rnnoise_demo.c:56,58): half the bytes over HBM and, in the host-fed path, PCIe.
int batch_process_device_impl(RNNoiseBatch *b, void *d_out_v, const void *d_in_v, float *d_vad, float *d_gains, int n_frames,
                              void *hip_stream, bool s16, const FrameIoHooks *hk) {
  if (!b || !d_out_v || !d_in_v || n_frames < 0) return -1;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  ON_DEVICE(b->device);
  const size_t N = b->n, esz = s16 ? sizeof(short) : sizeof(float);
  const char *d_in = static_cast<const char *>(d_in_v);
  char *d_out = static_cast<char *>(d_out_v);
  auto buf = [&](int f) -> size_t { return hk ? (size_t)(f % hk->ring) : (size_t)f; };  // frame f's place in the caller's buffers
  // Multi-frame calls are software-pipelined over three streams: C runs the high-pass of frames up to
  // f+2, B the analysis of frame f+1, A (the caller's stream) network + synthesis of frame f.
  // What makes that legal:
  //   * the pitch ring has 6 slots and analysis(g) reads slots g-3..g, so high-pass(f) only has to wait
  //     for analysis(f-3);
  //   * the spectra rotate through 3 slots and the per-step scratch (features, silence, pitch) is
  //     double-buffered, so analysis(f) only has to wait for synthesis(f-2);
  //   * every other piece of state is touched by one kernel only, in frame order on its own stream.
  // RNNOISE_AMD_PIPE (A/B runs only): 9 = no side streams, 1 = K0 on a side stream, 2 = K0 and K1 on side streams.
  // Measured after the fence-free events: the 3-stream schedule is the best or within noise of the best from 1 K to
  // 64 K streams (65,536: 20.2 M frames/s vs 20.0 M on one stream, 19.6 M with only K0 aside), so it is the only default.
  static const int pipe_env = [] { const char *e = getenv("RNNOISE_AMD_PIPE"); return e ? atoi(e) : 0; }();
  const int pipe_force = b->schedule ? b->schedule : pipe_env;
  const bool pipelined = n_frames > 1 && pipe_force != 9;
  const bool side_k1 = pipelined && pipe_force != 1;
  // $RNNOISE_AMD_SIDE_PRIO = <k1>,<hp> (A/B runs): queue priorities of the two side streams, -1 high / 0 normal / 1 low (the caller's
  // stream, which carries network + synthesis, is whatever the caller made it: normal for torch's)
  static const int side_prio[2] = {[] { const char *e = RN_LAB_ENV("SIDE_PRIO"); return e ? atoi(e) : 0; }(),
                                   [] { const char *e = RN_LAB_ENV("SIDE_PRIO"); const char *c = e ? strchr(e, ',') : nullptr; return c ? atoi(c + 1) : 0; }()};
  if (side_k1 && !b->side) HIP_OK(hipStreamCreateWithPriority(&b->side, hipStreamNonBlocking, side_prio[0]));
  if (pipelined && !b->side_hp) {
    HIP_OK(hipStreamCreateWithPriority(&b->side_hp, hipStreamNonBlocking, side_prio[1]));
    // ordering between streams of ONE device: no system-scope fence (it writes back and invalidates the caches at
    // every record, which the next kernels then pay for)
    // ($RNNOISE_AMD_EVENT_FENCE=1, A/B runs only: ordering events with the system-scope fence back on)
    static const bool sys_fence = [] { const char *e = RN_LAB_ENV("EVENT_FENCE"); return e && atoi(e) == 1; }();
    const unsigned evf = hipEventDisableTiming | (sys_fence ? 0u : (unsigned)hipEventDisableSystemFence);
    HIP_OK(hipEventCreateWithFlags(&b->ev_begin, evf));
    for (int k = 0; k < 8; k++) {
      HIP_OK(hipEventCreateWithFlags(&b->own_hp[k], evf));
      HIP_OK(hipEventCreateWithFlags(&b->own_k1[k], evf));
      HIP_OK(hipEventCreateWithFlags(&b->own_k3[k], evf));
    }
  }
  hipStream_t sb = side_k1 ? b->side : st, sc = pipelined ? b->side_hp : st;
  if (pipelined) {  // B and C start after everything already queued on the caller's stream
    HIP_OK(hipEventRecord(b->ev_begin, st));
    if (side_k1) HIP_OK(hipStreamWaitEvent(b->side, b->ev_begin, 0));
    HIP_OK(hipStreamWaitEvent(b->side_hp, b->ev_begin, 0));
  }
  auto frame_group = [&](int f) {
    RnGroupDev g = b->g;
    const int c = (int)((b->frame_no + f) & 1);
    g.features = b->features2[c];
    g.silence = b->silence2[c];
    g.pitch = b->pitch2[c];
    g.vad = d_vad ? d_vad + buf(f) * N : b->scratch_vad;
    g.gains = d_gains ? d_gains + buf(f) * N * RN_NB_BANDS : b->scratch_gains;
    return g;
  };
  auto highpass = [&](int f) -> int {  // K0 of frame f on stream sc
    // completion events ride in the dispatch packets (stop event of hipExtLaunchKernel): no record packets between
    // the kernels of a stream
    if (pipelined && f >= 3) HIP_OK(hipStreamWaitEvent(sc, b->cur_k1[(f - 3) & 7], 0));
    // ... and not before synthesis(f-4) is done, which is when analysis(f-2) starts: left to the ring alone, the high-pass
    // starts the moment analysis(f-3) ends -- together with the GRU layer kernels of frame f-4.  Its 1024 waves are one per
    // SIMD for 0.18 ms, and a GRU workgroup (2 waves x 240 VGPRs per SIMD) does not fit beside even one of them: the first
    // layer kernel of every frame waited that long (rocprofv3 timeline: 283 us instead of 115).  Beside the analysis kernel
    // (4 waves x 56 VGPRs per SIMD) it costs nothing.
    static const bool hp_early = RN_LAB_ENV("HP_EARLY") != nullptr;  // A/B runs only: the ring-bound start
    if (side_k1 && f >= 4 && !hp_early) HIP_OK(hipStreamWaitEvent(sc, b->cur_k3[(f - 4) & 7], 0));
    // ... and the same concern when only the high-pass runs aside (schedule 1: the host-fed path): there analysis(f-2) follows
    // synthesis(f-3) on the main stream, so that is the event to start behind -- the high-pass of frame f is launched after it (see the
    // frame loop).  Started at the ring's earliest moment it ran beside the layer kernels of frame f-3: network 0.73 ms instead of 0.57
    // (profiles/r5_hostio_sdma.txt).
    if (pipelined && !side_k1 && f >= 3 && !hp_early) HIP_OK(hipStreamWaitEvent(sc, b->cur_k3[(f - 3) & 7], 0));
    if (hk && hk->before_hp(f, sc)) return -1;
    {
      TimedLaunch t(b, 3);
      b->cur_hp[f & 7] = t.on ? t.stop() : (pipelined ? b->own_hp[f & 7] : nullptr);
      HIP_OK(rn_launch_hp(&b->g, d_in + buf(f) * N * RN_FRAME_SIZE * esz, s16, ((b->ring_slot + f) % RN_RING_SLOTS) | (pipelined ? 512 : 0), sc, t.start(),
                          b->cur_hp[f & 7]));
    }
    if (hk && hk->after_hp(f, sc)) return -1;
    return 0;
  };
  auto analysis = [&](int f) -> int {  // K1 of frame f on stream sb
    RnGroupDev g = frame_group(f);
    if (pipelined) {
      HIP_OK(hipStreamWaitEvent(sb, b->cur_hp[f & 7], 0));
      if (side_k1 && f >= 2) HIP_OK(hipStreamWaitEvent(sb, b->cur_k3[(f - 2) & 7], 0));
    }
    {
      TimedLaunch t(b, 0);
      b->cur_k1[f & 7] = t.on ? t.stop() : (pipelined ? b->own_k1[f & 7] : nullptr);
      HIP_OK(rn_launch_analysis(&g, &b->tb, (b->ring_slot + f) % RN_RING_SLOTS, (b->parity + f) % RN_SPEC_SLOTS, sb, t.start(),
                                b->cur_k1[f & 7]));
    }
    return 0;
  };
#if RN_INSTRUMENT
  // LAB ($RNNOISE_AMD_FUSE_K3K1=1, instrumented library; profiles/r6_fused_k3k1.txt): synthesis(f - 1) as the prologue of analysis(f) in ONE
  // kernel on the main stream -- [K3(f-1) . K1(f)] -> K2(f) -> [K3(f) . K1(f+1)] -> ... -- the high-pass ahead on its side stream, started
  // behind the network of frame f - 3, and one trailing stand-alone synthesis at the end of the call
  static const bool fuse_env = [] { const char *e = RN_LAB_ENV("FUSE_K3K1"); return e && atoi(e) == 1; }();
  if (fuse_env && n_frames > 1 && !hk && pipe_force != 9 && b->n >= 6144) {
    const bool whole = b->g.n_streams == b->g.n_stride && (size_t)b->g.n_streams * RN_GRU * 4 < (1ull << 32);
    if (!(whole && (b->nn_path == 2 || (b->nn_path == 1 && b->n >= nn_layers_min_streams())))) return -1;  // (layer-wise network only)
    for (int f = 0; f < 3 && f < n_frames; f++)
      if (highpass(f)) return -1;
    for (int f = 0; f < n_frames; f++) {
      RnGroupDev g = frame_group(f), gs = frame_group(f > 0 ? f - 1 : 0);
      const int cur = (b->parity + f) % RN_SPEC_SLOTS, prev = (cur + RN_SPEC_SLOTS - 1) % RN_SPEC_SLOTS, pprev = (prev + RN_SPEC_SLOTS - 1) % RN_SPEC_SLOTS;
      HIP_OK(hipStreamWaitEvent(st, b->cur_hp[f & 7], 0));
      {
        TimedLaunch t(b, 0);
        b->cur_k1[f & 7] = t.on ? t.stop() : b->own_k1[f & 7];
        HIP_OK(rn_launch_analysis_synth(&g, &gs, &b->tb, (b->ring_slot + f) % RN_RING_SLOTS, cur, d_out + buf(f > 0 ? f - 1 : 0) * N * RN_FRAME_SIZE * esz, s16,
                                        f > 0 ? prev : -1, pprev, st, t.start(), b->cur_k1[f & 7]));
      }
      {
        if (!b->img_valid) HIP_OK(rn_launch_nn_requant(&g, st));
        b->img_valid = true;
        std::unique_ptr<TimedLaunch> tl[5];
        hipEvent_t ev[5][2] = {};
        const int nl = rn_nn_layers_launches();
        for (int i = 0; i < nl; i++) {
          tl[i].reset(new TimedLaunch(b, 1));
          ev[i][0] = tl[i]->start();
          ev[i][1] = tl[i]->stop();
        }
        if (!ev[nl - 1][1]) ev[nl - 1][1] = b->own_k3[f & 7];
        b->cur_k3[f & 7] = ev[nl - 1][1];  // (here: "the network of frame f is done" -- what the high-pass three frames ahead starts behind)
        HIP_OK(rn_launch_nn_layers(&g, &b->m, &b->tb, st, ev));
      }
      if (f + 3 < n_frames) {
        HIP_OK(hipStreamWaitEvent(sc, b->cur_k3[f & 7], 0));  // (beside the fused kernel of frame f + 1, not beside this frame's layer kernels)
        if (highpass(f + 3)) return -1;
      }
      b->launches += b->timing ? 1 : 0;
    }
    {
      const int f = n_frames - 1;
      RnGroupDev g = frame_group(f);
      const int cur = (b->parity + f) % RN_SPEC_SLOTS, prev = (cur + RN_SPEC_SLOTS - 1) % RN_SPEC_SLOTS;
      TimedLaunch t(b, 2);
      HIP_OK(rn_launch_synthesis(&g, &b->tb, d_out + buf(f) * N * RN_FRAME_SIZE * esz, s16, cur, prev, st, t.start(), t.stop()));
    }
    b->parity = (b->parity + n_frames) % RN_SPEC_SLOTS;
    b->ring_slot = (b->ring_slot + n_frames) % RN_RING_SLOTS;
    b->frame_no += n_frames;
    return 0;
  }
#endif
  if (pipelined) {
    for (int f = 0; f < 3 && f < n_frames; f++)
      if (highpass(f)) return -1;
    if (analysis(0)) return -1;
  }
  for (int f = 0; f < n_frames; f++) {
    RnGroupDev g = frame_group(f);
    const int cur = (b->parity + f) % RN_SPEC_SLOTS, prev = (cur + RN_SPEC_SLOTS - 1) % RN_SPEC_SLOTS;
    if (!pipelined) {
      if (highpass(f) || analysis(f)) return -1;
    } else {
      if (side_k1 && f + 3 < n_frames && highpass(f + 3)) return -1;
      if (f + 1 < n_frames && analysis(f + 1)) return -1;
      if (side_k1) HIP_OK(hipStreamWaitEvent(st, b->cur_k1[f & 7], 0));
    }
    if (hk && hk->before_nn(f, st)) return -1;
    {
      // (the layer images are indexed by tile of the whole batch; the layer kernels use 32-bit byte offsets into a state plane)
      const bool whole = g.n_streams == g.n_stride && (size_t)g.n_streams * RN_GRU * 4 < (1ull << 32);
      if (whole && (b->nn_path == 2 || (b->nn_path == 1 && b->n >= nn_layers_min_streams()))) {
        if (!b->img_valid) HIP_OK(rn_launch_nn_requant(&g, st));
        b->img_valid = true;
        // four or five launches, each timed on its own (kind 1: the durations add up to the network's)
        std::unique_ptr<TimedLaunch> tl[5];
        hipEvent_t ev[5][2] = {};
        for (int i = 0, nl = rn_nn_layers_launches(); i < nl; i++) {
          tl[i].reset(new TimedLaunch(b, 1));
          ev[i][0] = tl[i]->start();
          ev[i][1] = tl[i]->stop();
        }
        HIP_OK(rn_launch_nn_layers(&g, &b->m, &b->tb, st, ev));
      } else {
        TimedLaunch t(b, 1);
        b->img_valid = false;
        if (b->nn_path >= 1) HIP_OK(rn_launch_nn_mfma(&g, &b->m, &b->tb, st, t.start(), t.stop(), !pipelined));
        else if (g.n_streams <= nn_one_max_streams()) HIP_OK(rn_launch_nn_one(&g, &b->m, &b->tb, st, t.start(), t.stop()));
        else HIP_OK(rn_launch_nn_vector(&g, &b->m, &b->tb, st, t.start(), t.stop()));
      }
    }
    {
      TimedLaunch t(b, 2);
      b->cur_k3[f & 7] = t.on ? t.stop() : (pipelined ? b->own_k3[f & 7] : nullptr);
      HIP_OK(rn_launch_synthesis(&g, &b->tb, d_out + buf(f) * N * RN_FRAME_SIZE * esz, s16, cur, prev, st, t.start(), b->cur_k3[f & 7]));
    }
    if (hk && hk->after_k3(f, st)) return -1;
    // (schedule 1: the high-pass three frames ahead goes out HERE, behind the synthesis launch whose end it starts at)
    if (pipelined && !side_k1 && f + 3 < n_frames && highpass(f + 3)) return -1;
    b->launches += b->timing ? 1 : 0;
  }
  b->parity = (b->parity + n_frames) % RN_SPEC_SLOTS;
  b->ring_slot = (b->ring_slot + n_frames) % RN_RING_SLOTS;
  b->frame_no += n_frames;
  return 0;
}

extern "C" int rnnoise_batch_process_device(RNNoiseBatch *b, float *d_out, const float *d_in, float *d_vad,
                                            float *d_gains, int n_frames, void *hip_stream) {
  return batch_process_device_impl(b, d_out, d_in, d_vad, d_gains, n_frames, hip_stream, false);
}

extern "C" int rnnoise_batch_process_device_s16(RNNoiseBatch *b, short *d_out, const short *d_in, float *d_vad,
                                                float *d_gains, int n_frames, void *hip_stream) {
  return batch_process_device_impl(b, d_out, d_in, d_vad, d_gains, n_frames, hip_stream, true);
}

// ---- training-feature extraction (reference loop src/dump_features.c:466-491) ----
extern "C" int rnnoise_batch_train_features_device(RNNoiseBatch *b, float *d_records, const float *d_clean,
                                                   const float *d_noisy, const float *d_vad, const int *d_lowpass,
                                                   const int *d_band_lp, const int *d_noise_free, int n_frames,
                                                   void *hip_stream) {
  if (!b || !d_records || !d_clean || !d_noisy || !d_vad || !d_lowpass || !d_band_lp || !d_noise_free || n_frames < 0)
    return -1;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  ON_DEVICE(b->device);
  const size_t N = b->n;
  for (int f = 0; f < n_frames; f++) {
    RnTrainArgs tr;
    tr.clean = d_clean + f * N * RN_FRAME_SIZE;
    tr.clean_mem = b->g.train_clean_mem;
    tr.vad = d_vad + f * N;
    tr.lowpass = d_lowpass;
    tr.band_lp = d_band_lp;
    tr.noise_free = d_noise_free;
    tr.rec = d_records + f * N * 98;
    HIP_OK(rn_launch_train_features(&b->g, &b->tb, d_noisy + f * N * RN_FRAME_SIZE, b->ring_slot, b->parity, &tr, st));
    b->parity = (b->parity + 1) % RN_SPEC_SLOTS;
    b->ring_slot = (b->ring_slot + 1) % RN_RING_SLOTS;
  }
  return 0;
}

extern "C" int rnnoise_batch_train_features(RNNoiseBatch *b, float *records, const float *clean, const float *noisy,
                                            const float *vad, const int *lowpass, const int *band_lp,
                                            const int *noise_free, int n_frames) {
  if (!b || !records || !clean || !noisy || !vad || !lowpass || !band_lp || !noise_free || n_frames <= 0) return -1;
  ON_DEVICE(b->device);
  const size_t N = b->n, fb = (size_t)n_frames * N * RN_FRAME_SIZE * 4;
  char *dev = nullptr;
  const size_t o_clean = 0, o_noisy = fb, o_vad = 2 * fb, o_rec = o_vad + (size_t)n_frames * N * 4,
               o_lp = o_rec + (size_t)n_frames * N * 98 * 4, o_bl = o_lp + N * 4, o_nf = o_bl + N * 4, total = o_nf + N * 4;
  HIP_OK(hipMalloc((void **)&dev, total));
  int rc = -1;
  if (hipMemcpy(dev + o_clean, clean, fb, hipMemcpyHostToDevice) == hipSuccess &&
      hipMemcpy(dev + o_noisy, noisy, fb, hipMemcpyHostToDevice) == hipSuccess &&
      hipMemcpy(dev + o_vad, vad, (size_t)n_frames * N * 4, hipMemcpyHostToDevice) == hipSuccess &&
      hipMemcpy(dev + o_lp, lowpass, N * 4, hipMemcpyHostToDevice) == hipSuccess &&
      hipMemcpy(dev + o_bl, band_lp, N * 4, hipMemcpyHostToDevice) == hipSuccess &&
      hipMemcpy(dev + o_nf, noise_free, N * 4, hipMemcpyHostToDevice) == hipSuccess &&
      rnnoise_batch_train_features_device(b, (float *)(dev + o_rec), (const float *)(dev + o_clean),
                                          (const float *)(dev + o_noisy), (const float *)(dev + o_vad),
                                          (const int *)(dev + o_lp), (const int *)(dev + o_bl), (const int *)(dev + o_nf),
                                          n_frames, nullptr) == 0 &&
      hipDeviceSynchronize() == hipSuccess &&
      hipMemcpy(records, dev + o_rec, (size_t)n_frames * N * 98 * 4, hipMemcpyDeviceToHost) == hipSuccess)
    rc = 0;
  hipFree(dev);
  return rc;
}

#define D2H(dst, src, count) HIP_OK(hipMemcpy(dst, src, (count) * 4, hipMemcpyDeviceToHost))
#define H2D(dst, src, count) HIP_OK(hipMemcpy(dst, src, (count) * 4, hipMemcpyHostToDevice))

// State migration: one gather / scatter kernel (state_kernels.hip) and one copy per call.  Synchronous with everything
// the batch has in flight (the caller's streams are not known here, so the device is drained first).
extern "C" int rnnoise_batch_export_state(RNNoiseBatch *b, int s, float *f) {
  if (!b || !f || s < 0 || s >= b->n) return -1;
  ON_DEVICE(b->device);
  HIP_OK(hipDeviceSynchronize());
  if (!b->state_stage) HIP_OK(hipMalloc((void **)&b->state_stage, RN_STATE_FLOATS * sizeof(float)));
  const RnGroupDev v = group_view(b->g, s, 1);
  HIP_OK(rn_launch_state_gather(&v, b->state_stage, (b->ring_slot + RN_RING_SLOTS - 1) % RN_RING_SLOTS,
                                (b->parity + RN_SPEC_SLOTS - 1) % RN_SPEC_SLOTS, nullptr));
  D2H(f, b->state_stage, RN_STATE_FLOATS);  // (a blocking copy on the null stream: ordered after the kernel)
  return 0;
}

extern "C" int rnnoise_batch_import_state(RNNoiseBatch *b, int s, const float *f) {
  if (!b || !f || s < 0 || s >= b->n) return -1;
  if (memcmp(f + RN_OFF_ANALYSIS, f + RN_OFF_PITCH_BUF + RN_PITCH_BUF_SIZE - RN_FRAME_SIZE, RN_FRAME_SIZE * 4)) {
    fprintf(stderr, "[rnnoise_amd] import_state: analysis_mem differs from the tail of pitch_buf\n");
    return -1;
  }
  ON_DEVICE(b->device);
  HIP_OK(hipDeviceSynchronize());
  if (!b->state_stage) HIP_OK(hipMalloc((void **)&b->state_stage, RN_STATE_FLOATS * sizeof(float)));
  H2D(b->state_stage, f, RN_STATE_FLOATS);
  b->img_valid = false;
  const RnGroupDev v = group_view(b->g, s, 1);
  HIP_OK(rn_launch_state_scatter(&v, b->state_stage, (b->ring_slot + RN_RING_SLOTS - 1) % RN_RING_SLOTS,
                                 (b->parity + RN_SPEC_SLOTS - 1) % RN_SPEC_SLOTS, nullptr));
  HIP_OK(hipStreamSynchronize(nullptr));
  return 0;
}

extern "C" int rnnoise_batch_debug_last(RNNoiseBatch *b, float *features, int *silence, int *pitch) {
  if (!b) return -1;
  ON_DEVICE(b->device);
  HIP_OK(hipDeviceSynchronize());
  if (features) {
    std::vector<float> tmp((size_t)b->n * 68);
    D2H(tmp.data(), b->features2[(b->frame_no + 1) & 1], tmp.size());
    for (int s = 0; s < b->n; s++) memcpy(features + (size_t)s * RN_NB_FEATURES, tmp.data() + (size_t)s * 68, RN_NB_FEATURES * 4);
  }
  if (silence) D2H(silence, b->silence2[(b->frame_no + 1) & 1], b->n);
  if (pitch) D2H(pitch, b->pitch2[(b->frame_no + 1) & 1], b->n);
  return 0;
}

#if RN_INSTRUMENT  // ---- test / measurement taps: instrumented build only (include/rnnoise_amd_debug.h) ----
// pitch stage taps of the last step ([N][RN_DBG_FLOATS]); the first call (dst==NULL) arms them
extern "C" int rnnoise_batch_debug_pitch(RNNoiseBatch *b, float *dst) {
  if (!b) return -1;
  ON_DEVICE(b->device);
  HIP_OK(hipDeviceSynchronize());
  if (!b->debug_buf) {
    HIP_OK(hipMalloc((void **)&b->debug_buf, (size_t)b->n * RN_DBG_FLOATS * 4));
    HIP_OK(hipMemset(b->debug_buf, 0, (size_t)b->n * RN_DBG_FLOATS * 4));
    b->g.debug = b->debug_buf;
  }
  if (dst) D2H(dst, b->debug_buf, (size_t)b->n * RN_DBG_FLOATS);
  return 0;
}

// n independent 960-point transforms through the register-resident FFT (fft_reg.h), `reps` passes each (the spectrum is
// fed back as the next input); variant 0 = all exchanges through ds_bpermute, 1 = the DPP / swizzle forms the kernels use.
// in / out: [n][960][2] host floats (natural order; the 1/960 input scale of kiss_fft.c:582 is applied on the first pass);
// clocks (optional): [n] shader clocks per wave; xlane (optional): [2][6][64] source lane delivered by each exchange
// primitive for xor masks 1,2,4,8,16,32.  Tests and tools only.
extern "C" int rnnoise_amd_debug_fft(int device, int variant, float *out, const float *in, int n, int reps,
                                     unsigned long long *clocks, int *xlane) {
  if (!out || !in || n <= 0 || reps <= 0) return -1;
  ON_DEVICE(device);
  RnTablesDev tb;
  if (tables_for_device(device, tb)) return -1;
  const size_t fb = (size_t)n * 960 * 2 * 4;
  char *d = nullptr;
  HIP_OK(hipMalloc((void **)&d, 2 * fb + (size_t)n * 8 + 2 * 6 * 64 * 4));
  float *d_in = (float *)d, *d_out = (float *)(d + fb);
  unsigned long long *d_clk = (unsigned long long *)(d + 2 * fb);
  int *d_x = (int *)(d + 2 * fb + (size_t)n * 8);
  int rc = -1;
  if (hipMemcpy(d_in, in, fb, hipMemcpyHostToDevice) == hipSuccess &&
      rn_launch_fft_probe(variant, d_in, d_out, d_clk, n, reps, &tb, nullptr) == hipSuccess &&
      rn_launch_xlane_probe(d_x, nullptr) == hipSuccess && hipStreamSynchronize(nullptr) == hipSuccess &&
      hipMemcpy(out, d_out, fb, hipMemcpyDeviceToHost) == hipSuccess &&
      (!clocks || hipMemcpy(clocks, d_clk, (size_t)n * 8, hipMemcpyDeviceToHost) == hipSuccess) &&
      (!xlane || hipMemcpy(xlane, d_x, 2 * 6 * 64 * 4, hipMemcpyDeviceToHost) == hipSuccess))
    rc = 0;
  hipFree(d);
  return rc;
}

// out[i] = (float)log10(1e-2 + (double)ex[i]) evaluated on the device by the feature stage's function (host buffers; tests only).
// ex == null: the n floats with bit patterns first_bits, first_bits + 1, ...; model 0: as the kernels of this process evaluate it
// (rnnoise_amd_log10_model()), 1: the device library's log10 whatever the process uses
extern "C" int rnnoise_amd_debug_log_energy_range(int device, float *out, const float *ex, unsigned first_bits, unsigned n, int model) {
  if (!out || n == 0) return -1;
  ON_DEVICE(device);
  RnTablesDev tb;
  if (tables_for_device(device, tb)) return -1;
  float *d = nullptr;
  HIP_OK(hipMalloc((void **)&d, (size_t)n * (ex ? 8 : 4)));
  int rc = -1;
  if ((!ex || hipMemcpy(d + n, ex, (size_t)n * 4, hipMemcpyHostToDevice) == hipSuccess) &&
      rn_launch_log_energy(ex ? d + n : nullptr, first_bits, d, n, model == 1 ? nullptr : tb.log_tab, nullptr) == hipSuccess &&
      hipStreamSynchronize(nullptr) == hipSuccess && hipMemcpy(out, d, (size_t)n * 4, hipMemcpyDeviceToHost) == hipSuccess)
    rc = 0;
  hipFree(d);
  return rc;
}
extern "C" int rnnoise_amd_debug_log_energy(int device, float *out, const float *ex, int n) {
  if (!ex || n <= 0) return -1;
  return rnnoise_amd_debug_log_energy_range(device, out, ex, 0, (unsigned)n, 0);
}

// the GRU layer kernel's row-buffer check (nn_layers.hip: CHK instantiations): copies the log out and clears it
extern "C" hipError_t rn_gru_race_log_read(unsigned *out, int words);
extern "C" int rnnoise_amd_debug_gru_race(int device, unsigned *log, int words) {
  if (!log) return -1;
  ON_DEVICE(device);
  HIP_OK(rn_gru_race_log_read(log, words));
  return 0;
}

#endif  // RN_INSTRUMENT

extern "C" int rnnoise_batch_enable_timing(RNNoiseBatch *b, int on) {
  if (!b) return -1;
  if (batch_flush_timing(b)) return -1;
  b->timing = on != 0;
  for (double &v : b->ms_sum) v = 0;
  b->launches = 0;
  return 0;
}

extern "C" int rnnoise_batch_kernel_ms(RNNoiseBatch *b, double ms[4], long *launches) {
  if (!b || !ms) return -1;
  if (batch_flush_timing(b)) return -1;
  for (int k = 0; k < 4; k++) ms[k] = b->launches ? b->ms_sum[k] / b->launches : 0.0;
  if (launches) *launches = b->launches;
  for (double &v : b->ms_sum) v = 0;
  b->launches = 0;
  return 0;
}

