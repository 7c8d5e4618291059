// host_io.cpp -- host-fed calls: pinned caller memory through a frame-granular ring beside one pipelined
// device call, pageable memory through double-buffered bounce chunks.
#include "shim.h"

#include <hsa/amd_hsa_signal.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <chrono>

// ---- host-fed path (pinned, double-buffered H2D / D2H) ----
namespace {
bool host_pinned(const void *p) {  // memory the DMA engines can reach directly (hipHostMalloc / hipHostRegister)
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return a.type == hipMemoryTypeHost;
}
}  // namespace


// ---------------------------------------------------------------------------------------------
// Copy mode "sdma": the two PCIe directions on two NAMED copy engines, at once.
// HIP picks the engine of a hipMemcpyAsync itself, and with an upload and a download in flight it runs one of them as a blit kernel
// whose PCIe-bound stores stall the kernels beside it (profiles/r3_hostio_traces.txt) -- which is why the other modes move an int16
// frame's 2 x 63 MB one direction at a time.  Underneath HIP, ROCr takes copies for a named engine
// (hsa_amd_memory_async_copy_on_engine); what it does not have is HIP's stream ordering, so the three cross-dependencies of the
// frame ring are built from pieces that profiles/r5_sdma_probe.txt shows working on this runtime:
//   copy engine -> stream  a second, 8-byte copy queued behind the payload on the same engine writes the running count of landed
//                          frames into a word of device memory; the stream waits for it with hipStreamWaitValue64;
//   stream -> copy engine  a one-thread kernel on the stream stores 0 (system-scope release) into the value word of an HSA signal that
//                          the copy lists as its dependency (rn_release_store_kernel);
//   copy engine -> host    the completion signal of the call's last copy (hsa_signal_wait).
// ---------------------------------------------------------------------------------------------
struct Sdma {
  bool ok = false, hsa_ref = false;
  hsa_agent_t gpu{}, cpu{};
  hsa_amd_sdma_engine_id_t e_up{}, e_dn{};
  uint64_t *d_flags = nullptr;  // device memory: [0] uploads landed so far, [32] downloads landed so far (separate lines)
  uint64_t *h_seq = nullptr;    // pinned [SEQ]: the source words of the count copies
  static constexpr int SEQ = 2048;
  uint64_t up_count = 0, dn_count = 0;
  std::vector<hsa_signal_t> pool;
  size_t used = 0;
  hsa_signal_t take(hsa_signal_value_t v) {
    if (used == pool.size()) {
      hsa_signal_t s{};
      if (hsa_signal_create(v, 0, nullptr, &s) != HSA_STATUS_SUCCESS) return hsa_signal_t{0};
      pool.push_back(s);
    }
    hsa_signal_t s = pool[used++];
    hsa_signal_store_relaxed(s, v);
    return s;
  }
};

namespace {
// Wait until the signal drops below 1, for at most `seconds` of WALL-CLOCK time.  hsa_signal_wait's timeout hint is in ticks of the
// HSA system timestamp (HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY, 100 MHz here -- not nanoseconds: round 5's "30 s" was 300 s), and the
// call may return early with the condition unmet: so it is asked for 0.1 s worth of ticks at a time until a steady clock says stop.
bool sdma_wait(hsa_signal_t s, double seconds) {
  uint64_t freq = 0;
  if (hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY, &freq) != HSA_STATUS_SUCCESS || !freq) freq = 100000000ull;
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(seconds);
  for (;;) {
    if (hsa_signal_wait_scacquire(s, HSA_SIGNAL_CONDITION_LT, 1, freq / 10, HSA_WAIT_STATE_BLOCKED) < 1) return true;
    if (std::chrono::steady_clock::now() >= deadline) return false;
  }
}
// The value word of an HSA signal as a kernel can store to it.  This reaches INTO ROCr: hsa_signal_t::handle is the address of an
// amd_signal_t (hsa/amd_hsa_signal.h, a public header of the runtime, but the identity handle == address is the runtime's own
// business).  Developed and soaked on the runtime profiles/r6_hostio_timeline.txt names (ROCm 7.2.0's ROCr); sdma_selftest() below
// proves the assumption -- a kernel's store into this word releases a copy that lists the signal as its dependency -- on whatever
// runtime the process has, once per batch, BEFORE the mode is used; a runtime where it does not hold gets the runtime's own copies.
void *sdma_value_word(hsa_signal_t s) { return static_cast<void *>(const_cast<int64_t *>(&reinterpret_cast<amd_signal_t *>(s.handle)->value)); }

bool sdma_owner(const void *p, hsa_agent_t &agent, const void *&agent_ptr) {  // who owns p, and p as that side's engines address it
  hsa_amd_pointer_info_t info;
  memset(&info, 0, sizeof info);
  info.size = sizeof info;
  if (hsa_amd_pointer_info(p, &info, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS || info.type == HSA_EXT_POINTER_TYPE_UNKNOWN) return false;
  agent = info.agentOwner;
  agent_ptr = p;
  if (info.type == HSA_EXT_POINTER_TYPE_LOCKED && info.hostBaseAddress && info.agentBaseAddress)  // hipHostRegister-ed memory
    agent_ptr = static_cast<const char *>(info.agentBaseAddress) + (static_cast<const char *>(p) - static_cast<const char *>(info.hostBaseAddress));
  return true;
}
// Everything copy mode "sdma" assumes about the runtime, exercised once with 8 bytes and wall-clock deadlines before the mode is
// trusted with a caller's frames (a runtime where the assumptions half-hold must fail HERE, at set-up, not as a stall in
// the middle of a call):
//   stream -> engine   an upload on the named upload engine that lists a signal as its dependency starts when a KERNEL stores 0 into
//                      the signal's value word (sdma_value_word: the amd_signal_t layout);
//   engine -> stream   the word the upload writes into device memory releases a stream parked on it by hipStreamWaitValue64;
//   engine -> engine   a download on the named download engine that depends on the upload's completion signal brings the word back;
//   engine -> host     hsa_signal_wait sees the download complete.
// Returns nullptr when all four hold, else the name of the step that did not.
const char *sdma_selftest(Sdma *s) {
  const uint64_t magic = 0x5EEDF00D5EEDull;
  hipStream_t st = nullptr;
  if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return "self-test: stream";
  const hsa_signal_t dep = s->take(1), up = s->take(1), dn = s->take(1);
  const char *bad = nullptr;
  s->h_seq[0] = magic;
  s->h_seq[1] = 0;
  if (!dep.handle || !up.handle || !dn.handle) bad = "self-test: signals";
  else if (hsa_amd_memory_async_copy_on_engine(&s->d_flags[16], s->gpu, &s->h_seq[0], s->cpu, 8, 1, &dep, up, s->e_up, false) != HSA_STATUS_SUCCESS)
    bad = "self-test: upload on the named engine";
  else if (hsa_amd_memory_async_copy_on_engine(&s->h_seq[1], s->cpu, &s->d_flags[16], s->gpu, 8, 1, &up, dn, s->e_dn, false) != HSA_STATUS_SUCCESS)
    bad = "self-test: download on the named engine";
  if (bad) {
    if (dep.handle) hsa_signal_store_screlease(dep, 0);  // (whatever was queued runs out)
    if (dn.handle) (void)sdma_wait(dn, 2.0);
  } else {
    // nothing may have moved yet: the upload waits for `dep`
    if (hsa_signal_load_scacquire(up) < 1) bad = "self-test: the upload did not wait for its dependency";
    else if (rn_launch_release_store(sdma_value_word(dep), 0, st) != hipSuccess) bad = "self-test: release kernel";
    else if (hipStreamWaitValue64(st, &s->d_flags[16], magic, hipStreamWaitValueEq, ~0ull) != hipSuccess) bad = "self-test: hipStreamWaitValue64";
    if (bad) hsa_signal_store_screlease(dep, 0);
    if (!sdma_wait(dn, 2.0)) bad = bad ? bad : "self-test: a kernel's store into the signal's value word did not release the copy (amd_signal_t layout?)";
    else if (!bad && s->h_seq[1] != magic) bad = "self-test: the word did not come back";
    // the stream parked on the word: released by the upload -- or by hand, so that nothing stays parked behind a failed test
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(2);
    while (hipStreamQuery(st) == hipErrorNotReady && std::chrono::steady_clock::now() < deadline) {}
    if (hipStreamQuery(st) == hipErrorNotReady) {
      bad = bad ? bad : "self-test: hipStreamWaitValue64 did not see the engine's write";
      (void)hipMemcpy(&s->d_flags[16], &magic, 8, hipMemcpyHostToDevice);
    }
  }
  (void)hipStreamSynchronize(st);
  (void)hipGetLastError();
  (void)hipStreamDestroy(st);
  s->used = 0;  // (the three signals go back to the pool)
  return bad;
}

Sdma *sdma_get(RNNoiseBatch *b, const void *host_ptr) {
  RNNoiseBatch::HostIo &io = b->io;
  if (io.sdma) return io.sdma->ok ? io.sdma : nullptr;
  Sdma *s = io.sdma = new Sdma();
  auto fail = [&](const char *what, hsa_status_t st) -> Sdma * {
    const char *m = nullptr;
    hsa_status_string(st, &m);
    fprintf(stderr, "[rnnoise_amd] copy mode sdma unavailable (%s: %s); using the runtime's copies\n", what, m ? m : "?");
    return nullptr;
  };
  hsa_status_t st = hsa_init();  // (reference-counted; the HIP runtime holds the first reference; ours is dropped in sdma_free)
  if (st != HSA_STATUS_SUCCESS) return fail("hsa_init", st);
  s->hsa_ref = true;
  const void *ap = nullptr;
  if (!sdma_owner(io.ring_mem, s->gpu, ap) || !sdma_owner(host_ptr, s->cpu, ap)) return fail("hsa_amd_pointer_info", HSA_STATUS_ERROR);
  uint32_t free_up = 0, free_dn = 0, pref_up = 0, pref_dn = 0;
  if ((st = hsa_amd_memory_copy_engine_status(s->gpu, s->cpu, &free_up)) != HSA_STATUS_SUCCESS && st != HSA_STATUS_ERROR_OUT_OF_RESOURCES)
    return fail("hsa_amd_memory_copy_engine_status", st);
  (void)hsa_amd_memory_copy_engine_status(s->cpu, s->gpu, &free_dn);
  (void)hsa_amd_memory_get_preferred_copy_engine(s->gpu, s->cpu, &pref_up);
  (void)hsa_amd_memory_get_preferred_copy_engine(s->cpu, s->gpu, &pref_dn);
  auto lowest = [](uint32_t m) { return m ? m & (~m + 1u) : 0u; };
  const uint32_t up = lowest(pref_up & free_up ? pref_up & free_up : free_up);
  uint32_t dn = lowest((pref_dn & free_dn & ~up) ? (pref_dn & free_dn & ~up) : (free_dn & ~up));
  if (!up || !dn) return fail("two free copy engines", HSA_STATUS_ERROR_OUT_OF_RESOURCES);
  s->e_up = (hsa_amd_sdma_engine_id_t)up;
  s->e_dn = (hsa_amd_sdma_engine_id_t)dn;
  if (hipMalloc((void **)&s->d_flags, 512) != hipSuccess || hipMemset(s->d_flags, 0, 512) != hipSuccess ||
      hipHostMalloc((void **)&s->h_seq, Sdma::SEQ * sizeof(uint64_t), hipHostMallocDefault) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    (void)hipGetLastError();
    return fail("flag memory", HSA_STATUS_ERROR_OUT_OF_RESOURCES);
  }
  int can = 0;
  if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, b->device) != hipSuccess || !can) return fail("hipStreamWaitValue64", HSA_STATUS_ERROR);
  if (const char *what = sdma_selftest(s)) return fail(what, HSA_STATUS_ERROR);
  s->ok = true;
  return s;
}
void sdma_free(RNNoiseBatch::HostIo &io) {
  if (!io.sdma) return;
  for (hsa_signal_t sg : io.sdma->pool) hsa_signal_destroy(sg);
  if (io.sdma->d_flags) hipFree(io.sdma->d_flags);
  if (io.sdma->h_seq) hipHostFree(io.sdma->h_seq);
  if (io.sdma->hsa_ref) (void)hsa_shut_down();
  delete io.sdma;
  io.sdma = nullptr;
}
}  // namespace

void host_io_release(RNNoiseBatch *b) {
  RNNoiseBatch::HostIo &io = b->io;
  sdma_free(io);
  for (int k = 0; k < 2; k++) {
    if (io.d_in[k]) hipFree(io.d_in[k]);
    if (io.h_in[k]) hipHostFree(io.h_in[k]);
    if (io.up_done[k]) { hipEventDestroy(io.up_done[k]); hipEventDestroy(io.run_done[k]); hipEventDestroy(io.down_done[k]); }
  }
  if (io.up) hipStreamDestroy(io.up);
  if (io.run) hipStreamDestroy(io.run);
  if (io.down) hipStreamDestroy(io.down);
  if (io.ring_mem) hipFree(io.ring_mem);
  for (int k = 0; k < RNNoiseBatch::HostIo::RING; k++)
    for (hipEvent_t e : {io.r_k3[k], io.r_down[k], io.r_up[k], io.r_hp[k]})
      if (e) hipEventDestroy(e);
  io = RNNoiseBatch::HostIo();
}

// device (and, for pageable callers, pinned host) staging for two chunks of `frames` frames each: one allocation per
// side and chunk, carved into in | out | vad | gains
static int host_io_prepare(RNNoiseBatch *b, int frames, size_t esz, bool bounce) {
  RNNoiseBatch::HostIo &io = b->io;
  const size_t N = b->n;
  size_t n_in = ((size_t)frames * N * RN_FRAME_SIZE * esz + 255) / 256 * 64;  // floats' worth of PCM per chunk and direction
  if (io.chunk_frames >= frames && io.pcm_floats >= n_in && (!bounce || io.h_in[0])) return 0;
  const bool had_bounce = io.h_in[0] != nullptr;
  frames = std::max(frames, io.chunk_frames);  // (float and s16 callers alternating on one batch: grow once, to both)
  n_in = std::max(n_in, io.pcm_floats);
  host_io_release(b);
  const size_t fr = (size_t)frames, n_vad = fr * N, n_g = fr * N * RN_NB_BANDS;
  const size_t total = (2 * n_in + n_vad + n_g) * sizeof(float);
  HIP_OK(hipStreamCreateWithFlags(&io.up, hipStreamNonBlocking));
  HIP_OK(hipStreamCreateWithFlags(&io.run, hipStreamNonBlocking));
  HIP_OK(hipStreamCreateWithFlags(&io.down, hipStreamNonBlocking));
  for (int k = 0; k < 2; k++) {
    HIP_OK(hipMalloc((void **)&io.d_in[k], total));
    io.d_out[k] = io.d_in[k] + n_in;
    io.d_vad[k] = io.d_out[k] + n_in;
    io.d_gains[k] = io.d_vad[k] + n_vad;
    if (bounce || had_bounce) {
      HIP_OK(hipHostMalloc((void **)&io.h_in[k], total, hipHostMallocDefault));
      io.h_out[k] = io.h_in[k] + n_in;
      io.h_vad[k] = io.h_out[k] + n_in;
      io.h_gains[k] = io.h_vad[k] + n_vad;
    }
    HIP_OK(hipEventCreateWithFlags(&io.up_done[k], hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&io.run_done[k], hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&io.down_done[k], hipEventDisableTiming));
  }
  io.chunk_frames = frames;
  io.pcm_floats = n_in;
  return 0;
}

// Host-fed path for pinned caller memory (hipHostMalloc / hipHostRegister): the DMA engines read and write it in place, one
// frame per copy, while ONE multi-frame device call runs the kernels as a frame pipeline (the high-pass up to three frames
// ahead on a side stream) over a ring of RING frame slots in HBM.  Per frame f: upload(f) [after high-pass(f - RING) has read the slot] -> high-pass(f) -> ... ->
// network(f), synthesis(f) [after download(f - RING) has drained the slot] -> download(f).  A call pays one frame's upload
// before and one frame's download after its kernels, whatever its length.
static int batch_process_pinned(RNNoiseBatch *b, char *out, const char *in, float *vad, float *gains, int n_frames, bool s16) {
  RNNoiseBatch::HostIo &io = b->io;
  constexpr int RING = RNNoiseBatch::HostIo::RING;
  const size_t N = b->n, esz = s16 ? sizeof(short) : sizeof(float), fsz = N * RN_FRAME_SIZE * esz;
  const size_t slot_pcm = (N * RN_FRAME_SIZE * sizeof(float) + 255) & ~size_t(255), slot_vad = (N * 4 + 255) & ~size_t(255),
               slot_g = (N * RN_NB_BANDS * 4 + 255) & ~size_t(255);
  if (!io.ring_mem || !io.run) {  // streams, ring and events together or not at all: a half-built set is torn down and retried
    auto build = [&]() -> int {
      if (!io.run) {
        HIP_OK(hipStreamCreateWithFlags(&io.run, hipStreamNonBlocking));
        HIP_OK(hipStreamCreateWithFlags(&io.down, hipStreamNonBlocking));
      }
      HIP_OK(hipMalloc((void **)&io.ring_mem, RING * (2 * slot_pcm + slot_vad + slot_g)));
      for (int k = 0; k < RING; k++) {
        HIP_OK(hipEventCreateWithFlags(&io.r_k3[k], hipEventDisableTiming));  // (a copy engine follows a kernel: system scope)
        HIP_OK(hipEventCreateWithFlags(&io.r_down[k], hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&io.r_up[k], hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&io.r_hp[k], hipEventDisableTiming | hipEventDisableSystemFence));  // (read-after-read ordering only)
      }
      return 0;
    };
    if (build()) {
      host_io_release(b);
      return -1;
    }
  }
  // the slots are addressed with the stride of the CALL's frame size (s16 frames use the first half of a slot's room)
  char *r_in = io.ring_mem, *r_out = r_in + RING * slot_pcm;
  float *r_vad = reinterpret_cast<float *>(r_out + RING * slot_pcm), *r_g = reinterpret_cast<float *>(reinterpret_cast<char *>(r_vad) + RING * slot_vad);
  FrameIoHooks hk;
  hk.ring = RING;
  // Copies.  Default: the two named copy engines above (mode sdma).  The runtime's own copies remain as the fallback (no two free
  // engines, no hipStreamWaitValue64, calls longer than the count ring, a batch whose engines once failed) and for A/B runs:
  // int16 frames: uploads and downloads ALTERNATE ON ONE COPY STREAM (io.down), in the order the frame pipeline asks
  // for them: each then finds the DMA engine free.  With two copies in flight the runtime executes one of them as a blit kernel
  // (256 workgroups x 512 lanes) whose PCIe-bound stores stall what runs beside it -- the analysis kernel took 2.3 ms instead
  // of 1.1 (rocprofv3 kernel + memory-copy trace) -- and a copy stream per direction plus the pipeline's three streams is more
  // than the four hardware queues the runtime multiplexes streams onto.  One direction at a time moves an int16 step's
  // 2 x 63 MB in 2.2 ms, just above the kernels' time: 28.0 M frames/s either way (profiles/r4_hostio_modes.txt).
  // float frames (2 x 126 MB per step) are bound by the link whatever the kernels do, and there both directions at once win:
  // uploads ride on the high-pass stream, downloads keep the copy stream -- 19.9 M frames/s against 14.6 M.
  // $RNNOISE_AMD_HOSTIO_COPY = one | hp | sdma forces a mode (A/B runs); profiles/r5_hostio_sdma.txt has the three side by side.
  static const int copy_mode_env = [] {
    const char *e = getenv("RNNOISE_AMD_HOSTIO_COPY");
    return !e ? 0 : (!strcmp(e, "one") ? 1 : (!strcmp(e, "sdma") ? 3 : 2));
  }();
  const bool one_copy_stream = (copy_mode_env == 1 || copy_mode_env == 2) ? copy_mode_env == 1 : s16;
  // (a call's copies are all queued while its kernels are: the count words' source ring must not wrap inside one call)
  Sdma *sd = ((copy_mode_env == 3 || copy_mode_env == 0) && n_frames <= Sdma::SEQ / 2) ? sdma_get(b, in) : nullptr;
  // per call: the signals of its frames (reused from call to call: the previous call ended with every copy complete)
  struct FrameSig { hsa_signal_t hp_read, k3_done, up, up_flag, dn, dn_flag; uint64_t dn_target; };
  std::vector<FrameSig> fs;
  hsa_agent_t a_in{}, a_out{}, a_vad{}, a_g{};
  const void *p_in = nullptr, *p_out = nullptr, *p_vad = nullptr, *p_g = nullptr;
  if (sd) {
    sd->used = 0;
    fs.resize((size_t)n_frames);
    if (!sdma_owner(in, a_in, p_in) || !sdma_owner(out, a_out, p_out) || (vad && !sdma_owner(vad, a_vad, p_vad)) || (gains && !sdma_owner(gains, a_g, p_g))) sd = nullptr;
  }
  auto value_word = [](hsa_signal_t s) { return sdma_value_word(s); };
#define HSA_OK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char *m_ = nullptr; hsa_status_string(s_, &m_); \
    fprintf(stderr, "[rnnoise_amd] %s: %s\n", #x, m_ ? m_ : "?"); return -1; } } while (0)
  hk.before_hp = [&](int f, hipStream_t sc) -> int {
    if (sd) {
      FrameSig &q = fs[(size_t)f];
      q.hp_read = sd->take(1);
      q.k3_done = sd->take(1);
      q.up = sd->take(1);
      q.up_flag = sd->take(1);
      q.dn = sd->take(1 + (vad ? 1 : 0) + (gains ? 1 : 0));
      q.dn_flag = sd->take(1);
      if (!q.hp_read.handle || !q.k3_done.handle || !q.up.handle || !q.up_flag.handle || !q.dn.handle || !q.dn_flag.handle) return -1;
      // upload(f) on the upload engine, once high-pass(f - RING) has read the slot; the count of landed uploads behind it
      const hsa_signal_t *dep = f >= RING ? &fs[(size_t)(f - RING)].hp_read : nullptr;
      HSA_OK(hsa_amd_memory_async_copy_on_engine(r_in + (size_t)(f % RING) * fsz, sd->gpu, static_cast<const char *>(p_in) + (size_t)f * fsz, a_in, fsz,
                                                 dep ? 1 : 0, dep, q.up, sd->e_up, false));
      const uint64_t c = ++sd->up_count;
      sd->h_seq[c % (Sdma::SEQ / 2)] = c;
      HSA_OK(hsa_amd_memory_async_copy_on_engine(&sd->d_flags[0], sd->gpu, &sd->h_seq[c % (Sdma::SEQ / 2)], sd->cpu, 8, 1, &q.up, q.up_flag, sd->e_up, false));
      HIP_OK(hipStreamWaitValue64(sc, &sd->d_flags[0], c, hipStreamWaitValueGte, ~0ull));
      return 0;
    }
    if (!one_copy_stream) {
      HIP_OK(hipMemcpyAsync(r_in + (size_t)(f % RING) * fsz, in + (size_t)f * fsz, fsz, hipMemcpyHostToDevice, sc));
      return 0;
    }
    if (f >= RING) HIP_OK(hipStreamWaitEvent(io.down, io.r_hp[f % RING], 0));   // high-pass(f - RING) has read the slot
    HIP_OK(hipMemcpyAsync(r_in + (size_t)(f % RING) * fsz, in + (size_t)f * fsz, fsz, hipMemcpyHostToDevice, io.down));
    HIP_OK(hipEventRecord(io.r_up[f % RING], io.down));
    HIP_OK(hipStreamWaitEvent(sc, io.r_up[f % RING], 0));
    return 0;
  };
  hk.after_hp = [&](int f, hipStream_t sc) -> int {
    if (sd) return rn_launch_release_store(value_word(fs[(size_t)f].hp_read), 0, sc) == hipSuccess ? 0 : -1;
    if (one_copy_stream) HIP_OK(hipEventRecord(io.r_hp[f % RING], sc));
    return 0;
  };
  hk.before_nn = [&](int f, hipStream_t st) -> int {  // network(f) writes vad / gains, synthesis(f) the PCM of slot f % RING
    if (sd) {
      if (f >= RING) HIP_OK(hipStreamWaitValue64(st, &sd->d_flags[32], fs[(size_t)(f - RING)].dn_target, hipStreamWaitValueGte, ~0ull));
      return 0;
    }
    if (f >= RING) HIP_OK(hipStreamWaitEvent(st, io.r_down[f % RING], 0));
    return 0;
  };
  // Downloads are hipMemcpyAsync (DMA).  $RNNOISE_AMD_D2H=kernel:<workgroups> (A/B runs) replaces them by a small copy kernel
  // writing the caller's pinned memory through its device address (state_kernels.hip: rn_copy_to_host_kernel): measured
  // slower than the serialised DMA copies (18 M against 23-28 M frames/s), kept for the record.
  static const int d2h_blocks = [] {
    const char *e = RN_LAB_ENV("D2H");
    if (e && !strncmp(e, "kernel:", 7)) return std::max(1, atoi(e + 7));
    return 0;
  }();
  auto dev_view = [](void *host) -> void * {  // device address of pinned host memory, or nullptr (then: hipMemcpyAsync)
    void *d = nullptr;
    if (!host || hipHostGetDevicePointer(&d, host, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return d;
  };
  char *out_dev = d2h_blocks ? static_cast<char *>(dev_view(out)) : nullptr;
  float *vad_dev = d2h_blocks && vad ? static_cast<float *>(dev_view(vad)) : nullptr;
  float *gains_dev = d2h_blocks && gains ? static_cast<float *>(dev_view(gains)) : nullptr;
  const bool by_kernel = out_dev && (!vad || vad_dev) && (!gains || gains_dev) && !(reinterpret_cast<uintptr_t>(out_dev) & 15) &&
                         !(reinterpret_cast<uintptr_t>(vad_dev) & 15) && !(reinterpret_cast<uintptr_t>(gains_dev) & 15) && N % 4 == 0;
  hk.after_k3 = [&](int f, hipStream_t st) -> int {
    const int k = f % RING;
    if (sd) {
      FrameSig &q = fs[(size_t)f];
      // download(f) on the download engine, released by a store on the kernels' stream; the count of landed downloads behind it
      HIP_OK(rn_launch_release_store(value_word(q.k3_done), 0, st));
      HSA_OK(hsa_amd_memory_async_copy_on_engine(const_cast<char *>(static_cast<const char *>(p_out)) + (size_t)f * fsz, a_out, r_out + (size_t)k * fsz, sd->gpu, fsz,
                                                 1, &q.k3_done, q.dn, sd->e_dn, false));
      if (vad) HSA_OK(hsa_amd_memory_async_copy_on_engine(const_cast<char *>(static_cast<const char *>(p_vad)) + (size_t)f * N * 4, a_vad, r_vad + (size_t)k * N, sd->gpu,
                                                          N * sizeof(float), 1, &q.k3_done, q.dn, sd->e_dn, false));
      if (gains) HSA_OK(hsa_amd_memory_async_copy_on_engine(const_cast<char *>(static_cast<const char *>(p_g)) + (size_t)f * N * RN_NB_BANDS * 4, a_g,
                                                            r_g + (size_t)k * N * RN_NB_BANDS, sd->gpu, N * RN_NB_BANDS * sizeof(float), 1, &q.k3_done, q.dn, sd->e_dn, false));
      const uint64_t c = ++sd->dn_count;
      sd->h_seq[Sdma::SEQ / 2 + c % (Sdma::SEQ / 2)] = c;
      q.dn_target = c;
      HSA_OK(hsa_amd_memory_async_copy_on_engine(&sd->d_flags[32], sd->gpu, &sd->h_seq[Sdma::SEQ / 2 + c % (Sdma::SEQ / 2)], sd->cpu, 8, 1, &q.dn, q.dn_flag, sd->e_dn, false));
      return 0;
    }
    HIP_OK(hipEventRecord(io.r_k3[k], st));
    HIP_OK(hipStreamWaitEvent(io.down, io.r_k3[k], 0));
    if (by_kernel) {
      HIP_OK(rn_launch_copy_to_host(out_dev + (size_t)f * fsz, r_out + (size_t)k * fsz, fsz, d2h_blocks, io.down));
      if (vad) HIP_OK(rn_launch_copy_to_host(vad_dev + (size_t)f * N, r_vad + (size_t)k * N, N * sizeof(float), 1, io.down));
      if (gains) HIP_OK(rn_launch_copy_to_host(gains_dev + (size_t)f * N * RN_NB_BANDS, r_g + (size_t)k * N * RN_NB_BANDS,
                                               N * RN_NB_BANDS * sizeof(float), std::max(1, d2h_blocks / 4), io.down));
    } else {
      HIP_OK(hipMemcpyAsync(out + (size_t)f * fsz, r_out + (size_t)k * fsz, fsz, hipMemcpyDeviceToHost, io.down));
      if (vad) HIP_OK(hipMemcpyAsync(vad + (size_t)f * N, r_vad + (size_t)k * N, N * sizeof(float), hipMemcpyDeviceToHost, io.down));
      if (gains) HIP_OK(hipMemcpyAsync(gains + (size_t)f * N * RN_NB_BANDS, r_g + (size_t)k * N * RN_NB_BANDS, N * RN_NB_BANDS * sizeof(float),
                                       hipMemcpyDeviceToHost, io.down));
    }
    HIP_OK(hipEventRecord(io.r_down[k], io.down));
    return 0;
  };
  // Stream budget: the runtime multiplexes HIP streams onto four hardware queues, one of which belongs to the application's
  // own stream.  The three-stream frame pipeline plus the copy stream would be four more, and the high-pass and analysis
  // streams then share a queue: what the high-pass stream carries queues up behind analysis kernels and the whole step
  // serialises (rocprofv3 trace: 3.5 ms per 65,536-stream s16 step).  Analysis therefore stays on the main stream here
  // (schedule 1: only the high-pass runs ahead on a side stream), which costs the 2-3 % the analysis overlap is worth.
  static const int sched_env = [] { const char *e = RN_LAB_ENV("HOSTIO_SCHEDULE"); return e ? atoi(e) : 1; }();  // (A/B runs)
  const int keep = b->schedule;
  if (b->schedule == 0) b->schedule = sched_env;
  const int rc = batch_process_device_impl(b, r_out, r_in, r_vad, r_g, n_frames, io.run, s16, &hk);
  b->schedule = keep;
  if (sd && !rc && n_frames > 0) {
    // every copy of the call is complete once the last frame's download count has landed (the engines work in order)
    const hsa_signal_t last = fs[(size_t)n_frames - 1].dn_flag;
    if (!sdma_wait(last, 30.0)) {
      // a copy that never completes leaves streams parked on the count words for ever: release them by hand (both counts past
      // anything this call waits for), give the mode up for this batch, drain, reset
      fprintf(stderr, "[rnnoise_amd] rnnoise_batch_process: the copy engines did not finish within 30 s; falling back to the runtime's copies\n");
      for (FrameSig &q : fs) {
        if (q.hp_read.handle) hsa_signal_store_screlease(q.hp_read, 0);
        if (q.k3_done.handle) hsa_signal_store_screlease(q.k3_done, 0);
      }
      const uint64_t past[2] = {~0ull >> 1, ~0ull >> 1};
      (void)hipMemcpy(&sd->d_flags[0], &past[0], 8, hipMemcpyHostToDevice);
      (void)hipMemcpy(&sd->d_flags[32], &past[1], 8, hipMemcpyHostToDevice);
      sd->ok = false;
      (void)hipDeviceSynchronize();
      (void)rnnoise_batch_reset(b);
      return -1;
    }
  }
  if (rc) {
    // whatever was queued must not outlive the caller's buffers; and some of the call's frames may have run while the batch's
    // frame bookkeeping was not advanced: the streams are no longer in a state any caller knows -- back to the initial one
    if (sd) {  // (copies parked on signals that no kernel will release any more: release them by hand, then drain)
      for (FrameSig &q : fs) {
        if (q.hp_read.handle) hsa_signal_store_screlease(q.hp_read, 0);
        if (q.k3_done.handle) hsa_signal_store_screlease(q.k3_done, 0);
      }
      for (FrameSig &q : fs)  // (in order on one engine: the first download that does not land in 2 s ends the waiting)
        if (q.dn_flag.handle && q.dn_target && !sdma_wait(q.dn_flag, 2.0)) break;
      // (and streams parked on a count that a copy which was never queued would have written)
      const uint64_t past = ~0ull >> 1;
      (void)hipMemcpy(&sd->d_flags[0], &past, 8, hipMemcpyHostToDevice);
      (void)hipMemcpy(&sd->d_flags[32], &past, 8, hipMemcpyHostToDevice);
      sd->ok = false;  // (the counts no longer mean anything: this batch goes on with the runtime's copies)
    }
    (void)hipDeviceSynchronize();
    fprintf(stderr, "[rnnoise_amd] rnnoise_batch_process: a GPU step failed inside the call; the batch has been reset\n");
    (void)rnnoise_batch_reset(b);
    return -1;
  }
  HIP_OK(hipStreamSynchronize(io.down));
  HIP_OK(hipStreamSynchronize(io.run));  // (the side streams of the pipelined schedule join `run` before its last kernel)
  return 0;
#undef HSA_OK
}

static int batch_process_host_impl(RNNoiseBatch *b, void *out_v, const void *in_v, float *vad, float *gains, int n_frames,
                                   bool s16) {
  if (!b || !out_v || !in_v || n_frames < 0) return -1;
  if (n_frames == 0) return 0;
  ON_DEVICE(b->device);
  const size_t esz = s16 ? sizeof(short) : sizeof(float);
  const size_t N = b->n, fsz = N * RN_FRAME_SIZE * esz;  // bytes of PCM per frame step
  const char *in = static_cast<const char *>(in_v);
  char *out = static_cast<char *>(out_v);
  const bool direct = host_pinned(in) && host_pinned(out) && (!vad || host_pinned(vad)) && (!gains || host_pinned(gains));
  if (direct) return batch_process_pinned(b, out, in, vad, gains, n_frames, s16);
  // Pageable memory goes through pinned bounce buffers (the copy in and out of them is the calling thread's work): chunks of
  // about 32 MB of PCM each way (at least one frame), two in flight.
  const int chunk = (int)std::min<size_t>((size_t)n_frames, std::max<size_t>(1, ((size_t)32 << 20) / fsz));
  if (host_io_prepare(b, chunk, esz, !direct)) return -1;
  RNNoiseBatch::HostIo &io = b->io;
  const int n_chunks = (n_frames + chunk - 1) / chunk;
  auto frames_of = [&](int c) { return std::min(chunk, n_frames - c * chunk); };
  auto upload = [&](int c) -> int {  // chunk c -> staging set c & 1 (free once chunk c-2 has been downloaded)
    const int k = c & 1, f = frames_of(c);
    const void *src = in + (size_t)c * chunk * fsz;
    if (c >= 2) HIP_OK(hipStreamWaitEvent(io.up, io.run_done[k], 0));  // its kernels no longer read d_in[k]
    if (!direct) {
      if (c >= 2) HIP_OK(hipEventSynchronize(io.up_done[k]));          // the bounce buffer has been sent
      memcpy(io.h_in[k], src, (size_t)f * fsz);
      src = io.h_in[k];
    }
    HIP_OK(hipMemcpyAsync(io.d_in[k], src, (size_t)f * fsz, hipMemcpyHostToDevice, io.up));
    HIP_OK(hipEventRecord(io.up_done[k], io.up));
    return 0;
  };
  auto collect = [&](int c) -> int {  // pageable callers: bounce buffer of chunk c -> caller memory
    const int k = c & 1, f = frames_of(c);
    HIP_OK(hipEventSynchronize(io.down_done[k]));
    memcpy(out + (size_t)c * chunk * fsz, io.h_out[k], (size_t)f * fsz);
    if (vad) memcpy(vad + (size_t)c * chunk * N, io.h_vad[k], (size_t)f * N * sizeof(float));
    if (gains) memcpy(gains + (size_t)c * chunk * N * RN_NB_BANDS, io.h_gains[k], (size_t)f * N * RN_NB_BANDS * sizeof(float));
    return 0;
  };
  // the chunk loop as one unit: any failure inside it leaves copies and kernels in flight on three streams and the batch's frame
  // bookkeeping out of step with what the GPU ran -- drain everything and put the batch back to its initial state, like the
  // pinned path does
  auto chunks = [&]() -> int {
    if (upload(0)) return -1;
    for (int c = 0; c < n_chunks; c++) {
      const int k = c & 1, f = frames_of(c);
      if (c + 1 < n_chunks && upload(c + 1)) return -1;
      HIP_OK(hipStreamWaitEvent(io.run, io.up_done[k], 0));
      if (c >= 2) HIP_OK(hipStreamWaitEvent(io.run, io.down_done[k], 0));  // d_out[k] of chunk c-2 has left
      if (batch_process_device_impl(b, io.d_out[k], io.d_in[k], io.d_vad[k], io.d_gains[k], f, io.run, s16)) return -1;
      HIP_OK(hipEventRecord(io.run_done[k], io.run));
      if (!direct && c >= 2 && collect(c - 2)) return -1;  // frees h_out[k] for the download queued below
      HIP_OK(hipStreamWaitEvent(io.down, io.run_done[k], 0));
      void *dst_out = direct ? static_cast<void *>(out + (size_t)c * chunk * fsz) : io.h_out[k];
      HIP_OK(hipMemcpyAsync(dst_out, io.d_out[k], (size_t)f * fsz, hipMemcpyDeviceToHost, io.down));
      if (vad) HIP_OK(hipMemcpyAsync(direct ? vad + (size_t)c * chunk * N : io.h_vad[k], io.d_vad[k], (size_t)f * N * sizeof(float),
                                     hipMemcpyDeviceToHost, io.down));
      if (gains) HIP_OK(hipMemcpyAsync(direct ? gains + (size_t)c * chunk * N * RN_NB_BANDS : io.h_gains[k], io.d_gains[k],
                                       (size_t)f * N * RN_NB_BANDS * sizeof(float), hipMemcpyDeviceToHost, io.down));
      HIP_OK(hipEventRecord(io.down_done[k], io.down));
    }
    if (!direct)
      for (int c = std::max(0, n_chunks - 2); c < n_chunks; c++)
        if (collect(c)) return -1;
    HIP_OK(hipStreamSynchronize(io.down));
    HIP_OK(hipStreamSynchronize(io.run));  // (the side streams of the pipelined schedule join `run` before its last kernel)
    return 0;
  };
  if (chunks()) {
    (void)hipDeviceSynchronize();
    (void)hipGetLastError();
    fprintf(stderr, "[rnnoise_amd] rnnoise_batch_process: a GPU step or copy failed inside the call; the batch has been reset\n");
    (void)rnnoise_batch_reset(b);
    return -1;
  }
  return 0;
}

extern "C" int rnnoise_batch_process(RNNoiseBatch *b, float *out, const float *in, float *vad, float *gains,
                                     int n_frames) {
  return batch_process_host_impl(b, out, in, vad, gains, n_frames, false);
}

extern "C" int rnnoise_batch_process_s16(RNNoiseBatch *b, short *out, const short *in, float *vad, float *gains,
                                         int n_frames) {
  return batch_process_host_impl(b, out, in, vad, gains, n_frames, true);
}

