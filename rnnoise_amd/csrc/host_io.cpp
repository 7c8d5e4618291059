// host_io.cpp -- host-fed calls: pinned caller memory through a frame-granular ring beside one pipelined
// device call, pageable memory through double-buffered bounce chunks.
#include "shim.h"

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

void host_io_release(RNNoiseBatch *b) {
  RNNoiseBatch::HostIo &io = b->io;
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
  // Copies.  int16 frames: uploads and downloads ALTERNATE ON ONE COPY STREAM (io.down), in the order the frame pipeline asks
  // for them: each then finds the DMA engine free.  With two copies in flight the runtime executes one of them as a blit kernel
  // (256 workgroups x 512 lanes) whose PCIe-bound stores stall what runs beside it -- the analysis kernel took 2.3 ms instead
  // of 1.1 (rocprofv3 kernel + memory-copy trace) -- and a copy stream per direction plus the pipeline's three streams is more
  // than the four hardware queues the runtime multiplexes streams onto.  One direction at a time moves an int16 step's
  // 2 x 63 MB in 2.2 ms, just above the kernels' time: 28.0 M frames/s either way (profiles/r4_hostio_modes.txt).
  // float frames (2 x 126 MB per step) are bound by the link whatever the kernels do, and there both directions at once win:
  // uploads ride on the high-pass stream, downloads keep the copy stream -- 19.9 M frames/s against 14.6 M.
  // $RNNOISE_AMD_HOSTIO_COPY = one | hp forces a mode (A/B runs).
  static const int copy_mode_env = [] { const char *e = getenv("RNNOISE_AMD_HOSTIO_COPY"); return !e ? 0 : (!strcmp(e, "one") ? 1 : 2); }();
  const bool one_copy_stream = copy_mode_env ? copy_mode_env == 1 : s16;
  hk.before_hp = [&](int f, hipStream_t sc) -> int {
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
    if (one_copy_stream) HIP_OK(hipEventRecord(io.r_hp[f % RING], sc));
    return 0;
  };
  hk.before_nn = [&](int f, hipStream_t st) -> int {  // network(f) writes vad / gains, synthesis(f) the PCM of slot f % RING
    if (f >= RING) HIP_OK(hipStreamWaitEvent(st, io.r_down[f % RING], 0));
    return 0;
  };
  // Downloads are hipMemcpyAsync (DMA).  $RNNOISE_AMD_D2H=kernel:<workgroups> (A/B runs) replaces them by a small copy kernel
  // writing the caller's pinned memory through its device address (state_kernels.hip: rn_copy_to_host_kernel): measured
  // slower than the serialised DMA copies (18 M against 23-28 M frames/s), kept for the record.
  static const int d2h_blocks = [] {
    const char *e = getenv("RNNOISE_AMD_D2H");
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
  static const int sched_env = [] { const char *e = getenv("RNNOISE_AMD_HOSTIO_SCHEDULE"); return e ? atoi(e) : 1; }();  // (A/B runs)
  const int keep = b->schedule;
  if (b->schedule == 0) b->schedule = sched_env;
  const int rc = batch_process_device_impl(b, r_out, r_in, r_vad, r_g, n_frames, io.run, s16, &hk);
  b->schedule = keep;
  if (rc) {
    // whatever was queued must not outlive the caller's buffers; and some of the call's frames may have run while the batch's
    // frame bookkeeping was not advanced: the streams are no longer in a state any caller knows -- back to the initial one
    (void)hipDeviceSynchronize();
    fprintf(stderr, "[rnnoise_amd] rnnoise_batch_process: a GPU step failed inside the call; the batch has been reset\n");
    (void)rnnoise_batch_reset(b);
    return -1;
  }
  HIP_OK(hipStreamSynchronize(io.down));
  HIP_OK(hipStreamSynchronize(io.run));  // (the side streams of the pipelined schedule join `run` before its last kernel)
  return 0;
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

