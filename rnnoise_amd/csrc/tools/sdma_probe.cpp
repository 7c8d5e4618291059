// sdma_probe.cpp -- can the two PCIe directions of the host-fed path run on DMA engines AT ONCE?
//
// The HIP runtime picks the copy engine: with a host->device and a device->host hipMemcpyAsync in flight on two streams it runs one
// of them as a blit KERNEL (256 workgroups whose PCIe-bound stores stall whatever computes beside them: profiles/r3_hostio_traces.txt),
// which is why the host-fed int16 path alternates the directions on ONE copy stream and is copy-bound at 28-29 M frames/s.
// Underneath HIP, ROCr has hsa_amd_memory_async_copy_on_engine(): a copy on a NAMED SDMA engine.  This probe measures, on
// pinned host memory <-> HBM, 63 MB per copy (one 65,536-stream int16 frame):
//   A  hipMemcpyAsync, one direction at a time           B  hipMemcpyAsync, both directions at once on two streams
//   C  HSA copies on the engines the runtime recommends, one direction at a time
//   D  HSA copies, H2D on one engine and D2H on another, both at once
//   E  D while a compute kernel streams HBM on the GPU (does the copy still take bandwidth from kernels, as the blit kernel did?)
// Build: hipcc --offload-arch=gfx950 -O2 -o build/sdma_probe tools/sdma_probe.cpp -lhsa-runtime64
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define HIPC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
static const char *hsa_str(hsa_status_t s) { const char *m = nullptr; hsa_status_string(s, &m); return m ? m : "?"; }
#define HSAC(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { printf("HSA call failed: %s -> %s (0x%x)\n", #x, hsa_str(s_), (unsigned)s_); return 1; } } while (0)

static hsa_agent_t g_gpu{}, g_cpu{};
static bool have_gpu = false, have_cpu = false;
static hsa_status_t on_agent(hsa_agent_t a, void *) {
  hsa_device_type_t t;
  hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
  if (t == HSA_DEVICE_TYPE_GPU && !have_gpu) { g_gpu = a; have_gpu = true; }
  if (t == HSA_DEVICE_TYPE_CPU && !have_cpu) { g_cpu = a; have_cpu = true; }
  return HSA_STATUS_SUCCESS;
}
__global__ void stream_kernel(float4 *p, size_t n, int reps) {
  for (int r = 0; r < reps; r++)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
      float4 v = p[i];
      v.x += 1.f;
      p[i] = v;
    }
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  const size_t B = (size_t)65536 * 480 * 2;  // one int16 frame of 65,536 streams
  const int R = 20;
  HIPC(hipSetDevice(0));
  char *h_up, *h_dn, *d_up, *d_dn;
  HIPC(hipHostMalloc((void **)&h_up, B, hipHostMallocDefault));
  HIPC(hipHostMalloc((void **)&h_dn, B, hipHostMallocDefault));
  HIPC(hipMalloc((void **)&d_up, B));
  HIPC(hipMalloc((void **)&d_dn, B));
  float4 *d_big;
  const size_t NB = (size_t)1 << 28;  // 4 GB of float4 traffic per rep / 16
  HIPC(hipMalloc((void **)&d_big, NB * sizeof(float4) / 16));
  hipStream_t s1, s2, s3;
  HIPC(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  HIPC(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  HIPC(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
  for (int i = 0; i < 3; i++) { HIPC(hipMemcpyAsync(d_up, h_up, B, hipMemcpyHostToDevice, s1)); HIPC(hipMemcpyAsync(h_dn, d_dn, B, hipMemcpyDeviceToHost, s2)); }
  HIPC(hipDeviceSynchronize());
  double t0 = now();
  for (int i = 0; i < R; i++) HIPC(hipMemcpyAsync(d_up, h_up, B, hipMemcpyHostToDevice, s1));
  HIPC(hipDeviceSynchronize());
  double a_up = B * (double)R / (now() - t0) / 1e9;
  t0 = now();
  for (int i = 0; i < R; i++) HIPC(hipMemcpyAsync(h_dn, d_dn, B, hipMemcpyDeviceToHost, s2));
  HIPC(hipDeviceSynchronize());
  double a_dn = B * (double)R / (now() - t0) / 1e9;
  printf("A  hipMemcpyAsync, one direction at a time:   H2D %.1f GB/s   D2H %.1f GB/s\n", a_up, a_dn);
  t0 = now();
  for (int i = 0; i < R; i++) { HIPC(hipMemcpyAsync(d_up, h_up, B, hipMemcpyHostToDevice, s1)); HIPC(hipMemcpyAsync(h_dn, d_dn, B, hipMemcpyDeviceToHost, s2)); }
  HIPC(hipDeviceSynchronize());
  printf("B  hipMemcpyAsync, both at once (2 streams):  %.1f GB/s per direction\n", B * (double)R / (now() - t0) / 1e9);

  // ---- underneath: ROCr ----
  HSAC(hsa_init());  // (reference-counted: HIP has initialised it already)
  HSAC(hsa_iterate_agents(on_agent, nullptr));
  if (!have_gpu || !have_cpu) { printf("no GPU / CPU agent\n"); return 1; }
  uint32_t m_up = 0, m_dn = 0, r_up = 0, r_dn = 0;
  hsa_status_t st = hsa_amd_memory_copy_engine_status(g_gpu, g_cpu, &m_up);
  printf("hsa_amd_memory_copy_engine_status(dst GPU, src CPU): %s, free engines mask 0x%x\n", hsa_str(st), m_up);
  st = hsa_amd_memory_copy_engine_status(g_cpu, g_gpu, &m_dn);
  printf("hsa_amd_memory_copy_engine_status(dst CPU, src GPU): %s, free engines mask 0x%x\n", hsa_str(st), m_dn);
  st = hsa_amd_memory_get_preferred_copy_engine(g_gpu, g_cpu, &r_up);
  printf("preferred engines H2D: %s mask 0x%x;", hsa_str(st), r_up);
  st = hsa_amd_memory_get_preferred_copy_engine(g_cpu, g_gpu, &r_dn);
  printf(" D2H: %s mask 0x%x\n", hsa_str(st), r_dn);
  // the buffers are HIP allocations: ROCr knows them (HIP allocates through ROCr); pinned host memory is accessible to the GPU agent
  auto pick = [](uint32_t mask, int skip) { for (int b = 0; b < 16; b++) if (mask >> b & 1) { if (!skip--) return 1u << b; } return 0u; };
  const uint32_t e_up = pick(r_up ? r_up : m_up, 0);
  uint32_t e_dn = pick(r_dn ? r_dn : m_dn, 0);
  if (e_dn == e_up) e_dn = pick((r_dn ? r_dn : m_dn) & ~e_up, 0) ? pick((r_dn ? r_dn : m_dn) & ~e_up, 0) : pick(m_dn & ~e_up, 0);
  printf("using engine mask 0x%x for H2D and 0x%x for D2H\n", e_up, e_dn);
  if (!e_up || !e_dn) { printf("no two distinct engines available\n"); return 1; }
  hsa_signal_t su, sd;
  HSAC(hsa_signal_create(1, 0, nullptr, &su));
  HSAC(hsa_signal_create(1, 0, nullptr, &sd));
  auto copy_up = [&]() { hsa_signal_store_relaxed(su, 1); return hsa_amd_memory_async_copy_on_engine(d_up, g_gpu, h_up, g_cpu, B, 0, nullptr, su, (hsa_amd_sdma_engine_id_t)e_up, false); };
  auto copy_dn = [&]() { hsa_signal_store_relaxed(sd, 1); return hsa_amd_memory_async_copy_on_engine(h_dn, g_cpu, d_dn, g_gpu, B, 0, nullptr, sd, (hsa_amd_sdma_engine_id_t)e_dn, false); };
  auto wait = [&](hsa_signal_t s) { while (hsa_signal_wait_scacquire(s, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED) >= 1) {} };
  HSAC(copy_up()); wait(su);
  HSAC(copy_dn()); wait(sd);
  t0 = now();
  for (int i = 0; i < R; i++) { HSAC(copy_up()); wait(su); }
  double c_up = B * (double)R / (now() - t0) / 1e9;
  t0 = now();
  for (int i = 0; i < R; i++) { HSAC(copy_dn()); wait(sd); }
  double c_dn = B * (double)R / (now() - t0) / 1e9;
  printf("C  HSA copies on named engines, one direction at a time:   H2D %.1f GB/s   D2H %.1f GB/s\n", c_up, c_dn);
  t0 = now();
  for (int i = 0; i < R; i++) { HSAC(copy_up()); HSAC(copy_dn()); wait(su); wait(sd); }
  printf("D  HSA copies, both directions at once on two engines:     %.1f GB/s per direction\n", B * (double)R / (now() - t0) / 1e9);
  // E: beside a kernel that streams HBM
  const size_t n4 = NB / 16;
  hipLaunchKernelGGL(stream_kernel, dim3(2048), dim3(256), 0, s3, d_big, n4, 1);
  HIPC(hipDeviceSynchronize());
  hipEvent_t k0, k1;
  HIPC(hipEventCreate(&k0)); HIPC(hipEventCreate(&k1));
  HIPC(hipEventRecord(k0, s3));
  hipLaunchKernelGGL(stream_kernel, dim3(2048), dim3(256), 0, s3, d_big, n4, 40);
  HIPC(hipEventRecord(k1, s3));
  HIPC(hipDeviceSynchronize());
  float alone_ms = 0;
  HIPC(hipEventElapsedTime(&alone_ms, k0, k1));
  HIPC(hipEventRecord(k0, s3));
  hipLaunchKernelGGL(stream_kernel, dim3(2048), dim3(256), 0, s3, d_big, n4, 40);
  HIPC(hipEventRecord(k1, s3));
  t0 = now();
  int done = 0;
  while (hipEventQuery(k1) == hipErrorNotReady) { HSAC(copy_up()); HSAC(copy_dn()); wait(su); wait(sd); done++; }
  double dt = now() - t0;
  HIPC(hipDeviceSynchronize());
  float with_ms = 0;
  HIPC(hipEventElapsedTime(&with_ms, k0, k1));
  printf("E  a kernel streaming %.1f GB of HBM: %.2f ms alone, %.2f ms with both HSA copies running beside it (%.1f GB/s per direction meanwhile)\n",
         40 * 2.0 * n4 * 16 / 1e9, alone_ms, with_ms, B * (double)done / dt / 1e9);
  // the same with HIP's two-stream copies (one of them a blit kernel)
  HIPC(hipEventRecord(k0, s3));
  hipLaunchKernelGGL(stream_kernel, dim3(2048), dim3(256), 0, s3, d_big, n4, 40);
  HIPC(hipEventRecord(k1, s3));
  t0 = now();
  done = 0;
  while (hipEventQuery(k1) == hipErrorNotReady) {
    HIPC(hipMemcpyAsync(d_up, h_up, B, hipMemcpyHostToDevice, s1));
    HIPC(hipMemcpyAsync(h_dn, d_dn, B, hipMemcpyDeviceToHost, s2));
    HIPC(hipStreamSynchronize(s1));
    HIPC(hipStreamSynchronize(s2));
    done++;
  }
  dt = now() - t0;
  HIPC(hipDeviceSynchronize());
  HIPC(hipEventElapsedTime(&with_ms, k0, k1));
  printf("E' the same kernel with hipMemcpyAsync on two streams beside it: %.2f ms (%.1f GB/s per direction meanwhile)\n", with_ms, B * (double)done / dt / 1e9);
  return 0;
}
