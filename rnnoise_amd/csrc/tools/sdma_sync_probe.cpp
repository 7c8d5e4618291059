// sdma_sync_probe.cpp -- the two cross-dependencies the host-fed path needs between HIP streams and explicit SDMA-engine copies:
//   T1  SDMA -> stream: a HIP stream waits (hipStreamWaitValue64) on a word of HIP "signal memory" that a small SDMA copy, queued
//       behind the payload copy on the same engine, writes;
//   T2  stream -> SDMA: a one-thread kernel on the stream stores 0 into the VALUE WORD of an HSA signal (amd_signal_t::value) that
//       a later SDMA copy lists as its dependency.
// Build: hipcc --offload-arch=gfx950 -O2 -o build/sdma_sync_probe tools/sdma_sync_probe.cpp -lhsa-runtime64
#include <hip/hip_runtime.h>
#include <hsa/amd_hsa_signal.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>

#define HIPC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error: %s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static const char *hsa_str(hsa_status_t s) { const char *m = nullptr; hsa_status_string(s, &m); return m ? m : "?"; }
#define HSAC(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { printf("HSA call failed: %s -> %s\n", #x, hsa_str(s_)); return 1; } } while (0)

__global__ void store64(volatile int64_t *p, int64_t v) { __hip_atomic_store(const_cast<int64_t *>(p), v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
__global__ void fill(unsigned *p, unsigned v, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v; }
__global__ void check(const unsigned *p, unsigned v, size_t n, unsigned *bad) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) if (p[i] != v) atomicAdd(bad, 1u); }
static hsa_agent_t owner_of(const void *p) {
  hsa_amd_pointer_info_t info;
  memset(&info, 0, sizeof info);
  info.size = sizeof info;
  hsa_amd_pointer_info(p, &info, nullptr, nullptr, nullptr);
  return info.agentOwner;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const size_t B = (size_t)64 << 20, NW = B / 4;
  HIPC(hipSetDevice(0));
  int can = 0;
  HIPC(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  unsigned *h_src, *h_dst, *d_buf, *d_bad;
  HIPC(hipHostMalloc((void **)&h_src, B, hipHostMallocDefault));
  HIPC(hipHostMalloc((void **)&h_dst, B, hipHostMallocDefault));
  HIPC(hipMalloc((void **)&d_buf, B));
  HIPC(hipMalloc((void **)&d_bad, 4));
  HIPC(hipMemset(d_bad, 0, 4));
  uint64_t *flag = nullptr, *h_seq = nullptr;
  HIPC(hipExtMallocWithFlags((void **)&flag, 8, hipMallocSignalMemory));
  HIPC(hipHostMalloc((void **)&h_seq, 4096 * 8, hipHostMallocDefault));
  *flag = 0;
  hipStream_t st;
  HIPC(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  HSAC(hsa_init());
  const hsa_agent_t gpu = owner_of(d_buf), cpu = owner_of(h_src), flag_owner = owner_of(flag);
  hsa_device_type_t tg, tc, tf;
  hsa_agent_get_info(gpu, HSA_AGENT_INFO_DEVICE, &tg);
  hsa_agent_get_info(cpu, HSA_AGENT_INFO_DEVICE, &tc);
  hsa_agent_get_info(flag_owner, HSA_AGENT_INFO_DEVICE, &tf);
  printf("owner of hipMalloc memory: %s agent; of hipHostMalloc memory: %s agent; of hipMallocSignalMemory: %s agent\n",
         tg == HSA_DEVICE_TYPE_GPU ? "GPU" : "CPU", tc == HSA_DEVICE_TYPE_GPU ? "GPU" : "CPU", tf == HSA_DEVICE_TYPE_GPU ? "GPU" : "CPU");
  hsa_signal_t c1, c2, g1, c3;
  HSAC(hsa_signal_create(1, 0, nullptr, &c1));
  HSAC(hsa_signal_create(1, 0, nullptr, &c2));
  HSAC(hsa_signal_create(1, 0, nullptr, &g1));
  HSAC(hsa_signal_create(1, 0, nullptr, &c3));
  const hsa_amd_sdma_engine_id_t E_UP = HSA_AMD_SDMA_ENGINE_0, E_DN = HSA_AMD_SDMA_ENGINE_1;

  // ---- T1: upload by SDMA, flag by SDMA behind it, stream waits for the flag, kernel checks the payload ----
  // variant 0: the flag is a word of DEVICE memory (hipMalloc); variant 1: HIP signal memory, the copy declared host -> GPU agent
  uint64_t *d_flag = nullptr;
  HIPC(hipMalloc((void **)&d_flag, 256));
  HIPC(hipMemset(d_flag, 0, 256));
  for (int variant = 0; variant < 2; variant++) {
    uint64_t *fw = variant == 0 ? d_flag : flag;
    const unsigned pat = 0xabcd0001u + variant;
    for (size_t i = 0; i < NW; i++) h_src[i] = pat;
    HIPC(hipMemset(d_bad, 0, 4));
    hipError_t we = hipStreamWaitValue64(st, fw, 1, hipStreamWaitValueGte, ~0ull);
    printf("T1.%d: hipStreamWaitValue64 on %s -> %s\n", variant, variant == 0 ? "device memory" : "signal memory", hipGetErrorString(we));
    if (we != hipSuccess) { (void)hipGetLastError(); continue; }
    hipLaunchKernelGGL(check, dim3(1024), dim3(256), 0, st, d_buf, pat, NW, d_bad);
    std::this_thread::sleep_for(std::chrono::milliseconds(20));
    printf("T1.%d: stream parked on the flag: hipStreamQuery -> %s\n", variant, hipStreamQuery(st) == hipErrorNotReady ? "not ready (as it should be)" : "READY?!");
    h_seq[1] = 1;
    hsa_signal_store_relaxed(c1, 1);
    hsa_signal_store_relaxed(c2, 1);
    hsa_status_t s1 = hsa_amd_memory_async_copy_on_engine(d_buf, gpu, h_src, cpu, B, 0, nullptr, c1, E_UP, false);
    hsa_status_t s2 = hsa_amd_memory_async_copy_on_engine(fw, gpu, &h_seq[1], cpu, 8, 1, &c1, c2, E_UP, false);
    printf("T1.%d: payload copy -> %s; flag copy -> %s\n", variant, hsa_str(s1), hsa_str(s2));
    const double t1 = now();
    while (hipStreamQuery(st) == hipErrorNotReady && now() - t1 < 3) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    printf("T1.%d: payload completion %ld, flag completion %ld, stream %s\n", variant, (long)hsa_signal_load_relaxed(c1), (long)hsa_signal_load_relaxed(c2),
           hipStreamQuery(st) == hipErrorNotReady ? "STILL PARKED" : "released");
    if (hipStreamQuery(st) == hipErrorNotReady) {  // release it by hand so that the probe can go on
      const uint64_t one = 1;
      if (variant == 0) (void)hipMemcpy(fw, &one, 8, hipMemcpyHostToDevice);
      else *reinterpret_cast<volatile uint64_t *>(fw) = 1;
      std::this_thread::sleep_for(std::chrono::milliseconds(100));
      printf("T1.%d: after a host write of 1 to the flag: stream %s\n", variant, hipStreamQuery(st) == hipErrorNotReady ? "STILL PARKED" : "released");
      if (hipStreamQuery(st) == hipErrorNotReady) return 1;
    }
    HIPC(hipStreamSynchronize(st));
    unsigned bad = 1;
    HIPC(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
    printf("T1.%d: SDMA upload -> SDMA flag -> hipStreamWaitValue64 -> kernel: %u wrong words of %zu  => %s\n", variant, bad, NW, bad ? "FAILED" : "works");
  }

  // ---- T2: kernel produces, a one-thread kernel releases an HSA signal, the SDMA download depends on it ----
  amd_signal_t *gs = reinterpret_cast<amd_signal_t *>(g1.handle);
  memset(h_dst, 0, B);
  HSAC(hsa_amd_memory_async_copy_on_engine(h_dst, cpu, d_buf, gpu, B, 1, &g1, c3, E_DN, false));  // queued FIRST: must wait for g1
  std::this_thread::sleep_for(std::chrono::milliseconds(20));
  printf("T2: download queued behind the signal: completion signal = %ld (1 = still waiting)\n", (long)hsa_signal_load_relaxed(c3));
  hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, st, d_buf, 0x5eed0002u, NW);
  hipLaunchKernelGGL(store64, dim3(1), dim3(1), 0, st, &gs->value, (int64_t)0);
  const double t0 = now();
  while (hsa_signal_wait_scacquire(c3, HSA_SIGNAL_CONDITION_LT, 1, 2000000000ull, HSA_WAIT_STATE_BLOCKED) >= 1 && now() - t0 < 5) {}
  size_t wrong = 0;
  for (size_t i = 0; i < NW; i++) wrong += h_dst[i] != 0x5eed0002u;
  printf("T2: kernel -> store to amd_signal_t::value -> SDMA download: completion %ld, %zu wrong words of %zu  => %s\n", (long)hsa_signal_load_relaxed(c3), wrong, NW,
         (wrong || hsa_signal_load_relaxed(c3) >= 1) ? "FAILED" : "works");
  return 0;
}
