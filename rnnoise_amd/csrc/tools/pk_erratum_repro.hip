// pk_erratum_repro.hip -- stand-alone reproducer (no dependencies beyond the HIP runtime), gfx950 / MI355X, ROCm 7.x:
//   a v_pk_mul_f32 whose op_sel makes the LOW result take the HIGH register of a source pair returns a wrong product in lanes
//   48..63 of a wavefront while ANOTHER wavefront of the same SIMD issues v_mfma_i32_16x16x64_i8 (128-bit A/B operands).
// Every lane of a victim wave runs the same chain on the same operands, so all 64 lanes must end with lane 0's bits; the control
// (the same chain with the operand in the LOW register and no op_sel) never differs.
//   hipcc --offload-arch=gfx950 -O2 pk_erratum_repro.hip -o pk_erratum_repro && ./pk_erratum_repro
// Expected on correct hardware:   control: 0 / N wrong   op_sel:[0,1]: 0 / N wrong
// Observed (profiles/r6_pk_sweep.txt):  control: 0 / N   op_sel:[0,1]: thousands of waves wrong, wrong lanes only in 48..63
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));

__device__ float lcg(unsigned &s) {  // pseudo-random operand in [-1, 1), the same in every lane
  s = s * 1664525u + 1013904223u;
  return (float)(int)(s >> 9) * (1.f / 4194304.f) - 1.f;
}
// 8 waves per workgroup = 2 per SIMD: waves 0-3 issue MFMAs, waves 4-7 run the packed-math chain
extern "C" __global__ void __launch_bounds__(512) repro(int use_op_sel, int iters, int nops, unsigned *out, int *sink) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave < 4) {  // aggressor: twelve independent int8 MFMAs per iteration
    v4i acc[12];
    for (int k = 0; k < 12; k++) acc[k] = v4i{lane, k, 0, 0};
    const v4i a = {lane * 3, 7, lane, 1}, b = {lane, 5, 9, lane * 7};
    for (int i = 0; i < iters / 2; i++)
#pragma unroll
      for (int k = 0; k < 12; k++) acc[k] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[k], 0, 0, 0);
    int x = 0;
    for (int k = 0; k < 12; k++) x ^= acc[k][0] ^ acc[k][1] ^ acc[k][2] ^ acc[k][3];
    if (x == 0x12345678) *sink = x;
    return;
  }
  v2f acc = {0.f, 0.f}, w = {0.f, 0.f}, p;
  unsigned s = 12345u;
  for (int i = 0; i < iters; i++) {
    for (int d = 0; d < nops; d++) asm volatile("s_nop 0");  // (moves the victim's issue cadence against the MFMA stream)
    const float x0 = lcg(s);
    if (use_op_sel) {  // p = (w.x * x.y, w.y * x.y): both halves of src1 from its HIGH register
      const v2f x = {0.f, x0};
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(p) : "v"(w), "v"(x));
    } else {           // the same products with the operand replicated: no operand select
      const v2f x = {x0, x0};
      asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p) : "v"(w), "v"(x));
    }
    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc) : "v"(p));
    w.y = w.x;
    w.x = x0;
  }
  const unsigned rx = __float_as_uint(acc.x), ry = __float_as_uint(acc.y);
  const bool bad = rx != (unsigned)__builtin_amdgcn_readfirstlane(rx) || ry != (unsigned)__builtin_amdgcn_readfirstlane(ry);
  const unsigned long long mask = __ballot(bad);
  if (lane == 0) {
    atomicAdd(&out[0], 1u);
    if (mask) atomicAdd(&out[1], 1u);
  }
  if (bad) atomicAdd(&out[2 + (lane >> 4)], 1u);
}

#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
  unsigned *d_out, h[8];
  int *d_sink;
  OK(hipMalloc(&d_out, sizeof h));
  OK(hipMalloc(&d_sink, 4));
  hipDeviceProp_t prop;
  OK(hipGetDeviceProperties(&prop, 0));
  printf("%s: 1,024 workgroups x (4 MFMA waves + 4 packed-math waves), 4,000 steps per chain, cadences 0..15, 2 launches each\n", prop.gcnArchName);
  for (int sel = 0; sel < 2; sel++) {
    unsigned long long tot[6] = {};
    int fired = 0;
    for (int nops = 0; nops < 16; nops++) {
      unsigned cell = 0;
      for (int rep = 0; rep < 2; rep++) {
        OK(hipMemset(d_out, 0, sizeof h));
        hipLaunchKernelGGL(repro, dim3(1024), dim3(512), 0, 0, sel, 4000, nops, d_out, d_sink);
        OK(hipDeviceSynchronize());
        OK(hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost));
        for (int k = 0; k < 6; k++) tot[k] += h[k];
        cell += h[1];
      }
      fired += cell != 0;
    }
    printf("%-28s %8llu / %llu victim waves with a wrong lane (at %d of 16 cadences); wrong lanes in 0-15 | 16-31 | 32-47 | 48-63: %llu | %llu | %llu | %llu\n",
           sel ? "v_pk_mul_f32 op_sel:[0,1]:" : "v_pk_mul_f32 (control):", tot[1], tot[0], fired, tot[2], tot[3], tot[4], tot[5]);
  }
  return 0;
}
