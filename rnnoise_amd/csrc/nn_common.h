// nn_common.h -- pieces shared by the MFMA network kernels (nn_mfma.hip: the fused 16-stream tile kernel;
// nn_layers.hip: the layer-wise kernels for large batches): x86-profile activations, the u8 quantiser, fragment maps.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rn_dev.h"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define TS 16          // streams per workgroup
#define NWAVES 8       // 2 waves per SIMD: one wave's weight loads / epilogue overlap the other's MFMAs
#define NTHREADS (64 * NWAVES)
#define KT 6           // 384 / 64 k-tiles of every int8 layer

// ---- x86-profile activations (same arithmetic as nn_kernels.hip; LUT staged in LDS) ----
__device__ __forceinline__ float rcp_x86(float x, const uint16_t *lut) { return rn_rcp_x86(x, lut); }
__device__ __forceinline__ float tanh_x86(float x, const uint16_t *lut) {  // src/vec_avx.h:398-416
  const float N0 = 952.52801514f, N1 = 96.39235687f, N2 = 0.60863042f;
  const float D0 = 952.72399902f, D1 = 413.36801147f, D2 = 11.88600922f;
  float x2 = x * x;
  float num = fmaf(fmaf(N2, x2, N1), x2, N0);
  float den = fmaf(fmaf(D2, x2, D1), x2, D0);
  num = num * x;
  num = num * rcp_x86(den, lut);
  num = (1.f < num) ? 1.f : num;
  return (-1.f > num) ? -1.f : num;
}
__device__ __forceinline__ float sigmoid_x86(float x, const uint16_t *lut) {  // src/vec_avx.h:426-445
  const float N0 = 238.13200378f, N1 = 6.02452230f, N2 = 0.00950985f;
  const float D0 = 952.72399902f, D1 = 103.34200287f, D2 = 0.74287558f;
  float x2 = x * x;
  float num = fmaf(fmaf(N2, x2, N1), x2, N0);
  float den = fmaf(fmaf(D2, x2, D1), x2, D0);
  num = num * x;
  num = fmaf(num, rcp_x86(den, lut), .5f);
  num = (1.f < num) ? 1.f : num;
  return (0.f > num) ? 0.f : num;
}
__device__ __forceinline__ int quant_s8(float x) {  // src/vec_avx.h:326-341, then -128
  float xf = fmaf(x, 127.f, 127.f);
  int xi = (xf >= -2147483648.f && xf < 2147483648.f) ? (int)rintf(xf) : INT32_MIN;
  int u16 = xi < 0 ? 0 : (xi > 65535 ? 65535 : xi);
  int s16 = (int)(int16_t)(uint16_t)u16;
  int u8 = s16 < 0 ? 0 : (s16 > 255 ? 255 : s16);
  return u8 - 128;
}
__device__ __forceinline__ int pack4(float a, float b, float c, float d) {
  return (quant_s8(a) & 0xff) | ((quant_s8(b) & 0xff) << 8) | ((quant_s8(c) & 0xff) << 16) | ((quant_s8(d) & 0xff) << 24);
}

// byte offset of activation (stream n, input k) inside a B-fragment-ordered buffer:
// [k/64][lane = n + 16*((k%64)/16)][k%16]
__device__ __forceinline__ int frag_off(int n, int k) { return (((k >> 6) * 64 + n + 16 * ((k >> 4) & 3)) << 4) + (k & 15); }


// float(acc_x86)*scale + subias for the 4 rows a lane owns (src/nnet_arch.h:145-151)
__device__ __forceinline__ v4f int8_finish(const RnLinearDev &l, int row0, v4i acc) {
  const v4i rs = *reinterpret_cast<const v4i *>(l.rowsum128 + row0);
  const v4f sc = *reinterpret_cast<const v4f *>(l.scale + row0);
  const v4f sb = *reinterpret_cast<const v4f *>(l.bias + row0);
  v4f o;
#pragma unroll
  for (int r = 0; r < 4; r++) o[r] = (float)(acc[r] + rs[r]) * sc[r] + sb[r];
  return o;
}
