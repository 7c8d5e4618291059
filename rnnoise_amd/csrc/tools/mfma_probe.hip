// mfma_probe.hip -- developer probe (not part of the library): checks on real hardware the
// operand/result lane maps assumed by nn_mfma.hip and that the f32 MFMA is bitwise a k-ordered
// fmaf chain (cdna_hip_programming.md section 3), which is what makes it usable for the
// reference's AVX2 FMA-chain layers.   build: make tools ; run: build/mfma_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

// A: [16][64] int8 row-major, B: [64][16] int8 (k-major), D: [16][16] int32
__global__ void k_i8(const int8_t *A, const int8_t *B, int *D) {
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  v4i a, b, c = {0, 0, 0, 0};
  int8_t ta[16], tb[16];
  for (int q = 0; q < 16; q++) { ta[q] = A[i * 64 + g * 16 + q]; tb[q] = B[(g * 16 + q) * 16 + i]; }
  memcpy(&a, ta, 16);
  memcpy(&b, tb, 16);
  c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; r++) D[(g * 4 + r) * 16 + i] = c[r];  // hypothesis: row = 4*(l>>4)+r (A's index), col = l&15 (B's index)
}

// A: [16][K] f32, B: [K][16] f32 -> D = A.B chained 4 k per MFMA, K multiple of 4
__global__ void k_f32(const float *A, const float *B, float *D, int K) {
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  v4f c = {0, 0, 0, 0};
  for (int k0 = 0; k0 < K; k0 += 4) c = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * K + k0 + g], B[(k0 + g) * 16 + i], c, 0, 0, 0);
  for (int r = 0; r < 4; r++) D[(g * 4 + r) * 16 + i] = c[r];
}

int main() {
  int8_t hA[16 * 64], hB[64 * 16];
  int hD[256], ref[256];
  srand(7);
  for (int q = 0; q < 1024; q++) { hA[q] = (int8_t)(rand() % 255 - 127); hB[q] = (int8_t)(rand() % 255 - 127); }
  for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { int s = 0; for (int k = 0; k < 64; k++) s += hA[i * 64 + k] * hB[k * 16 + j]; ref[i * 16 + j] = s; }
  int8_t *dA, *dB; int *dD;
  hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 1024);
  hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
  k_i8<<<1, 64>>>(dA, dB, dD);
  hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
  int ok = 1, okT = 1;
  for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { ok &= hD[i * 16 + j] == ref[i * 16 + j]; okT &= hD[j * 16 + i] == ref[i * 16 + j]; }
  printf("i8_16x16x64: direct=%d transposed=%d\n", ok, okT);

  const int K = 196;
  static float fA[16 * K], fB[K * 16], fD[256], fr[256], fr2[256];
  for (int q = 0; q < 16 * K; q++) { fA[q] = (float)((rand() % 2001 - 1000) / 7.0) * powf(2.f, rand() % 24 - 12); fB[q] = (float)((rand() % 2001 - 1000) / 3.0); }
  for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
    float s = 0, s2 = 0;
    for (int k = 0; k < K; k++) { s = fmaf(fA[i * K + k], fB[k * 16 + j], s); s2 = s2 + fA[i * K + k] * fB[k * 16 + j]; }
    fr[i * 16 + j] = s; fr2[i * 16 + j] = s2;
  }
  float *gA, *gB, *gD;
  hipMalloc(&gA, sizeof fA); hipMalloc(&gB, sizeof fB); hipMalloc(&gD, sizeof fD);
  hipMemcpy(gA, fA, sizeof fA, hipMemcpyHostToDevice); hipMemcpy(gB, fB, sizeof fB, hipMemcpyHostToDevice);
  k_f32<<<1, 64>>>(gA, gB, gD, K);
  hipMemcpy(fD, gD, sizeof fD, hipMemcpyDeviceToHost);
  int eq = 0, eqT = 0, equnf = 0;
  for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
    eq += !memcmp(&fD[i * 16 + j], &fr[i * 16 + j], 4);
    eqT += !memcmp(&fD[j * 16 + i], &fr[i * 16 + j], 4);
    equnf += !memcmp(&fD[i * 16 + j], &fr2[i * 16 + j], 4);
  }
  printf("f32_16x16x4 chained K=%d: bitwise==fmaf-chain %d/256 (transposed %d/256, ==unfused %d/256)\n", K, eq, eqT, equnf);
  return 0;
}
