// nn_kernels.hip -- K2: the conv + 3xGRU + dense network (src/rnn.c:44-60) on the
// vector path: one 384-thread workgroup per stream, thread = output row / hidden unit,
// int8 weights consumed with v_dot4 (s8 x s8 plus a per-row offset that reproduces the
// x86 s8 x u8 accumulator exactly), float layers as one FMA chain per output in the
// reference's AVX2 order.  Follows oracle/rn_oracle.c (lin_float / lin_int8 / gru_step).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rn_dev.h"

#define NN_THREADS 384

// ---- x86-profile activations (src/vec_avx.h:398-445) with the captured rcpps table ----
__device__ __forceinline__ float rcp_x86(float x, const uint16_t *__restrict__ lut) { return rn_rcp_x86(x, lut); }

__device__ __forceinline__ float tanh_x86(float x, const uint16_t *__restrict__ lut) {
  const float N0 = 952.52801514f, N1 = 96.39235687f, N2 = 0.60863042f;
  const float D0 = 952.72399902f, D1 = 413.36801147f, D2 = 11.88600922f;
  float x2 = x * x;
  float num = fmaf(fmaf(N2, x2, N1), x2, N0);
  float den = fmaf(fmaf(D2, x2, D1), x2, D0);
  num = num * x;
  den = rcp_x86(den, lut);
  num = num * den;
  num = (1.f < num) ? 1.f : num;
  return (-1.f > num) ? -1.f : num;
}

__device__ __forceinline__ float sigmoid_x86(float x, const uint16_t *__restrict__ lut) {
  const float N0 = 238.13200378f, N1 = 6.02452230f, N2 = 0.00950985f;
  const float D0 = 952.72399902f, D1 = 103.34200287f, D2 = 0.74287558f;
  float x2 = x * x;
  float num = fmaf(fmaf(N2, x2, N1), x2, N0);
  float den = fmaf(fmaf(D2, x2, D1), x2, D0);
  num = num * x;
  den = rcp_x86(den, lut);
  num = fmaf(num, den, .5f);
  num = (1.f < num) ? 1.f : num;
  return (0.f > num) ? 0.f : num;
}

// AVX2 vector_ps_to_epi8 (src/vec_avx.h:326-341) -> u8, then re-centred by -128 so that it
// fits the signed operand of v_dot4_i32_i8; the row offset 128*sum(w) restores s8 x u8.
__device__ __forceinline__ int quant_s8(float x) {
  float xf = fmaf(x, 127.f, 127.f);
  int xi = (xf >= -2147483648.f && xf < 2147483648.f) ? (int)rintf(xf) : INT32_MIN;
  int u16 = xi < 0 ? 0 : (xi > 65535 ? 65535 : xi);
  int s16 = (int)(int16_t)(uint16_t)u16;
  int u8 = s16 < 0 ? 0 : (s16 > 255 ? 255 : s16);
  return u8 - 128;
}

__device__ __forceinline__ int pack4(const float *x) {
  return (quant_s8(x[0]) & 0xff) | ((quant_s8(x[1]) & 0xff) << 8) | ((quant_s8(x[2]) & 0xff) << 16) |
         ((quant_s8(x[3]) & 0xff) << 24);
}

// one output row of a block-sparse (or dense, cols==null) int8 layer:
// float(acc_x86) * scale + subias   (src/vec_avx.h:778-877, src/nnet_arch.h:145-151)
__device__ __forceinline__ float int8_row(const RnLinearDev &l, int row, const int *xq) {
  const int grp = row >> 3, sub = row & 7;
  const int *wd = reinterpret_cast<const int *>(l.w);
  int acc = 0;
  if (l.cols) {
    const int b0 = l.grp_start[grp], b1 = l.grp_start[grp + 1];
    for (int b = b0; b < b1; b++) acc = __builtin_amdgcn_sdot4(wd[b * 8 + sub], xq[l.cols[b] >> 2], acc, false);
  } else {
    const int nb = l.nin >> 2;
    const int *wg = wd + (size_t)grp * nb * 8 + sub;
    for (int jb = 0; jb < nb; jb++) acc = __builtin_amdgcn_sdot4(wg[jb * 8], xq[jb], acc, false);
  }
  acc += l.rowsum128[row];
  return (float)acc * l.scale[row] + l.bias[row];
}

struct NnLds {
  float tmp1[196];  // conv1 input  [t-2 | t-1 | t]   (src/nnet.c:118-119)
  float c1[128];
  float tmp2[384];  // conv2 input
  float cat[1536];  // conv2 out | gru1 | gru2 | gru3  (src/rnn.c:53-55)
  int xq[96];       // quantised layer input, 4 per dword
  int hq[96];       // quantised recurrent state
};

extern "C" __global__ void __launch_bounds__(NN_THREADS)
rn_nn_vector_kernel(RnGroupDev g, RnModelDev m, RnTablesDev tb) {
  __shared__ NnLds L;
  const int s = blockIdx.x, t = threadIdx.x;
  const uint16_t *lut = tb.rcp16;
  if (g.silence[s]) {  // src/denoise.c:474: the network and its state are untouched on silent frames
    if (t < RN_NB_BANDS) g.gains[(size_t)s * RN_NB_BANDS + t] = 0;
    if (t == 0) g.vad[s] = 0;
    return;
  }
  float *c1s = g.conv1_state + (size_t)s * 130;
  float *c2s = g.conv2_state + (size_t)s * 256;
  if (t < 130) L.tmp1[t] = c1s[t];
  if (t < 65) L.tmp1[130 + t] = g.features[(size_t)s * 68 + t];
  if (t >= 128) L.tmp2[t - 128] = c2s[t - 128];  // 256 history values
  __syncthreads();
  // conv1: float, 195 -> 128, one FMA chain per output (src/vec_avx.h:672-730), tanh
  if (t < 128) {
    float acc = 0;
    for (int j = 0; j < RN_CONV1_K; j++) acc = fmaf(m.conv1.fw[j * 128 + t], L.tmp1[j], acc);
    float v = tanh_x86(acc + m.conv1.bias[t], lut);
    L.c1[t] = v;
    L.tmp2[256 + t] = v;
  }
  if (t >= 192 && t < 192 + 130) c1s[t - 192] = L.tmp1[65 + t - 192];  // history <- last two frames
  __syncthreads();
  if (t < 96) L.xq[t] = pack4(L.tmp2 + 4 * t);
  if (t >= 128) c2s[t - 128] = L.tmp2[t];  // conv2 history <- tmp2[128..383]
  __syncthreads();
  // conv2: int8 dense 384 -> 384, tanh
  {
    float v = tanh_x86(int8_row(m.conv2, t, L.xq), lut);
    L.cat[t] = v;
  }
  __syncthreads();
  // three GRUs (src/nnet.c:65-94); thread = hidden unit
  for (int k = 0; k < 3; k++) {
    float *st = g.gru_state + ((size_t)k * g.n_stride + s) * RN_GRU;
    const float *xin = L.cat + k * RN_GRU;  // conv2 out, then the previous GRU's new state
    const float h_old = st[t];
    L.cat[(k + 1) * RN_GRU + t] = h_old;
    __syncthreads();
    if (t < 96) L.xq[t] = pack4(xin + 4 * t);
    else if (t >= 128 && t < 224) L.hq[t - 128] = pack4(L.cat + (k + 1) * RN_GRU + 4 * (t - 128));
    __syncthreads();
    const RnLinearDev &wi = m.gru_in[k], &wr = m.gru_rec[k];
    float zi = int8_row(wi, t, L.xq), ri = int8_row(wi, RN_GRU + t, L.xq), hi = int8_row(wi, 2 * RN_GRU + t, L.xq);
    float zr = int8_row(wr, t, L.hq), rr = int8_row(wr, RN_GRU + t, L.hq), hr = int8_row(wr, 2 * RN_GRU + t, L.hq);
    zr += wr.diag[t] * h_old;  // src/nnet_arch.h:153-161
    rr += wr.diag[RN_GRU + t] * h_old;
    hr += wr.diag[2 * RN_GRU + t] * h_old;
    float z = sigmoid_x86(zi + zr, lut);
    float r = sigmoid_x86(ri + rr, lut);
    float h = tanh_x86(hi + hr * r, lut);
    h = z * h_old + (1 - z) * h;
    __syncthreads();
    L.cat[(k + 1) * RN_GRU + t] = h;
    st[t] = h;
  }
  __syncthreads();
  // dense_out (1536 -> 32, FMA chains, sigmoid) and vad_dense (1536 -> 1: the scalar tail of
  // sgemv, src/vec_avx.h:732-736, unfused mul+add)
  if (t < RN_NB_BANDS) {
    float acc = 0;
    for (int j = 0; j < RN_CAT; j++) acc = fmaf(m.dense_out.fw[j * RN_NB_BANDS + t], L.cat[j], acc);
    g.gains[(size_t)s * RN_NB_BANDS + t] = sigmoid_x86(acc + m.dense_out.bias[t], lut);
  } else if (t == 64) {
    float acc = 0;
    for (int j = 0; j < RN_CAT; j++) acc = acc + m.vad_dense.fw[j] * L.cat[j];
    g.vad[s] = sigmoid_x86(acc + m.vad_dense.bias[0], lut);
  }
}

extern "C" hipError_t rn_launch_nn_vector(const RnGroupDev *g, const RnModelDev *m, const RnTablesDev *tb,
                                          hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  RN_LAUNCH(rn_nn_vector_kernel, dim3(g->n_streams), dim3(NN_THREADS), 0, st, e0, e1, *g, *m, *tb);
  return hipGetLastError();
}

// MFMA path: see nn_mfma.hip
