// shim.cpp -- host side of librnnoise_amd.so: the rnnoise.h drop-in API, the additive
// batched API (include/rnnoise_amd.h), the "DNNw" blob reader and the GPU re-layout of the
// model.  Plain C ABI outward; HIP runtime inward.  No CPU compute fallback exists: every
// entry point that needs the GPU fails loudly when there is none.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <functional>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rnnoise_amd.h"
#include "rn_dev.h"
#if RN_INSTRUMENT
#include "../../include/rnnoise_amd_debug.h"
#endif
#include "rcp_profiles.h"

extern "C" hipError_t rn_launch_hp(const RnGroupDev *, const void *, int in_s16, int, hipStream_t, hipEvent_t, hipEvent_t);
extern "C" hipError_t rn_launch_analysis(const RnGroupDev *, const RnTablesDev *, int, int, hipStream_t, hipEvent_t, hipEvent_t);
extern "C" hipError_t rn_launch_synthesis(const RnGroupDev *, const RnTablesDev *, void *, int out_s16, int, int, hipStream_t,
                                          hipEvent_t, hipEvent_t);
extern "C" hipError_t rn_launch_train_features(const RnGroupDev *, const RnTablesDev *, const float *, int, int,
                                               const RnTrainArgs *, hipStream_t);
extern "C" hipError_t rn_launch_nn_vector(const RnGroupDev *, const RnModelDev *, const RnTablesDev *, hipStream_t, hipEvent_t,
                                          hipEvent_t);
extern "C" hipError_t rn_launch_nn_one(const RnGroupDev *, const RnModelDev *, const RnTablesDev *, hipStream_t, hipEvent_t, hipEvent_t);
extern "C" hipError_t rn_launch_nn_mfma(const RnGroupDev *, const RnModelDev *, const RnTablesDev *, hipStream_t, hipEvent_t,
                                        hipEvent_t);
extern "C" hipError_t rn_launch_nn_layers(const RnGroupDev *, const RnModelDev *, const RnTablesDev *, hipStream_t, hipEvent_t[5][2]);
extern "C" hipError_t rn_launch_nn_requant(const RnGroupDev *, hipStream_t);
extern "C" int rn_nn_mfma_available(void);
#if RN_INSTRUMENT
extern "C" hipError_t rn_launch_log_energy(const float *, float *, int, hipStream_t);
extern "C" hipError_t rn_launch_fft_probe(int, const float *, float *, unsigned long long *, int, int, const RnTablesDev *, hipStream_t);
extern "C" hipError_t rn_launch_xlane_probe(int *, hipStream_t);
#endif
extern "C" hipError_t rn_launch_state_gather(const RnGroupDev *, float *, int, int, hipStream_t);
extern "C" hipError_t rn_launch_state_scatter(const RnGroupDev *, const float *, int, int, hipStream_t);
extern "C" hipError_t rn_launch_copy_to_host(void *, const void *, size_t, int, hipStream_t);


// Every entry point works on the batch's device and leaves the calling thread's current device as it found it
// (a host thread may be driving another GPU: torch on cuda:0 beside a batch on device 1).
struct DeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    ok = (prev == device) || hipSetDevice(device) == hipSuccess;
    if (prev == device) prev = -1;  // nothing to restore
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};
#define ON_DEVICE(dev)                                                                                 \
  DeviceGuard guard_(dev);                                                                             \
  if (!guard_.ok) {                                                                                    \
    fprintf(stderr, "[rnnoise_amd] cannot select HIP device %d (%s:%d)\n", (dev), __FILE__, __LINE__);  \
    return -1;                                                                                         \
  }

#define HIP_OK(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess) {                                                                            \
      fprintf(stderr, "[rnnoise_amd] %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return -1;                                                                                       \
    }                                                                                                  \
  } while (0)

// =============================================================================================
// "DNNw" weight blob (reference: src/nnet.h:41-62 header, src/parse_lpcnet_weights.c:37-78
// record walk, :80-176 per-layer size checks, src/write_weights.c:46-69 writer)
// =============================================================================================
namespace {

struct BlobRecord {
  std::string name;
  int type, size;
  const uint8_t *data;
};

struct BlobHeader {
  char head[4];
  int32_t version, type, size, block_size;
  char name[44];
};
static_assert(sizeof(BlobHeader) == 64, "DNNw header is 64 bytes");

// Walks the record stream with the reference's acceptance rules (magic/version are not
// checked there either, parse_lpcnet_weights.c:37-52).
bool blob_parse(const void *blob, int len, std::vector<BlobRecord> &out) {
  const uint8_t *p = static_cast<const uint8_t *>(blob);
  while (len > 0) {
    if (len < 64) return false;
    BlobHeader h;
    memcpy(&h, p, 64);
    if (h.block_size < h.size || h.block_size > len - 64 || h.name[43] != 0 || h.size <= 0) return false;
    out.push_back({std::string(h.name), h.type, h.size, p + 64});
    p += 64 + h.block_size;
    len -= 64 + h.block_size;
  }
  return true;
}

const BlobRecord *blob_find(const std::vector<BlobRecord> &recs, const std::string &name, int size) {
  for (const auto &r : recs)
    if (r.name == name) return (size < 0 || r.size == size) ? &r : nullptr;
  return nullptr;
}

// Host view of one layer (pointers alias the blob, like the reference's LinearLayer)
struct HostLinear {
  const float *bias = nullptr, *subias = nullptr, *fw = nullptr, *diag = nullptr, *scale = nullptr;
  const int8_t *w = nullptr;
  const int32_t *idx = nullptr;
  int idx_words = 0, nblocks = 0, nin = 0, nout = 0;
  bool is_int8() const { return w != nullptr; }
};

// kind: 0 float dense, 1 int8 dense, 2 int8 block-sparse, 3 int8 block-sparse + diagonal
bool linear_from_blob(HostLinear &l, const std::vector<BlobRecord> &recs, const std::string &layer, int nin, int nout,
                      int kind) {
  l = HostLinear();
  l.nin = nin;
  l.nout = nout;
  const BlobRecord *r;
  if (!(r = blob_find(recs, layer + "_bias", nout * 4))) return false;
  l.bias = reinterpret_cast<const float *>(r->data);
  if (kind == 0) {
    if (!(r = blob_find(recs, layer + "_weights_float", nin * nout * 4))) return false;
    l.fw = reinterpret_cast<const float *>(r->data);
    return true;
  }
  if (!(r = blob_find(recs, layer + "_subias", nout * 4))) return false;
  l.subias = reinterpret_cast<const float *>(r->data);
  if (!(r = blob_find(recs, layer + "_scale", nout * 4))) return false;
  l.scale = reinterpret_cast<const float *>(r->data);
  if (kind == 1) {
    if (!(r = blob_find(recs, layer + "_weights_int8", nin * nout))) return false;
    l.w = reinterpret_cast<const int8_t *>(r->data);
    l.nblocks = (nin / 4) * (nout / 8);
    return true;
  }
  if (!(r = blob_find(recs, layer + "_weights_idx", -1))) return false;
  l.idx = reinterpret_cast<const int32_t *>(r->data);
  l.idx_words = r->size / 4;
  {  // index stream validation, parse_lpcnet_weights.c:98-121
    int remain = l.idx_words, rows = nout, total = 0;
    const int32_t *idx = l.idx;
    while (remain > 0) {
      int nb = *idx++;
      if (nb < 0 || nb > remain - 1) return false;  // (remain < nb + 1 would overflow for nb == INT_MAX)
      for (int i = 0; i < nb; i++) {
        int pos = *idx++;
        if (pos < 0 || pos + 3 >= nin || (pos & 3)) return false;
      }
      rows -= 8;
      remain -= nb + 1;
      total += nb;
    }
    if (rows != 0) return false;
    l.nblocks = total;
  }
  if (!(r = blob_find(recs, layer + "_weights_int8", 32 * l.nblocks))) return false;
  l.w = reinterpret_cast<const int8_t *>(r->data);
  if (kind == 3) {
    if (!(r = blob_find(recs, layer + "_weights_diag", nout * 4))) return false;
    l.diag = reinterpret_cast<const float *>(r->data);
  }
  return true;
}

struct HostModel {
  HostLinear conv1, conv2, gru_in[3], gru_rec[3], dense_out, vad_dense;
};

// the ten layers of the default architecture and their byte-exact shapes
// (init_rnnoise of the generated rnnoise_data.c)
bool host_model_from_blob(HostModel &m, const void *blob, int len) {
  std::vector<BlobRecord> recs;
  if (!blob || len <= 0 || !blob_parse(blob, len, recs)) return false;
  bool ok = linear_from_blob(m.conv1, recs, "conv1", RN_CONV1_K, RN_CONV1_OUT, 0) &&
            linear_from_blob(m.conv2, recs, "conv2", RN_CONV2_K, RN_CONV2_OUT, 1);
  for (int k = 0; k < 3 && ok; k++) {
    std::string base = "gru" + std::to_string(k + 1);
    ok = linear_from_blob(m.gru_in[k], recs, base + "_input", RN_GRU, RN_GRU3, 2) &&
         linear_from_blob(m.gru_rec[k], recs, base + "_recurrent", RN_GRU, RN_GRU3, 3);
  }
  return ok && linear_from_blob(m.dense_out, recs, "dense_out", RN_CAT, RN_NB_BANDS, 0) &&
         linear_from_blob(m.vad_dense, recs, "vad_dense", RN_CAT, 1, 0);
}

long linear_weight_bytes(const HostLinear &l) {
  if (!l.is_int8()) return 4L * ((long)l.nin * l.nout + l.nout);
  long b = 32L * l.nblocks + 8L * l.nout;  // weights + subias + scale
  if (l.idx) b += 4L * (l.nblocks + l.nout / 8);
  if (l.diag) b += 4L * l.nout;
  return b;
}

// ---------------------------------------------------------------------------------------------
// device arena: one allocation, 256-byte aligned carve-outs
// ---------------------------------------------------------------------------------------------
struct Staging {
  std::vector<uint8_t> bytes;
  size_t add(const void *src, size_t n) {
    size_t off = (bytes.size() + 255) & ~size_t(255);
    bytes.resize(off + n);
    if (src) memcpy(bytes.data() + off, src, n);
    else memset(bytes.data() + off, 0, n);
    return off;
  }
};

struct DevLinearOffsets {
  size_t bias = 0, fw = 0, scale = 0, diag = 0, w = 0, wmf = 0, rowsum = 0, grp = 0, cols = 0;  // (float layers: wmf = MFMA-ordered copy)
  bool has_fw = false, has_diag = false, has_cols = false, is_int8 = false;
};

DevLinearOffsets stage_linear(Staging &st, const HostLinear &l) {
  DevLinearOffsets o;
  o.is_int8 = l.is_int8();
  if (!o.is_int8) {
    o.bias = st.add(l.bias, 4 * l.nout);
    o.fw = st.add(l.fw, 4L * l.nin * l.nout);
    o.has_fw = true;
    if (l.nout % 16 == 0) {
      // copy in the operand order of v_mfma_f32_16x16x4_f32 chains (nn_mfma.hip): [row tile][step / 4][lane][step % 4], the
      // element of step t for lane l being W[k = 4t + (l >> 4)][16 rt + (l & 15)] (0 past the last input): a lane's weights of
      // four consecutive steps are one 16-byte load, a wave's are 1 KB contiguous
      const int steps4 = (l.nin + 15) / 16, RT = l.nout / 16;
      std::vector<float> fm((size_t)RT * steps4 * 64 * 4, 0.f);
      for (int rt = 0; rt < RT; rt++)
        for (int t = 0; t < 4 * steps4; t++)
          for (int lane = 0; lane < 64; lane++) {
            const int k = 4 * t + (lane >> 4);
            if (k < l.nin) fm[(((size_t)rt * steps4 + t / 4) * 64 + lane) * 4 + (t & 3)] = l.fw[(size_t)k * l.nout + 16 * rt + (lane & 15)];
          }
      o.wmf = st.add(fm.data(), 4 * fm.size());
    }
    return o;
  }
  o.bias = st.add(l.subias, 4 * l.nout);  // x86 profile adds subias to int8 layers
  o.scale = st.add(l.scale, 4 * l.nout);
  o.w = st.add(l.w, 32L * l.nblocks);
  std::vector<int32_t> rowsum(l.nout, 0), grp(l.nout / 8 + 1, 0);
  std::vector<uint16_t> cols;
  const int8_t *w = l.w;
  const int32_t *idx = l.idx;
  for (int g = 0; g < l.nout / 8; g++) {
    int nb = idx ? *idx++ : l.nin / 4;
    grp[g + 1] = grp[g] + nb;
    for (int b = 0; b < nb; b++) {
      int col = idx ? *idx++ : 4 * b;
      cols.push_back((uint16_t)col);
      for (int r = 0; r < 8; r++)
        for (int c = 0; c < 4; c++) rowsum[8 * g + r] += w[r * 4 + c];
      w += 32;
    }
  }
  for (auto &v : rowsum) v *= 128;
  o.rowsum = st.add(rowsum.data(), 4 * rowsum.size());
  {  // MFMA copy: zero-fill to dense [nout][nin], then A-fragment order (nn_mfma.hip)
    std::vector<int8_t> dense((size_t)l.nout * l.nin, 0), frag((size_t)l.nout * l.nin, 0);
    const int8_t *wb = l.w;
    for (int g = 0, b = 0; g < l.nout / 8; g++)
      for (; b < grp[g + 1]; b++, wb += 32)
        for (int r = 0; r < 8; r++)
          for (int c = 0; c < 4; c++) dense[(size_t)(8 * g + r) * l.nin + cols[b] + c] = wb[r * 4 + c];
    const int KTn = l.nin / 64;
    for (int rt = 0; rt < l.nout / 16; rt++)
      for (int kt = 0; kt < KTn; kt++)
        for (int lane = 0; lane < 64; lane++)
          for (int e = 0; e < 16; e++)
            frag[(((size_t)rt * KTn + kt) * 64 + lane) * 16 + e] =
                dense[(size_t)(16 * rt + (lane & 15)) * l.nin + 64 * kt + 16 * (lane >> 4) + e];
    o.wmf = st.add(frag.data(), frag.size());
  }
  o.grp = st.add(grp.data(), 4 * grp.size());
  o.cols = st.add(cols.data(), 2 * cols.size());
  o.has_cols = l.idx != nullptr;
  if (l.diag) {
    o.diag = st.add(l.diag, 4 * l.nout);
    o.has_diag = true;
  }
  return o;
}

RnLinearDev resolve_linear(const uint8_t *base, const DevLinearOffsets &o, const HostLinear &l) {
  RnLinearDev d;
  memset(&d, 0, sizeof d);
  d.nin = l.nin;
  d.nout = l.nout;
  d.bias = reinterpret_cast<const float *>(base + o.bias);
  if (o.has_fw) d.fw = reinterpret_cast<const float *>(base + o.fw);
  if (o.has_fw && o.wmf) d.fwm = reinterpret_cast<const float *>(base + o.wmf);
  if (o.is_int8) {
    d.scale = reinterpret_cast<const float *>(base + o.scale);
    d.w = reinterpret_cast<const int8_t *>(base + o.w);
    d.wmf = reinterpret_cast<const int8_t *>(base + o.wmf);
    d.rowsum128 = reinterpret_cast<const int *>(base + o.rowsum);
    d.grp_start = reinterpret_cast<const int *>(base + o.grp);
    if (o.has_cols) d.cols = reinterpret_cast<const uint16_t *>(base + o.cols);
  }
  if (o.has_diag) d.diag = reinterpret_cast<const float *>(base + o.diag);
  return d;
}

// ---------------------------------------------------------------------------------------------
// static tables by formula (reference generator: src/dump_rnnoise_tables.c:54,84-96 and
// src/kiss_fft.c:412-419; band geometry src/denoise.c:63-65,100)
// ---------------------------------------------------------------------------------------------
const int kEband[RN_NB_BANDS + 2] = {0,  2,  4,  6,  8,  10, 12, 15, 18,  21,  24,  28,  32,  36,  41,  47,  53,
                                     60, 68, 77, 87, 98, 110, 124, 140, 157, 176, 198, 223, 251, 282, 317, 356, 400};

struct DeviceTables {
  int device = -1;
  void *mem = nullptr;
  RnTablesDev dev{};
};
std::mutex g_tables_mu;
std::vector<DeviceTables> g_tables;

// ---------------------------------------------------------------------------------------------
// rcpps profile (rcp_profiles.h): which CPU family's approximate reciprocal the activations reproduce.
// $RNNOISE_AMD_RCP_PROFILE = host (default; alias auto) | intel | amd-zen5 (alias amd), or rnnoise_amd_set_rcp_profile().
// "host" captures the table from the CPU this process runs on; if that CPU's rcpps does not have the tabulated form
// (never seen) the built-in table of the cpuid vendor is used and the fact goes to stderr.
// ---------------------------------------------------------------------------------------------
struct RcpProfile {
  unsigned short t[RN_RCP_ENTRIES];
  std::string name;
  bool set = false;
};
RcpProfile g_rcp;  // guarded by g_tables_mu

bool cpu_vendor_is_amd() {
#if defined(__x86_64__) || defined(__i386__)
  unsigned a = 0, b = 0, c = 0, d = 0;
  __asm__ volatile("cpuid" : "=a"(a), "=b"(b), "=c"(c), "=d"(d) : "a"(0), "c"(0));
  return b == 0x68747541u && d == 0x69746e65u && c == 0x444d4163u;  // "AuthenticAMD"
#else
  return false;
#endif
}

int rcp_select_locked(const char *want) {
  std::string w = want ? want : "";
  if (w.empty() || w == "auto") w = "host";
  if (w == "amd") w = "amd-zen5";
  if (w == "host") {
    if (rn_rcp_capture_host(g_rcp.t) == 0) {
      const bool is_intel = !memcmp(g_rcp.t, RN_RCP16_INTEL, sizeof g_rcp.t), is_amd = !memcmp(g_rcp.t, RN_RCP16_AMD_ZEN5, sizeof g_rcp.t);
      g_rcp.name = is_intel ? "host=intel" : (is_amd ? "host=amd-zen5" : "host=captured");
    } else {
      const bool amd = cpu_vendor_is_amd();
      fprintf(stderr, "[rnnoise_amd] this CPU's rcpps does not have the 12-bit table form; using the built-in '%s' profile\n",
              amd ? "amd-zen5" : "intel");
      memcpy(g_rcp.t, amd ? RN_RCP16_AMD_ZEN5 : RN_RCP16_INTEL, sizeof g_rcp.t);
      g_rcp.name = amd ? "amd-zen5" : "intel";
    }
  } else if (w == "intel") {
    memcpy(g_rcp.t, RN_RCP16_INTEL, sizeof g_rcp.t);
    g_rcp.name = "intel";
  } else if (w == "amd-zen5") {
    memcpy(g_rcp.t, RN_RCP16_AMD_ZEN5, sizeof g_rcp.t);
    g_rcp.name = "amd-zen5";
  } else {
    fprintf(stderr, "[rnnoise_amd] unknown rcp profile '%s' (host | intel | amd-zen5)\n", w.c_str());
    return -1;
  }
  g_rcp.set = true;
  return 0;
}

int rcp_ensure_locked() {
  if (g_rcp.set) return 0;
  if (rcp_select_locked(getenv("RNNOISE_AMD_RCP_PROFILE")) == 0) return 0;
  return rcp_select_locked("host");
}

int tables_for_device(int device, RnTablesDev &out) {
  std::lock_guard<std::mutex> lk(g_tables_mu);
  for (auto &t : g_tables)
    if (t.device == device) {
      out = t.dev;
      return 0;
    }
  std::vector<float> window(RN_FRAME_SIZE), dct(RN_NB_BANDS * RN_NB_BANDS), tw(2 * RN_WINDOW_SIZE), frac(400);
  std::vector<uint8_t> band_of(400);
  std::vector<uint16_t> bitrev(RN_WINDOW_SIZE);
  for (int i = 0; i < RN_WINDOW_SIZE; i++) {  // digit reversal for radices 5,3,4,4,4 (src/kiss_fft.c:314-346)
    int j0 = i % 5, j1 = (i / 5) % 3, j2 = (i / 15) % 4, j3 = (i / 60) % 4, j4 = i / 240;
    const int r = j0 * 192 + j1 * 64 + j2 * 16 + j3 * 4 + j4;
    bitrev[i] = (uint16_t)(r + 2 * (r >> 4));  // position in the padded FFT work area (FPAD, dsp_kernels.hip)
  }
  for (int i = 0; i < RN_FRAME_SIZE; i++) {
    double a = .5 * M_PI * (i + .5) / RN_FRAME_SIZE;
    window[i] = (float)sin(.5 * M_PI * sin(a) * sin(a));
  }
  for (int i = 0; i < RN_NB_BANDS; i++)
    for (int j = 0; j < RN_NB_BANDS; j++) {
      float v = (float)cos((i + .5) * j * M_PI / RN_NB_BANDS);
      if (j == 0) v = (float)(v * sqrt(.5));
      dct[i * RN_NB_BANDS + j] = v;
    }
  for (int i = 0; i < RN_WINDOW_SIZE; i++) {
    const double pi = 3.14159265358979323846264338327;
    double phase = (-2 * pi / RN_WINDOW_SIZE) * i;
    tw[2 * i] = (float)cos(phase);
    tw[2 * i + 1] = (float)sin(phase);
  }
  for (int b = 0; b <= RN_NB_BANDS; b++) {
    int bs = kEband[b + 1] - kEband[b];
    for (int j = 0; j < bs; j++) {
      frac[kEband[b] + j] = (float)j / bs;
      band_of[kEband[b] + j] = (uint8_t)b;
    }
  }
  // per-lane twiddles of the register-resident FFT (fft_reg.h): lane l ends a radix-4 stage holding the position whose
  // digit is the bit-reversed lane digit, so every stage's twiddle index is tabulated with the true position
  std::vector<float> ftw(16 * 64 * 2);
  for (int l = 0; l < 64; l++) {
    auto sig = [](int d) { return ((d & 1) << 1) | (d >> 1); };
    const int p4 = sig(l & 3), p16 = p4 + 4 * sig((l >> 2) & 3), q = p16 + 16 * sig(l >> 4);
    auto put = [&](int row, int e) {
      ftw[(row * 64 + l) * 2] = tw[2 * (e % RN_WINDOW_SIZE)];
      ftw[(row * 64 + l) * 2 + 1] = tw[2 * (e % RN_WINDOW_SIZE) + 1];
    };
    put(0, 60 * p4 * ((l >> 2) & 3));   // m = 4 stage,  fstride 60 (src/kiss_fft.c:141-165)
    put(1, 15 * p16 * (l >> 4));        // m = 16 stage, fstride 15
    put(2, 5 * q);                      // radix 3, m = 64, fstride 5 (:201-225)
    put(3, 10 * q);
    for (int t = 0; t < 3; t++)         // radix 5, m = 192, fstride 1 (:269-302), j = 64 t + q
      for (int mlt = 1; mlt <= 4; mlt++) put(4 + 4 * t + (mlt - 1), mlt * (64 * t + q));
  }
  // band-sum layout (dsp_kernels.hip: band_products / band_chain): accumulator k's terms -- band k-1's `frac` parts, then
  // band k's `1-frac` parts, in bin order (src/denoise.c:90-113) -- sit contiguously from a 16-byte aligned start, so the
  // serial sum reads them four at a time.  band_q[bin] = address of the (1-frac) term | address of the frac term << 11 |
  // band << 22;  band_chain[k] = start | length << 16.
  std::vector<uint32_t> band_q(400), band_chain(RN_NB_BANDS + 2);
  std::vector<uint16_t> band_pad(64);
  {
    int start[RN_NB_BANDS + 2], lo[RN_NB_BANDS + 2], pos = 0;
    // Lane k sums accumulator k with 16-byte reads.  ds_read_b128 serves a wave in four groups of 16 lanes
    // ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...; MI355X_MICROARCH.md, LDS) over 64 banks, i.e. 16 slots of 16 bytes: the
    // reads of a group are conflict-free when its lanes' slot numbers (start / 4 mod 16) differ, and all lanes advance in
    // lock-step, so it is enough to place the STARTS that way (a few floats of padding).
    auto group_of = [](int lane) {
      const int l = lane & 31, hi = lane >> 5;
      const bool a = l < 4 || (l >= 12 && l < 16) || (l >= 20 && l < 28);
      return 2 * hi + (a ? 0 : 1);
    };
    bool used[4][16] = {};
    for (int k = 0; k < RN_NB_BANDS + 2; k++) {
      lo[k] = k ? kEband[k - 1] : 0;
      const int len = (k <= RN_NB_BANDS ? kEband[k + 1] : 400) - lo[k];
      while (used[group_of(k)][(pos / 4) % 16]) pos += 4;
      used[group_of(k)][(pos / 4) % 16] = true;
      start[k] = pos;
      band_chain[k] = (uint32_t)pos | ((uint32_t)len << 16);
      pos += (len + 3) & ~3;
    }
    if (pos > RN_BAND_QSTRIDE) {  // one product array of the analysis / synthesis kernels (dsp_kernels.hip)
      fprintf(stderr, "[rnnoise_amd] band-sum layout needs %d floats\n", pos);
      return -1;
    }
    // The serial sums read whole 16-byte slots: the 1..3 floats between an accumulator's last term and the end of its last
    // slot hold +0.0f (adding it changes no bit), written by the first lanes of the wave before every use -- band_pad[lane]
    // = the lane's pad float (there are 48; the remaining lanes rewrite the last one).
    {
      int np = 0;
      for (int k = 0; k < RN_NB_BANDS + 2; k++) {
        const int len = (int)(band_chain[k] >> 16), st0 = (int)(band_chain[k] & 0xffff);
        for (int i = len; i < ((len + 3) & ~3); i++) {
          if (np >= 64) {
            fprintf(stderr, "[rnnoise_amd] band-sum layout has more than 64 pad floats\n");
            return -1;
          }
          band_pad[np++] = (uint16_t)(st0 + i);
        }
      }
      for (int i = np; i < 64; i++) band_pad[i] = band_pad[np - 1];
    }
    for (int b = 0; b <= RN_NB_BANDS; b++)
      for (int bin = kEband[b]; bin < kEband[b + 1]; bin++)
        band_q[bin] = (uint32_t)(start[b] + bin - lo[b]) | ((uint32_t)(start[b + 1] + bin - lo[b + 1]) << 11) | ((uint32_t)b << 22);
  }
  if (rcp_ensure_locked()) return -1;
  Staging st;
  size_t o_ftw = st.add(ftw.data(), 4 * ftw.size());
  size_t o_bq = st.add(band_q.data(), 4 * band_q.size()), o_bc = st.add(band_chain.data(), 4 * band_chain.size()),
         o_bp = st.add(band_pad.data(), 2 * band_pad.size());
  size_t o_w = st.add(window.data(), 4 * window.size()), o_d = st.add(dct.data(), 4 * dct.size()),
         o_t = st.add(tw.data(), 4 * tw.size()), o_f = st.add(frac.data(), 4 * frac.size()),
         o_b = st.add(band_of.data(), band_of.size()), o_r = st.add(g_rcp.t, sizeof g_rcp.t),
         o_br = st.add(bitrev.data(), 2 * bitrev.size());
  DeviceTables t;
  t.device = device;
  ON_DEVICE(device);
  HIP_OK(hipMalloc(&t.mem, st.bytes.size()));
  HIP_OK(hipMemcpy(t.mem, st.bytes.data(), st.bytes.size(), hipMemcpyHostToDevice));
  const uint8_t *base = static_cast<const uint8_t *>(t.mem);
  t.dev.half_window = reinterpret_cast<const float *>(base + o_w);
  t.dev.dct = reinterpret_cast<const float *>(base + o_d);
  t.dev.twiddles = reinterpret_cast<const float *>(base + o_t);
  t.dev.band_frac = reinterpret_cast<const float *>(base + o_f);
  t.dev.band_of_bin = base + o_b;
  t.dev.bitrev = reinterpret_cast<const uint16_t *>(base + o_br);
  t.dev.rcp16 = reinterpret_cast<const uint16_t *>(base + o_r);
  t.dev.fft_tw = reinterpret_cast<const float *>(base + o_ftw);
  t.dev.band_q = reinterpret_cast<const uint32_t *>(base + o_bq);
  t.dev.band_chain = reinterpret_cast<const uint32_t *>(base + o_bc);
  t.dev.band_pad = reinterpret_cast<const uint16_t *>(base + o_bp);
  t.dev.dct_scale = sqrt(2. / 22);
  g_tables.push_back(t);
  out = t.dev;
  return 0;
}

struct DeviceModel {
  int device = -1;
  void *mem = nullptr, *mem_rows = nullptr;  // the staged model; the row-major int8 copies of the vector path
  RnModelDev dev{};
};

}  // namespace

// =============================================================================================
// public types
// =============================================================================================
struct StatePool;
namespace { struct StagedModel; }
struct RNNModel {
  const void *const_blob = nullptr;  // borrowed (rnnoise_model_from_buffer)
  void *blob = nullptr;              // owned (rnnoise_model_from_file)
  int blob_len = 0;
  FILE *file = nullptr;
  std::mutex mu;
  int parsed = 0;  // 0 not yet, 1 ok, -1 rejected
  HostModel host;                  // layer views into a "DNNw" blob (unused for a packed model)
  StagedModel *staged = nullptr;   // device layout of every layer, built from the blob or taken from an "RNPK" pack
  long weight_bytes = 0;           // SURVEY 8d "W"
  std::vector<DeviceModel> dev;
  std::vector<StatePool *> pools;  // device-resident one-stream states behind rnnoise_create / rnnoise_process_frame
  const void *bytes() const { return blob ? blob : const_blob; }
};

struct RNNoiseBatch {
  RNNModel *model = nullptr;
  int device = 0, n = 0, nn_path = 0;
  bool img_valid = false;  // g.act_q[1..3] mirror gru_state (rn_dev.h); cleared by whatever else writes the state
  int schedule = 0;  // 0: default (3-stream frame pipeline in multi-frame calls); 9: one stream; 1: only the high-pass aside
  int parity = 0;  // spectra slot (mod RN_SPEC_SLOTS) the next frame writes; the previous one holds the delayed spectra
  long frame_no = 0;  // selects the per-step scratch copy (features / silence / pitch are double-buffered)
  float *features2[2] = {nullptr, nullptr};
  int *silence2[2] = {nullptr, nullptr}, *pitch2[2] = {nullptr, nullptr};
  int ring_slot = 0;  // pitch-ring slot the next frame is written to
  // side stream + events: in multi-frame calls the (latency-bound, 1 lane per stream) high-pass of frame
  // f+1 runs beside analysis/network/synthesis of frame f
  hipStream_t side = nullptr, side_hp = nullptr;
  // ordering events of the pipelined schedule: own_* are the batch's persistent events, cur_* the handle that marks the
  // completion of hp / analysis / synthesis of frame f & 7 (an own_* event, or the stop event of a timed launch)
  hipEvent_t ev_begin = nullptr, own_hp[8] = {}, own_k1[8] = {}, own_k3[8] = {};
  hipEvent_t cur_hp[8] = {}, cur_k1[8] = {}, cur_k3[8] = {};
  void *arena = nullptr;
  size_t arena_bytes = 0;
  RnGroupDev g{};
  RnModelDev m{};
  RnTablesDev tb{};
  float *scratch_gains = nullptr, *scratch_vad = nullptr;
  float *debug_buf = nullptr;
  float *state_stage = nullptr;  // one flat state in HBM: export / import go through the gather / scatter kernels
  // host-fed path (rnnoise_batch_process): two chunks in flight -- H2D of chunk i+1 and D2H of chunk i-1 overlap the
  // kernels of chunk i; pinned bounce buffers are used only when the caller's memory is pageable
  struct HostIo {
    hipStream_t up = nullptr, run = nullptr, down = nullptr;
    hipEvent_t up_done[2] = {}, run_done[2] = {}, down_done[2] = {};
    float *d_in[2] = {}, *d_out[2] = {}, *d_vad[2] = {}, *d_gains[2] = {};
    float *h_in[2] = {}, *h_out[2] = {}, *h_vad[2] = {}, *h_gains[2] = {};
    int chunk_frames = 0;
    size_t pcm_floats = 0;  // capacity of d_in / d_out (and h_in / h_out) of one chunk
    // pinned callers: a ring of RING frame slots, filled and drained frame by frame beside the kernels
    static constexpr int RING = 6;
    char *ring_mem = nullptr;
    hipEvent_t r_k3[RING] = {}, r_down[RING] = {}, r_up[RING] = {}, r_hp[RING] = {};
  } io;
  // timing
  bool timing = false;
  struct Ev { hipEvent_t a, b; int kind; };
  std::vector<Ev> pending, pool;
  double ms_sum[4] = {0, 0, 0, 0};  // analysis, network, synthesis, high-pass
  long launches = 0;
};

// A pool of device-resident one-stream states of one model on one device: the arrays of a POOL_SLOTS-stream batch, of
// which every rnnoise_create() owns one row.  The kernels are pointed at a row through a one-stream view (rn_dev.h:
// n_stride), so a frame of a pooled state costs one 1,920-byte upload, four launches and one 1,924-byte download.
struct StatePool {
  static constexpr int POOL_SLOTS = 64;
  RNNoiseBatch *batch = nullptr;   // owns the arena; never processed as a whole
  static constexpr int FLAT_IO = RN_STATE_FLOATS + 2;           // frame offset inside a staging block (16-byte aligned)
  static constexpr int FLAT_BLK = FLAT_IO + RN_FRAME_SIZE + 4;  // state | pad | frame (in, then out in place) | vad | pad
  float *d_io = nullptr;           // [POOL_SLOTS][2][484]: in[480] | pad, out[480] | vad | pad
  float *d_flat = nullptr;         // [POOL_SLOTS][FLAT_BLK] staging for self-contained states (rnnoise_init path)
  std::mutex mu;
  unsigned long long used = 0;     // bit per slot
};

// What a DenoiseState holds when it came from rnnoise_create(): a row of a StatePool plus the host-side frame
// bookkeeping of that row and its own stream / pinned buffers, so that states on different threads run concurrently.
struct PooledRef {
  StatePool *pool;
  int slot;
  int parity, ring_slot;
  long frame_no;
  hipStream_t stream;
  float *h_io;        // pinned [2][484]
  std::mutex *mu;     // one frame at a time per state (the reference's states are not re-entrant either)
};

struct DenoiseState {
  uint32_t magic;
  uint32_t pad;
  RNNModel *model;
  union {
    float state[RN_STATE_FLOATS];  // rnnoise_init() on caller memory: self-contained POD, no library-owned resource (SURVEY 8b "Types")
    PooledRef ref;                 // rnnoise_create(): device-resident, released by rnnoise_destroy()
  };
};
static const uint32_t kStateMagic = 0x524e4e41u;   // "RNNA": self-contained state
static const uint32_t kPooledMagic = 0x524e4e50u;  // "RNNP": row of a StatePool

namespace {

// ---------------------------------------------------------------------------------------------
// "RNPK": the GPU-native packed model.  What model_on_device() uploads -- every layer already in
// its device layout: int8 blocks in exporter order + column / group tables for the vector path, the same weights
// zero-filled to dense and pre-swizzled into MFMA A-fragment order, row sums, float layers as they are -- preceded by a
// header with a version tag, the architecture the layouts were made for and the per-layer offsets.  Loading a pack skips
// the blob walk and the re-layout; rnnoise_model_from_buffer / _file / _filename accept either format.
// ---------------------------------------------------------------------------------------------
struct PackLayer {
  uint64_t bias, fw, scale, diag, w, wmf, rowsum, grp, cols;
  uint32_t has_fw, has_diag, has_cols, is_int8;
  int32_t nin, nout, nblocks, pad;
};
struct PackHeader {
  char magic[4];        // "RNPK"
  uint32_t version;     // RN_PACK_VERSION
  uint32_t dims[8];     // conv1 in/out, conv2 in/out, GRU size, concat size, bands, MFMA k-tile (64)
  int64_t weight_bytes; // SURVEY 8d "W" of the source blob
  uint64_t payload_bytes;
  PackLayer layers[10]; // conv1, conv2, gru1..3 input, gru1..3 recurrent (interleaved in, rec), dense_out, vad_dense
};
static const uint32_t RN_PACK_VERSION = 2;  // 2: float layers carry an MFMA-ordered copy
static const uint32_t kPackDims[8] = {RN_CONV1_K, RN_CONV1_OUT, RN_CONV2_K, RN_CONV2_OUT, RN_GRU, RN_CAT, RN_NB_BANDS, 64};

struct StagedModel {
  Staging st;
  DevLinearOffsets off[10];
  HostLinear lin[10];  // only nin / nout / nblocks are meaningful for a model that came from a pack
};

long host_weight_bytes(const HostModel &h) {
  long w = linear_weight_bytes(h.conv1) + linear_weight_bytes(h.conv2) + linear_weight_bytes(h.dense_out) +
           linear_weight_bytes(h.vad_dense);
  for (int k = 0; k < 3; k++) w += linear_weight_bytes(h.gru_in[k]) + linear_weight_bytes(h.gru_rec[k]);
  return w;
}

void stage_model(const HostModel &h, StagedModel &sm) {
  const HostLinear *order[10] = {&h.conv1, &h.conv2, &h.gru_in[0], &h.gru_rec[0], &h.gru_in[1], &h.gru_rec[1],
                                 &h.gru_in[2], &h.gru_rec[2], &h.dense_out, &h.vad_dense};
  for (int i = 0; i < 10; i++) {
    sm.lin[i] = *order[i];
    sm.off[i] = stage_linear(sm.st, *order[i]);
  }
}

bool is_pack(const void *p, int len) { return p && len >= (int)sizeof(PackHeader) && !memcmp(p, "RNPK", 4); }

// header + payload of a pack -> staged form (bounds-checked: a pack is untrusted input like a blob)
bool unpack_model(const void *p, int len, StagedModel &sm, long &weight_bytes) {
  PackHeader h;
  memcpy(&h, p, sizeof h);
  if (h.version != RN_PACK_VERSION || memcmp(h.dims, kPackDims, sizeof kPackDims)) return false;
  if (h.payload_bytes != (uint64_t)len - sizeof h || h.weight_bytes <= 0) return false;
  const uint64_t n = h.payload_bytes;
  static const int want[10][2] = {{RN_CONV1_K, RN_CONV1_OUT}, {RN_CONV2_K, RN_CONV2_OUT}, {RN_GRU, RN_GRU3}, {RN_GRU, RN_GRU3},
                                  {RN_GRU, RN_GRU3}, {RN_GRU, RN_GRU3}, {RN_GRU, RN_GRU3}, {RN_GRU, RN_GRU3},
                                  {RN_CAT, RN_NB_BANDS}, {RN_CAT, 1}};
  for (int i = 0; i < 10; i++) {
    const PackLayer &l = h.layers[i];
    // What kind of layer sits at position i is the architecture's business, not the file's: conv1 / dense_out / vad_dense
    // float, conv2 dense int8, gru input matrices block-sparse int8, recurrent ones block-sparse int8 + diagonal
    // (linear_from_blob's `kind`).  A pack whose flags say otherwise would make the kernels dereference null weight /
    // scale / diagonal pointers or take the dense branch over a sparse weight array.
    const bool k_int8 = i >= 1 && i <= 7, k_diag = i == 3 || i == 5 || i == 7, k_cols = i >= 2 && i <= 7;
    if ((l.is_int8 != 0) != k_int8 || (l.has_fw != 0) != !k_int8 || (l.has_diag != 0) != k_diag || (l.has_cols != 0) != k_cols)
      return false;
    if (l.nin != want[i][0] || l.nout != want[i][1] || l.nblocks < 0 || l.nblocks > (l.nin / 4) * (l.nout / 8)) return false;
    if (k_int8 && !k_cols && l.nblocks != (l.nin / 4) * (l.nout / 8)) return false;  // a dense int8 layer has every block
    auto fits = [&](uint64_t off, uint64_t bytes) { return off <= n && bytes <= n - off && !(off & 15); };
    const uint64_t no = l.nout, ni = l.nin;
    if (!fits(l.bias, 4 * no)) return false;
    if (l.is_int8) {
      if (!fits(l.scale, 4 * no) || !fits(l.w, 32ull * l.nblocks) || !fits(l.wmf, no * ni) || !fits(l.rowsum, 4 * no) ||
          !fits(l.grp, 4 * (no / 8 + 1)) || !fits(l.cols, 2ull * l.nblocks))
        return false;
      if (l.has_diag && !fits(l.diag, 4 * no)) return false;
      // the group / column tables index the weight array: they must stay inside it
      const int32_t *grp = reinterpret_cast<const int32_t *>(static_cast<const uint8_t *>(p) + sizeof h + l.grp);
      const uint16_t *cols = reinterpret_cast<const uint16_t *>(static_cast<const uint8_t *>(p) + sizeof h + l.cols);
      if (grp[0] != 0 || grp[no / 8] != l.nblocks) return false;
      for (uint64_t gidx = 0; gidx < no / 8; gidx++)
        if (grp[gidx + 1] < grp[gidx]) return false;
      for (int b = 0; b < l.nblocks; b++)
        if (cols[b] + 3 >= l.nin || (cols[b] & 3)) return false;
    } else if (!fits(l.fw, 4 * no * ni) || (no % 16 == 0 && (!l.wmf || !fits(l.wmf, 4 * no * ((ni + 15) / 16) * 16)))) {
      return false;
    }
    DevLinearOffsets &o = sm.off[i];
    o.bias = l.bias; o.fw = l.fw; o.scale = l.scale; o.diag = l.diag; o.w = l.w; o.wmf = l.wmf; o.rowsum = l.rowsum;
    o.grp = l.grp; o.cols = l.cols;
    o.has_fw = !k_int8; o.has_diag = k_diag; o.has_cols = k_cols; o.is_int8 = k_int8;
    if (!k_int8 && no % 16 != 0) o.wmf = 0;  // (vad_dense: no MFMA-ordered copy; whatever the file says there is not used)
    sm.lin[i] = HostLinear();
    sm.lin[i].nin = l.nin;
    sm.lin[i].nout = l.nout;
    sm.lin[i].nblocks = l.nblocks;
  }
  const uint8_t *payload = static_cast<const uint8_t *>(p) + sizeof h;
  sm.st.bytes.assign(payload, payload + n);
  weight_bytes = (long)h.weight_bytes;
  return true;
}

int model_parse_locked(RNNModel *m) {
  if (m->parsed == 0) {
    if (is_pack(m->bytes(), m->blob_len)) {
      m->staged = new StagedModel();
      m->parsed = unpack_model(m->bytes(), m->blob_len, *m->staged, m->weight_bytes) ? 1 : -1;
    } else {
      m->parsed = host_model_from_blob(m->host, m->bytes(), m->blob_len) ? 1 : -1;
      if (m->parsed == 1) m->weight_bytes = host_weight_bytes(m->host);
    }
  }
  return m->parsed == 1 ? 0 : -1;
}

int model_on_device(RNNModel *m, int device, RnModelDev &out) {
  std::lock_guard<std::mutex> lk(m->mu);
  if (model_parse_locked(m)) return -1;
  for (auto &d : m->dev)
    if (d.device == device) {
      out = d.dev;
      return 0;
    }
  if (!m->staged) {  // "DNNw" blob: re-layout once per process
    m->staged = new StagedModel();
    stage_model(m->host, *m->staged);
  }
  const StagedModel &sm = *m->staged;
  DeviceModel d;
  d.device = device;
  ON_DEVICE(device);
  HIP_OK(hipMalloc(&d.mem, sm.st.bytes.size()));
  HIP_OK(hipMemcpy(d.mem, sm.st.bytes.data(), sm.st.bytes.size(), hipMemcpyHostToDevice));
  const uint8_t *base = static_cast<const uint8_t *>(d.mem);
  RnLinearDev *dst[10] = {&d.dev.conv1, &d.dev.conv2, &d.dev.gru_in[0], &d.dev.gru_rec[0], &d.dev.gru_in[1], &d.dev.gru_rec[1],
                          &d.dev.gru_in[2], &d.dev.gru_rec[2], &d.dev.dense_out, &d.dev.vad_dense};
  for (int i = 0; i < 10; i++) *dst[i] = resolve_linear(base, sm.off[i], sm.lin[i]);
  {  // row-major copies of the int8 layers (rn_dev.h: wrow / cq / grp4), derived from the staged block streams
    Staging rows;
    size_t o_w[10] = {}, o_c[10] = {}, o_g[10] = {};
    for (int i = 1; i <= 7; i++) {
      const DevLinearOffsets &o = sm.off[i];
      const int nout = sm.lin[i].nout, ng = nout / 8;
      const int8_t *w = reinterpret_cast<const int8_t *>(sm.st.bytes.data() + o.w);
      const int32_t *grp = reinterpret_cast<const int32_t *>(sm.st.bytes.data() + o.grp);
      const uint16_t *cols = reinterpret_cast<const uint16_t *>(sm.st.bytes.data() + o.cols);
      std::vector<int32_t> g4(ng + 1, 0);
      for (int g = 0; g < ng; g++) g4[g + 1] = g4[g] + (grp[g + 1] - grp[g] + 3) / 4;
      std::vector<int32_t> wrow((size_t)g4[ng] * 8 * 4, 0);
      std::vector<uint32_t> cq((size_t)g4[ng], 0);
      for (int g = 0; g < ng; g++) {
        const int len = grp[g + 1] - grp[g];
        for (int k = 0; k < len; k++) {
          const int b = grp[g] + k;
          const uint32_t col4 = o.has_cols ? cols[b] >> 2 : (uint32_t)k;
          cq[g4[g] + k / 4] |= col4 << (8 * (k & 3));
          for (int sub = 0; sub < 8; sub++)
            memcpy(&wrow[((size_t)(g4[g] + k / 4) * 8 + sub) * 4 + (k & 3)], w + (size_t)b * 32 + sub * 4, 4);
        }
      }
      o_w[i] = rows.add(wrow.data(), 4 * wrow.size());
      o_c[i] = rows.add(cq.data(), 4 * cq.size());
      o_g[i] = rows.add(g4.data(), 4 * g4.size());
    }
    size_t o_fw4 = 0;
    {  // dense_out (layer 8), float: [input / 4][output][input % 4]
      const float *fw = reinterpret_cast<const float *>(sm.st.bytes.data() + sm.off[8].fw);
      const int nin = sm.lin[8].nin, nout = sm.lin[8].nout;
      std::vector<float> fw4((size_t)nin * nout);
      for (int j = 0; j < nin; j++)
        for (int i = 0; i < nout; i++) fw4[((size_t)(j / 4) * nout + i) * 4 + (j & 3)] = fw[(size_t)j * nout + i];
      o_fw4 = rows.add(fw4.data(), 4 * fw4.size());
    }
    HIP_OK(hipMalloc(&d.mem_rows, rows.bytes.size()));
    HIP_OK(hipMemcpy(d.mem_rows, rows.bytes.data(), rows.bytes.size(), hipMemcpyHostToDevice));
    const uint8_t *rb = static_cast<const uint8_t *>(d.mem_rows);
    for (int i = 1; i <= 7; i++) {
      dst[i]->wrow = reinterpret_cast<const int *>(rb + o_w[i]);
      dst[i]->cq = reinterpret_cast<const uint32_t *>(rb + o_c[i]);
      dst[i]->grp4 = reinterpret_cast<const int *>(rb + o_g[i]);
    }
    dst[8]->fw4 = reinterpret_cast<const float *>(rb + o_fw4);
  }
  m->dev.push_back(d);
  out = d.dev;
  return 0;
}

template <typename T>
T *carve(uint8_t *&p, size_t count) {
  T *r = reinterpret_cast<T *>(p);
  p += (count * sizeof(T) + 255) & ~size_t(255);
  return r;
}

// Batches from this size up run the network layer by layer (nn_layers.hip: 64 streams per GRU workgroup); below it the
// five launches and the smaller grids cost more than the weight reuse gains.  $RNNOISE_AMD_NN_LAYERS_MIN overrides (A/B runs).
int nn_layers_min_streams() {
  static const int v = [] {
    const char *e = getenv("RNNOISE_AMD_NN_LAYERS_MIN");
    return e ? atoi(e) : 16384;
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

size_t batch_layout(RnGroupDev &g, uint8_t *base, int n) {
  uint8_t *p = base;
  size_t N = n;
  g.n_streams = n;
  g.n_stride = n;
  g.mem_hp = carve<float>(p, 2 * N);
  g.pitch_ring = carve<float>(p, RN_RING_SIZE * N);
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

// rows [first, first + count) of a batch as a group of their own (rn_dev.h: n_stride keeps the plane strides)
RnGroupDev group_view(const RnGroupDev &g, int first, int count) {
  RnGroupDev v = g;
  const size_t f = first;
  v.n_streams = count;
  v.mem_hp += 2 * f;
  v.pitch_ring += RN_RING_SIZE * f;
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
// Select the rcpps profile (rcp_profiles.h) for every model and batch of this process, now and later: the table is
// re-uploaded to each device that already holds one (after draining it).  name: "host" | "intel" | "amd-zen5".
extern "C" int rnnoise_amd_set_rcp_profile(const char *name) {
  std::lock_guard<std::mutex> lk(g_tables_mu);
  RcpProfile keep = g_rcp;
  if (rcp_select_locked(name && *name ? name : "host")) {
    g_rcp = keep;
    return -1;
  }
  for (auto &t : g_tables) {
    ON_DEVICE(t.device);
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(const_cast<uint16_t *>(t.dev.rcp16), g_rcp.t, sizeof g_rcp.t, hipMemcpyHostToDevice));
  }
  return 0;
}

// Name of the active profile: "intel", "amd-zen5", or "host=intel" / "host=amd-zen5" / "host=captured" when the table was
// taken from this CPU (and which built-in table, if any, it equals).  The pointer stays valid until the next set call.
extern "C" const char *rnnoise_amd_rcp_profile(void) {
  std::lock_guard<std::mutex> lk(g_tables_mu);
  if (rcp_ensure_locked()) return "";
  return g_rcp.name.c_str();
}

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
  if (device < 0 || device >= rnnoise_amd_device_count()) {
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

static void host_io_release(RNNoiseBatch *b);
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

// Hooks of the host-fed path: the frame buffers of a call are then a ring of `ring` frame slots in HBM (frame f lives in
// slot f % ring) that uploads fill and downloads drain while the kernels run; each hook is called on the host right where
// the named kernel of frame f is enqueued, with the stream it goes to.
struct FrameIoHooks {
  int ring = 0;
  std::function<int(int, hipStream_t)> before_hp, after_hp, before_nn, after_k3;
};

// PCM frames are float (the reference API's sample type) or, with s16 set, int16 converted at the two ends of the step as the
// reference's only caller does (examples/rnnoise_demo.c:56,58): half the bytes over HBM and, in the host-fed path, PCIe.
static int batch_process_device_impl(RNNoiseBatch *b, void *d_out_v, const void *d_in_v, float *d_vad, float *d_gains,
                                     int n_frames, void *hip_stream, bool s16, const FrameIoHooks *hk = nullptr) {
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
  if (side_k1 && !b->side) HIP_OK(hipStreamCreateWithFlags(&b->side, hipStreamNonBlocking));
  if (pipelined && !b->side_hp) {
    HIP_OK(hipStreamCreateWithFlags(&b->side_hp, hipStreamNonBlocking));
    // ordering between streams of ONE device: no system-scope fence (it writes back and invalidates the caches at
    // every record, which the next kernels then pay for)
    const unsigned evf = hipEventDisableTiming | hipEventDisableSystemFence;
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
    static const bool hp_early = getenv("RNNOISE_AMD_HP_EARLY") != nullptr;  // A/B runs only: the ring-bound start
    if (side_k1 && f >= 4 && !hp_early) HIP_OK(hipStreamWaitEvent(sc, b->cur_k3[(f - 4) & 7], 0));
    if (hk && hk->before_hp(f, sc)) return -1;
    {
      TimedLaunch t(b, 3);
      b->cur_hp[f & 7] = t.on ? t.stop() : (pipelined ? b->own_hp[f & 7] : nullptr);
      HIP_OK(rn_launch_hp(&b->g, d_in + buf(f) * N * RN_FRAME_SIZE * esz, s16, (b->ring_slot + f) % RN_RING_SLOTS, sc, t.start(),
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
      if (f + 3 < n_frames && highpass(f + 3)) return -1;
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
        // five launches, each timed on its own (kind 1: the durations add up to the network's)
        TimedLaunch t0(b, 1), t1(b, 1), t2(b, 1), t3(b, 1), t4(b, 1);
        hipEvent_t ev[5][2] = {{t0.start(), t0.stop()}, {t1.start(), t1.stop()}, {t2.start(), t2.stop()}, {t3.start(), t3.stop()},
                               {t4.start(), t4.stop()}};
        HIP_OK(rn_launch_nn_layers(&g, &b->m, &b->tb, st, ev));
      } else {
        TimedLaunch t(b, 1);
        b->img_valid = false;
        if (b->nn_path >= 1) HIP_OK(rn_launch_nn_mfma(&g, &b->m, &b->tb, st, t.start(), t.stop()));
        else if (g.n_streams <= nn_one_max_streams()) HIP_OK(rn_launch_nn_one(&g, &b->m, &b->tb, st, t.start(), t.stop()));
        else HIP_OK(rn_launch_nn_vector(&g, &b->m, &b->tb, st, t.start(), t.stop()));
      }
    }
    {
      TimedLaunch t(b, 2);
      b->cur_k3[f & 7] = t.on ? t.stop() : (side_k1 ? b->own_k3[f & 7] : nullptr);
      HIP_OK(rn_launch_synthesis(&g, &b->tb, d_out + buf(f) * N * RN_FRAME_SIZE * esz, s16, cur, prev, st, t.start(), b->cur_k3[f & 7]));
    }
    if (hk && hk->after_k3(f, st)) return -1;
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

static void host_io_release(RNNoiseBatch *b) {
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
  // Copies.  Uploads and downloads ALTERNATE ON ONE COPY STREAM (io.down), in the order the frame pipeline asks for them:
  // each then finds the DMA engine free.  With two copies in flight the runtime executes one of them as a blit kernel (256
  // workgroups x 512 lanes) whose PCIe-bound stores stall what runs beside it -- the analysis kernel took 2.3 ms instead of
  // 1.1 (rocprofv3 kernel + memory-copy trace) -- and a copy stream per direction plus the pipeline's three streams is more
  // than the four hardware queues the runtime multiplexes streams onto.  $RNNOISE_AMD_HOSTIO_COPY=hp (A/B runs): uploads on
  // the high-pass stream instead, downloads alone on the copy stream.
  static const bool one_copy_stream = [] { const char *e = getenv("RNNOISE_AMD_HOSTIO_COPY"); return !e || !strcmp(e, "one"); }();
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
    (void)hipDeviceSynchronize();  // whatever was queued must not outlive the caller's buffers
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
}

extern "C" int rnnoise_batch_process(RNNoiseBatch *b, float *out, const float *in, float *vad, float *gains,
                                     int n_frames) {
  return batch_process_host_impl(b, out, in, vad, gains, n_frames, false);
}

extern "C" int rnnoise_batch_process_s16(RNNoiseBatch *b, short *out, const short *in, float *vad, float *gains,
                                         int n_frames) {
  return batch_process_host_impl(b, out, in, vad, gains, n_frames, true);
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

extern "C" long rnnoise_model_weight_bytes(RNNModel *model) {
  if (!model) return -1;
  std::lock_guard<std::mutex> lk(model->mu);
  if (model_parse_locked(model)) return -1;
  return model->weight_bytes;
}

// Serialise `model` (from a "DNNw" blob or from a pack) as an "RNPK" pack.  Returns the pack's size in bytes; the bytes
// are written only if cap is large enough (call with out == NULL to size the buffer).  -1 on a rejected model.  Host only.
extern "C" long rnnoise_amd_model_pack(RNNModel *model, void *out, long cap) {
  if (!model) return -1;
  std::lock_guard<std::mutex> lk(model->mu);
  if (model_parse_locked(model)) return -1;
  if (!model->staged) {
    model->staged = new StagedModel();
    stage_model(model->host, *model->staged);
  }
  const StagedModel &sm = *model->staged;
  const long total = (long)(sizeof(PackHeader) + sm.st.bytes.size());
  if (!out || cap < total) return total;
  PackHeader h;
  memset(&h, 0, sizeof h);
  memcpy(h.magic, "RNPK", 4);
  h.version = RN_PACK_VERSION;
  memcpy(h.dims, kPackDims, sizeof kPackDims);
  h.weight_bytes = model->weight_bytes;
  h.payload_bytes = sm.st.bytes.size();
  for (int i = 0; i < 10; i++) {
    const DevLinearOffsets &o = sm.off[i];
    PackLayer &l = h.layers[i];
    l.bias = o.bias; l.fw = o.fw; l.scale = o.scale; l.diag = o.diag; l.w = o.w; l.wmf = o.wmf; l.rowsum = o.rowsum;
    l.grp = o.grp; l.cols = o.cols;
    l.has_fw = o.has_fw; l.has_diag = o.has_diag; l.has_cols = o.has_cols; l.is_int8 = o.is_int8;
    l.nin = sm.lin[i].nin; l.nout = sm.lin[i].nout; l.nblocks = sm.lin[i].nblocks;
  }
  memcpy(out, &h, sizeof h);
  memcpy(static_cast<uint8_t *>(out) + sizeof h, sm.st.bytes.data(), sm.st.bytes.size());
  return total;
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

// out[i] = (float)log10(1e-2 + (double)ex[i]) evaluated on the device (host buffers; tests only)
extern "C" int rnnoise_amd_debug_log_energy(int device, float *out, const float *ex, int n) {
  if (!out || !ex || n <= 0) return -1;
  ON_DEVICE(device);
  float *d = nullptr;
  HIP_OK(hipMalloc((void **)&d, (size_t)n * 8));
  int rc = -1;
  if (hipMemcpy(d, ex, (size_t)n * 4, hipMemcpyHostToDevice) == hipSuccess &&
      rn_launch_log_energy(d, d + n, n, nullptr) == hipSuccess && hipStreamSynchronize(nullptr) == hipSuccess &&
      hipMemcpy(out, d + n, (size_t)n * 4, hipMemcpyDeviceToHost) == hipSuccess)
    rc = 0;
  hipFree(d);
  return rc;
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

// =============================================================================================
// drop-in rnnoise.h API (reference implementation: src/denoise.c:227-325,457-504)
// =============================================================================================
extern "C" RNNModel *rnnoise_model_from_buffer(const void *ptr, int len) {
  if (!ptr || len <= 0) return nullptr;
  RNNModel *m = new RNNModel();
  m->const_blob = ptr;
  m->blob_len = len;
  return m;
}

extern "C" RNNModel *rnnoise_model_from_file(FILE *f) {
  if (!f) return nullptr;
  if (fseek(f, 0, SEEK_END)) return nullptr;
  long len = ftell(f);
  if (len <= 0 || len > 0x7fffffffL || fseek(f, 0, SEEK_SET)) return nullptr;
  void *buf = malloc(len);
  if (!buf) return nullptr;
  if (fread(buf, len, 1, f) != 1) {
    free(buf);
    return nullptr;
  }
  RNNModel *m = new RNNModel();
  m->blob = buf;
  m->blob_len = (int)len;
  return m;
}

extern "C" RNNModel *rnnoise_model_from_filename(const char *filename) {
  FILE *f = filename ? fopen(filename, "rb") : nullptr;
  if (!f) return nullptr;  // the reference dereferences NULL here (denoise.c:246-248); we refuse instead
  RNNModel *m = rnnoise_model_from_file(f);
  if (!m) {
    fclose(f);
    return nullptr;
  }
  m->file = f;
  return m;
}

// ---------------------------------------------------------------------------------------------
// model == NULL: the reference falls back to its compiled-in weights (include/rnnoise.h:64-76,
// src/denoise.c:298-303).  Those are a separate download upstream (download_model.sh) and are not
// compiled in here either; the equivalent is a weight blob found at run time:
// $RNNOISE_AMD_DEFAULT_MODEL, else weights_blob.bin beside this library (the file name the
// reference's own dump_weights_blob writes).  Loaded once per process, never freed.
// ---------------------------------------------------------------------------------------------
static RNNModel *default_model() {
  static std::mutex mu;
  static RNNModel *model = nullptr;
  static bool tried = false;
  std::lock_guard<std::mutex> lk(mu);
  if (tried) return model;
  tried = true;
  std::string path;
  if (const char *e = getenv("RNNOISE_AMD_DEFAULT_MODEL")) path = e;
  else {
    Dl_info info;
    if (dladdr(reinterpret_cast<void *>(&default_model), &info) && info.dli_fname) {
      path = info.dli_fname;
      const size_t slash = path.rfind('/');
      path = (slash == std::string::npos ? std::string(".") : path.substr(0, slash)) + "/weights_blob.bin";
    }
  }
  if (!path.empty()) model = rnnoise_model_from_filename(path.c_str());
  if (!model)
    fprintf(stderr, "[rnnoise_amd] NULL model: no default weight blob (set RNNOISE_AMD_DEFAULT_MODEL or put weights_blob.bin "
                    "beside the library; tried '%s')\n", path.c_str());
  return model;
}

// ---------------------------------------------------------------------------------------------
// device-resident one-stream states (rnnoise_create / rnnoise_destroy)
// ---------------------------------------------------------------------------------------------
namespace {

StatePool *pool_new(RNNModel *model, int device) {
  StatePool *p = new StatePool();
  p->batch = rnnoise_batch_create(model, StatePool::POOL_SLOTS, device);
  if (!p->batch) {
    delete p;
    return nullptr;
  }
  DeviceGuard guard(device);
  if (!guard.ok || hipMalloc((void **)&p->d_io, (size_t)StatePool::POOL_SLOTS * 2 * 484 * sizeof(float)) != hipSuccess ||
      hipMalloc((void **)&p->d_flat, (size_t)StatePool::POOL_SLOTS * StatePool::FLAT_BLK * sizeof(float)) != hipSuccess) {
    if (p->d_io) hipFree(p->d_io);
    rnnoise_batch_destroy(p->batch);
    delete p;
    return nullptr;
  }
  return p;
}

// a free row of one of the model's pools on device 0 (a new pool when all are full); zeroed like rnnoise_init()
int pool_acquire(RNNModel *model, StatePool *&pool, int &slot) {
  std::unique_lock<std::mutex> lk(model->mu);
  for (int pass = 0; pass < 2; pass++) {
    for (StatePool *p : model->pools) {
      std::lock_guard<std::mutex> pl(p->mu);
      if (~p->used) {
        slot = __builtin_ctzll(~p->used);
        p->used |= 1ull << slot;
        pool = p;
        return 0;
      }
    }
    if (pass == 0) {
      lk.unlock();  // rnnoise_batch_create takes the model lock itself
      StatePool *p = pool_new(model, 0);
      lk.lock();
      if (!p) return -1;
      model->pools.push_back(p);
    }
  }
  return -1;
}

void pool_release(StatePool *p, int slot) {
  std::lock_guard<std::mutex> pl(p->mu);
  p->used &= ~(1ull << slot);
}

// the stateful arrays of one row back to all-zero (what rnnoise_init does to a DenoiseState, src/denoise.c:286)
int pool_zero_row(StatePool *p, int slot, hipStream_t st) {
  const RnGroupDev v = group_view(p->batch->g, slot, 1);
  const size_t N = p->batch->n;
  HIP_OK(hipMemsetAsync(v.mem_hp, 0, 2 * 4, st));
  HIP_OK(hipMemsetAsync(v.pitch_ring, 0, RN_RING_SIZE * 4, st));
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

}  // namespace

extern "C" void rnnoise_model_free(RNNModel *model) {
  if (!model) return;
  for (StatePool *p : model->pools) {
    {
      DeviceGuard guard(p->batch->device);
      hipFree(p->d_io);
      hipFree(p->d_flat);
    }
    rnnoise_batch_destroy(p->batch);
    delete p;
  }
  for (auto &d : model->dev) {
    DeviceGuard guard(d.device);
    hipFree(d.mem);
    hipFree(d.mem_rows);
  }
  if (model->file) fclose(model->file);
  delete model->staged;
  free(model->blob);
  delete model;
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
  if (!guard.ok || hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking) != hipSuccess ||
      hipHostMalloc((void **)&r.h_io, 2 * 484 * sizeof(float), hipHostMallocDefault) != hipSuccess ||
      pool_zero_row(r.pool, r.slot, r.stream) || hipStreamSynchronize(r.stream) != hipSuccess) {
    if (r.h_io) hipHostFree(r.h_io);
    if (r.stream) hipStreamDestroy(r.stream);
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
    {
      DeviceGuard guard(r.pool->batch->device);
      hipStreamSynchronize(r.stream);
      hipStreamDestroy(r.stream);
      hipHostFree(r.h_io);
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
    // device-resident state: four launches on the state's own stream
    PooledRef &r = st->ref;
    std::lock_guard<std::mutex> lk(*r.mu);
    DeviceGuard guard(r.pool->batch->device);
    if (!guard.ok) return frame_failed(out, "cannot select the HIP device");
    // The frame travels through the state's pinned block, which the kernels address directly (host memory mapped into the
    // device's address space: the first kernel reads its 1,920 bytes over PCIe, the last ones write frame and VAD back): no
    // copy commands, four launches and one wait per frame.  $RNNOISE_AMD_POOL_IO=copy stages through HBM instead (A/B runs).
    static const bool zero_copy = [] { const char *e = getenv("RNNOISE_AMD_POOL_IO"); return !e || strcmp(e, "copy"); }();
    float *h_in = r.h_io, *h_out = r.h_io + 484;
    memcpy(h_in, in, RN_FRAME_SIZE * sizeof(float));
    bool ok;
    if (zero_copy) {
      ok = pool_step(r.pool, r.slot, r.parity, r.ring_slot, r.frame_no, h_out, h_in, h_out + RN_FRAME_SIZE, r.stream) == 0 &&
           hipStreamSynchronize(r.stream) == hipSuccess;
    } else {
      float *d_in = r.pool->d_io + (size_t)r.slot * 2 * 484, *d_out = d_in + 484;
      ok = hipMemcpyAsync(d_in, h_in, RN_FRAME_SIZE * sizeof(float), hipMemcpyHostToDevice, r.stream) == hipSuccess &&
           pool_step(r.pool, r.slot, r.parity, r.ring_slot, r.frame_no, d_out, d_in, d_out + RN_FRAME_SIZE, r.stream) == 0 &&
           hipMemcpyAsync(h_out, d_out, (RN_FRAME_SIZE + 1) * sizeof(float), hipMemcpyDeviceToHost, r.stream) == hipSuccess &&
           hipStreamSynchronize(r.stream) == hipSuccess;
    }
    if (!ok) return frame_failed(out, "GPU step failed");
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
  if (pool_acquire(m, pool, slot)) return frame_failed(out, "no GPU state row available (no CPU fallback)");
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
