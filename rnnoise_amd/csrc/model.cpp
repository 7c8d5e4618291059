// model.cpp -- the model either side of the boundary: the reference's "DNNw" weight blob (src/nnet.h:41-62,
// src/parse_lpcnet_weights.c, src/write_weights.c), its re-layout for the GPU, the "RNPK" pack of that layout, and the
// rnnoise_model_* entry points of include/rnnoise.h.
#include "shim.h"

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
}  // namespace

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

}  // namespace
struct StagedModel {
  Staging st;
  DevLinearOffsets off[10];
  HostLinear lin[10];  // only nin / nout / nblocks are meaningful for a model that came from a pack
};
namespace {

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

}  // namespace

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
  if (hipMemcpy(d.mem, sm.st.bytes.data(), sm.st.bytes.size(), hipMemcpyHostToDevice) != hipSuccess) {
    fprintf(stderr, "[rnnoise_amd] cannot copy the model to device %d\n", device);
    (void)hipGetLastError();
    hipFree(d.mem);
    return -1;
  }
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
    if (hipMalloc(&d.mem_rows, rows.bytes.size()) != hipSuccess ||
        hipMemcpy(d.mem_rows, rows.bytes.data(), rows.bytes.size(), hipMemcpyHostToDevice) != hipSuccess) {
      fprintf(stderr, "[rnnoise_amd] cannot place the model's row-major copy on device %d\n", device);
      (void)hipGetLastError();
      if (d.mem_rows) hipFree(d.mem_rows);
      hipFree(d.mem);
      return -1;
    }
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
RNNModel *default_model() {
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
extern "C" void rnnoise_model_free(RNNModel *model) {
  if (!model) return;
  pools_free(model);
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

