// C-ABI (include/teramind_hip.h) + host-side executor of the UNet / sampler / gene-attention
// path.  The layer graph is the reference's `ours` model (model/unet_ours.py:82-426),
// re-scheduled for inference: time embedding and every ResBlock's emb_layers computed once
// per image, concat / collage / resample never materialised on their own (they are gather
// rules of the prep kernel), skip tensors never cloned, `o == 1` decoder pass optional.
#include "../../include/teramind_hip.h"
#include "tm_kernels.h"

#include <map>
#include <string>
#include <vector>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

using namespace tmk;

static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
// shared with tm_io.hip (same thread-local message buffer behind tm_last_error)
int tmk::io_fail(int code, const char* msg) { return fail(code, "%s", msg); }
#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) return fail(TM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

static const int RNA_TAIL[3] = {128, 64, 32};          // model/unet_ours.py:278-279
static inline bool is_h16(int dtype) { return dtype == TM_DTYPE_BF16 || dtype == TM_DTYPE_F16; }   // 16-bit operand modes

// ------------------------------------------------------------------------------------------
struct HostParam {
  std::vector<int64_t> shape;
  std::vector<float> data;
  bool loaded = false;
};

struct ResW {
  std::string pfx;
  std::vector<int> seg;       // real channels of each concat source
  int cin = 0, cbi = 0, cout = 0;
  bool has_skip = false;
  bool up = false;            // ResBlock(up=True): in_layers' conv also packed as phase weights of the upsampled-input form
  ConvW c1, c2, skip, c1u;
  const float* n1 = nullptr;  // [cbi*8] virtual order
  const float* n2 = nullptr;  // [cout]
  const uint16_t *c1h = nullptr, *c2h = nullptr;   // bf16 packed 3x3x3 weights (TM_DTYPE_BF16)
  const uint16_t* c1uh = nullptr;                  // up blocks: in_layers' conv as phase weights of the upsampled-input form
  const uint16_t* skiph = nullptr;
  int emb_off = 0;
};
struct AttnW {
  std::string pfx;
  int C = 0, G = 0;
  ConvW ada, q, kv, proj, fc1, fc2;
  const float *n1 = nullptr, *n2 = nullptr, *qn = nullptr, *kn = nullptr;
  const uint16_t *adah = nullptr, *qh = nullptr, *kvh = nullptr, *projh = nullptr, *fc1h = nullptr, *fc2h = nullptr;
};
struct DirectW {
  const float* w = nullptr;   // [tap][Cin][Cop]
  const float* bias = nullptr;
  int Cin = 0, Cout = 0, kz = 1, ky = 1, kx = 1;
};
struct Op {           // one entry of a TimestepEmbedSequential
  int kind;           // 0 res, 1 attn
  int idx;            // index into res / attn
  int mode;           // RS_*
};
struct EncEntry { int lvl; bool cat; std::vector<Op> ops; };
struct DecEntry { int lvl; std::vector<Op> ops; };

struct tm_model {
  tm_config cfg;
  int z = 0, gn = 0, D = 0, L = 4;
  int rw[4];                          // rna pyramid widths
  std::vector<std::pair<std::string, std::vector<int64_t>>> spec;
  std::map<std::string, HostParam> host;
  bool finalized = false;
  float* arena = nullptr;
  size_t arena_floats = 0;
  // graph
  std::vector<ResW> res;
  std::vector<AttnW> attn;
  std::vector<EncEntry> enc;
  std::vector<Op> mid;
  std::vector<DecEntry> dec;
  int emb_tot = 0;
  // packed pointers
  const float *te_w1 = nullptr, *te_b1 = nullptr, *te_w2 = nullptr, *te_b2 = nullptr;
  const float *emb_w = nullptr, *emb_b = nullptr;
  GeneW gene;
  ConvW downz, pyr[3];
  ConvW pyr16[3];                                    // 16-bit modes: the pyramid convs as 27-tap 16-bit convs (centre z slice)
  const uint16_t* pyrh[3] = {nullptr, nullptr, nullptr};
  DirectW stem, head, downz_d;       // downz_d: direct-conv form of down_z when the MFMA form does not apply
  bool downz_mfma = true;            // kz == 3 on a 4 x 4 (checkpoint config) or 8 x 8 gene grid
  bool gene_mfma = true;             // D == 64, G <= 232, no gene index table: fused MFMA gene-attention kernel
  const int* gene_idx = nullptr;     // device table: gene g reads slot gene_idx[g] (81-gene M2H subset) or null
  const float* out_norm = nullptr;
  // measurement hooks (tm_profile_*)
  bool prof_on = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_ev;
  size_t prof_used = 0;
  double prof_nominal = 0, prof_bytes = 0;
  struct ProfTag { int cin, cout, S, N; double nominal; };
  std::vector<ProfTag> prof_tag;     // per counted launch (TM_PROF_LAYERS)
};

// ------------------------------------------------------------------------------------------
// graph + key enumeration (mirrors the constructor, model/unet_ours.py:134-296)
// ------------------------------------------------------------------------------------------
typedef std::vector<std::pair<std::string, std::vector<int64_t>>> Spec;
static void K(Spec& s, const std::string& k, std::vector<int64_t> shp) { s.emplace_back(k, std::move(shp)); }

static void spec_res(Spec& s, const std::string& p, int cin, int cout, int E) {
  K(s, p + ".in_layers.0.weight", {1, cin, 1, 1});
  K(s, p + ".in_layers.2.weight", {cout, cin, 3, 3, 3});
  K(s, p + ".in_layers.2.bias", {cout});
  K(s, p + ".emb_layers.1.weight", {2 * cout, E});
  K(s, p + ".emb_layers.1.bias", {2 * cout});
  K(s, p + ".out_layers.0.weight", {1, cout, 1, 1});
  K(s, p + ".out_layers.3.weight", {cout, cout, 3, 3, 3});
  K(s, p + ".out_layers.3.bias", {cout});
  if (cin != cout) {
    K(s, p + ".skip_connection.weight", {cout, cin, 1, 1, 1});
    K(s, p + ".skip_connection.bias", {cout});
  }
}
static void spec_attn(Spec& s, const std::string& p, int c, int g) {
  K(s, p + ".norm1.weight", {c});
  for (const char* n : {"q", "k", "v"}) {
    K(s, p + ".attn." + n + ".weight", {c, c});
    K(s, p + ".attn." + n + ".bias", {c});
  }
  K(s, p + ".attn.q_norm.weight", {c});
  K(s, p + ".attn.k_norm.weight", {c});
  K(s, p + ".attn.proj.weight", {c, c});
  K(s, p + ".attn.proj.bias", {c});
  K(s, p + ".norm2.weight", {c});
  K(s, p + ".mlp.fc1.weight", {4 * c, c});
  K(s, p + ".mlp.fc1.bias", {4 * c});
  K(s, p + ".mlp.fc2.weight", {c, 4 * c});
  K(s, p + ".mlp.fc2.bias", {c});
  K(s, p + ".adaLN_modulation.1.weight", {7 * c, g});
  K(s, p + ".adaLN_modulation.1.bias", {7 * c});
}

static int add_res(tm_model* m, const std::string& pfx, std::vector<int> seg, int cout) {
  ResW r;
  r.pfx = pfx; r.seg = seg; r.cout = cout;
  for (int c : seg) { r.cin += c; r.cbi += (c + 7) / 8; }
  r.has_skip = r.cin != cout;
  r.emb_off = m->emb_tot;
  m->emb_tot += 2 * cout;
  m->res.push_back(r);
  return (int)m->res.size() - 1;
}
static int add_attn(tm_model* m, const std::string& pfx, int C, int G) {
  AttnW a;
  a.pfx = pfx; a.C = C; a.G = G;
  m->attn.push_back(a);
  return (int)m->attn.size() - 1;
}

static int build_graph(tm_model* m) {
  const tm_config& c = m->cfg;
  Spec& s = m->spec;
  const int E = c.embed_ch, ch0 = c.net_ch;
  K(s, "time_embed.time_embed.0.weight", {E, ch0});
  K(s, "time_embed.time_embed.0.bias", {E});
  K(s, "time_embed.time_embed.2.weight", {E, E});
  K(s, "time_embed.time_embed.2.bias", {E});
  const int d = m->D, g = c.rna_num;
  const int kzt[17] = {0, 1, 0, 0, 3, 0, 0, 0, 5, 0, 0, 0, 0, 0, 0, 0, 9};   // MBAblocks.py:472
  const int kz = kzt[c.rna_slc];
  const std::string gp = "rna_blocks.0.0";
  for (const char* n : {"q", "v"}) {
    K(s, gp + ".attn." + n + ".weight", {d, d});
    K(s, gp + ".attn." + n + ".bias", {d});
  }
  K(s, gp + ".attn.q_norm.weight", {d});
  K(s, gp + ".attn.proj.weight", {d, d});
  K(s, gp + ".attn.proj.bias", {d});
  K(s, gp + ".norm2.weight", {d});
  K(s, gp + ".mlp.fc1.weight", {4 * d, d});
  K(s, gp + ".mlp.fc1.bias", {4 * d});
  K(s, gp + ".mlp.fc2.weight", {d, 4 * d});
  K(s, gp + ".mlp.fc2.bias", {d});
  K(s, gp + ".down_z.weight", {g, g, kz, 3, 3});
  K(s, gp + ".down_z.bias", {g});
  if (c.vis_only) return TM_OK;
  for (int rid = 1; rid < 4; ++rid) {
    const std::string p = "rna_blocks." + std::to_string(rid) + ".1";
    K(s, p + ".weight", {m->rw[rid], m->rw[rid - 1], 1, 3, 3});
    K(s, p + ".bias", {m->rw[rid]});
  }
  K(s, "input_blocks.0.0.weight", {ch0, c.n_stain, 1, 3, 3});
  K(s, "input_blocks.0.0.bias", {ch0});
  const int L = m->L;
  int ch = ch0, res = c.patch_size, k = 1;
  std::vector<std::vector<int>> enc_ch(L);
  enc_ch[0].push_back(ch);
  for (int lvl = 0; lvl < L; ++lvl) {
    const int rd = m->rw[L - 1 - lvl];
    for (int i = 0; i < c.num_res_blocks; ++i) {
      const int cout = c.ch_mult[lvl] * ch0;
      const std::string p = "input_blocks." + std::to_string(k);
      EncEntry e; e.lvl = lvl; e.cat = true;
      spec_res(s, p + ".0", ch + rd, cout, E);
      e.ops.push_back({0, add_res(m, p + ".0", {ch, rd}, cout), RS_SAME});
      ch = cout;
      if (res == c.attn_res) {
        spec_attn(s, p + ".1", ch, rd);
        e.ops.push_back({1, add_attn(m, p + ".1", ch, rd), 0});
      }
      m->enc.push_back(e);
      enc_ch[lvl].push_back(ch);
      ++k;
    }
    if (lvl != L - 1) {
      res /= 2;
      const std::string p = "input_blocks." + std::to_string(k) + ".0";
      spec_res(s, p, ch, ch, E);
      EncEntry e; e.lvl = lvl + 1; e.cat = false;
      e.ops.push_back({0, add_res(m, p, {ch}, ch), RS_DOWN2});
      m->enc.push_back(e);
      enc_ch[lvl + 1].push_back(ch);
      ++k;
    }
  }
  spec_res(s, "middle_block.0", ch + m->rw[0], ch, E);
  m->mid.push_back({0, add_res(m, "middle_block.0", {ch, m->rw[0]}, ch), RS_SAME});
  spec_attn(s, "middle_block.1", ch, m->rw[0]);
  m->mid.push_back({1, add_attn(m, "middle_block.1", ch, m->rw[0]), 0});
  spec_res(s, "middle_block.2", ch, ch, E);
  m->mid.push_back({0, add_res(m, "middle_block.2", {ch}, ch), RS_SAME});
  k = 0;
  for (int lvl = L - 1; lvl >= 0; --lvl) {
    const int rd = m->rw[L - 1 - lvl];
    for (int i = 0; i < c.num_res_blocks + 1; ++i) {
      const int skip = enc_ch[lvl].back();
      enc_ch[lvl].pop_back();
      const int cout = c.ch_mult[lvl] * ch0;
      const std::string p = "output_blocks." + std::to_string(k);
      DecEntry e; e.lvl = lvl;
      spec_res(s, p + ".0", ch + skip + rd, cout, E);
      e.ops.push_back({0, add_res(m, p + ".0", {ch, skip, rd}, cout), RS_SAME});
      ch = cout;
      int nxt = 1;
      if (res == c.attn_res) {
        spec_attn(s, p + ".1", ch, rd);
        e.ops.push_back({1, add_attn(m, p + ".1", ch, rd), 0});
        nxt = 2;
      }
      if (lvl && i == c.num_res_blocks) {
        res *= 2;
        const std::string pu = p + "." + std::to_string(nxt);
        spec_res(s, pu, ch, ch, E);
        e.ops.push_back({0, add_res(m, pu, {ch}, ch), RS_UP2});
        m->res.back().up = true;
      }
      m->dec.push_back(e);
      ++k;
    }
  }
  K(s, "out.0.weight", {1, ch, 1, 1});
  K(s, "out.2.weight", {c.n_stain, ch0, 1, 3, 3});
  K(s, "out.2.bias", {c.n_stain});
  return TM_OK;
}

// ------------------------------------------------------------------------------------------
extern "C" int tm_version(void) { return TM_ABI_VERSION; }

extern "C" int tm_gene_tile_dense(const int32_t* crd, const void* dat, int64_t nnz, int gblk, int shift_h, int shift_w,
                                  int gsz, int chan_in, int zpad_ch, void* out, void* stream) {
  if (!out || nnz < 0 || (nnz > 0 && (!crd || !dat))) return fail(TM_ERR_ARG, "null / negative argument");
  if (gblk < 1 || gsz < 1 || chan_in < 1 || zpad_ch < 0) return fail(TM_ERR_ARG, "bad gene-tile geometry");
  HIP_TRY(launch_gene_tile_scatter(crd, (const float*)dat, (long)nnz, gblk, shift_h, shift_w, gsz, chan_in, zpad_ch,
                                   (float*)out, (hipStream_t)stream));
  return TM_OK;
}

extern "C" int tm_blosc_decompress(const void* src, size_t src_bytes, void* dst, size_t dst_cap, size_t* out_bytes) {
  return blosc_decompress(src, src_bytes, dst, dst_cap, out_bytes);
}
extern "C" const char* tm_last_error(void) { return g_err; }

extern "C" int tm_model_create(const tm_config* cfg, tm_model** out) {
  if (!cfg || !out) return fail(TM_ERR_ARG, "null argument");
  if (cfg->dtype != TM_DTYPE_F32 && !is_h16(cfg->dtype)) return fail(TM_ERR_ARG, "dtype must be TM_DTYPE_F32, TM_DTYPE_BF16 or TM_DTYPE_F16");
  if (cfg->patch_size != 32 && cfg->patch_size != 64 && cfg->patch_size != 128)
    return fail(TM_ERR_ARG, "patch_size must be 32, 64 or 128 (config_parm.py:47-55), got %d", cfg->patch_size);
  if (cfg->rna_slc != 1 && cfg->rna_slc != 4 && cfg->rna_slc != 8 && cfg->rna_slc != 16)
    return fail(TM_ERR_ARG, "rna_slc must be 1, 4, 8 or 16 (train.py:24-26), got %d", cfg->rna_slc);
  if (cfg->n_stain < 1 || cfg->n_stain > 2 || cfg->rna_num < 1 || cfg->rna_num > 512)
    return fail(TM_ERR_ARG, "n_stain/rna_num out of range");
  if ((cfg->patch_size / 16) * (cfg->patch_size / 16) * cfg->rna_slc > 512)
    return fail(TM_ERR_ARG, "gene-token width gn^2 * rna_slc = %d > 512 is not implemented (patch_size 128 with rna_slc 16)",
                (cfg->patch_size / 16) * (cfg->patch_size / 16) * cfg->rna_slc);
  if (cfg->rna_num == 81 && cfg->rna_slc != 1)
    return fail(TM_ERR_ARG, "the 81-gene human-brain subset requires rna_slc = 1 (model/unet_ours.py:313-316)");
  if (is_h16(cfg->dtype) && cfg->patch_size == 32)
    return fail(TM_ERR_ARG, "TM_DTYPE_BF16 / TM_DTYPE_F16 need patch_size 64 or 128 (no 4 x 4 tile form of the 16-bit conv)");
  if (cfg->net_ch % 64 || cfg->embed_ch % 64 || cfg->embed_ch > 1024)
    return fail(TM_ERR_ARG, "net_ch must be a multiple of 64, embed_ch a multiple of 64 <= 1024");
  tm_model* m = new tm_model();
  m->cfg = *cfg;
  m->z = (cfg->rna_slc + 1) / 2;
  m->gn = cfg->patch_size / 16;
  m->D = m->gn * m->gn * cfg->rna_slc;
  m->gene_mfma = m->D == 64 && cfg->rna_num <= 232 && cfg->rna_num != 81;
  m->downz_mfma = cfg->rna_slc == 4 && (m->gn == 4 || m->gn == 8);   // kz == 3 on a grid the MFMA tile forms cover
  m->rw[0] = cfg->rna_num;
  for (int i = 0; i < 3; ++i) m->rw[i + 1] = RNA_TAIL[i];
  int rc = build_graph(m);
  if (rc != TM_OK) { delete m; return rc; }
  for (auto& kv : m->spec) {
    HostParam hp; hp.shape = kv.second;
    m->host[kv.first] = hp;
  }
  *out = m;
  return TM_OK;
}

extern "C" int tm_model_num_params(const tm_model* m) { return m ? (int)m->spec.size() : 0; }
extern "C" const char* tm_model_param_key(const tm_model* m, int i) {
  if (!m || i < 0 || i >= (int)m->spec.size()) return nullptr;
  return m->spec[i].first.c_str();
}

extern "C" int tm_model_load_param(tm_model* m, const char* ref_key, const void* host_ptr, const int64_t* shape,
                                   int ndim, int dtype) {
  if (!m || !ref_key || !host_ptr || !shape) return fail(TM_ERR_ARG, "null argument");
  if (m->finalized) return fail(TM_ERR_STATE, "model already finalized");
  if (dtype != TM_DTYPE_F32) return fail(TM_ERR_ARG, "only fp32 parameters accepted");
  auto it = m->host.find(ref_key);
  if (it == m->host.end()) return fail(TM_ERR_KEY, "unexpected key '%s'", ref_key);
  HostParam& hp = it->second;
  if ((int)hp.shape.size() != ndim) return fail(TM_ERR_KEY, "key '%s': rank %d, expected %d", ref_key, ndim, (int)hp.shape.size());
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) {
    if (hp.shape[i] != shape[i]) return fail(TM_ERR_KEY, "key '%s': dim %d is %lld, expected %lld", ref_key, i,
                                             (long long)shape[i], (long long)hp.shape[i]);
    n *= (size_t)shape[i];
  }
  hp.data.assign((const float*)host_ptr, (const float*)host_ptr + n);
  hp.loaded = true;
  return TM_OK;
}

// ---- arena packing -------------------------------------------------------------------------
struct Packer {
  std::vector<float> buf;
  size_t reserve(size_t n) {                 // 64-byte aligned sub-allocations
    size_t off = (buf.size() + 15) / 16 * 16;
    buf.resize(off + n, 0.f);
    return off;
  }
};
struct Fix { const float** slot; size_t off; };
struct FixH { const uint16_t** slot; size_t off; };   // offsets in floats into the same arena

static const std::vector<float>& P(tm_model* m, const std::string& k) { return m->host[k].data; }

static void pack_conv(tm_model* m, Packer& pk, std::vector<Fix>& fx, ConvW& cw, const std::string& wkey,
                      const std::string& bkey, int Cout, const std::vector<int>& seg, int taps, bool centre_slice = false) {
  int cbi = 0, cin = 0;
  for (int c : seg) { cbi += (c + 7) / 8; cin += c; }
  // centre_slice: a 3x3x3 pad-1 conv applied to ONE plane (z_size 1, rna_slc 1) only ever multiplies its kz = 1 slice
  // with data -- pack those 9 taps and run the in-plane kernel
  std::vector<float> centre;
  const float* wsrc = P(m, wkey).data();
  if (centre_slice) {
    centre.resize((size_t)Cout * cin * 9);
    for (size_t i = 0; i < (size_t)Cout * cin; ++i)
      for (int t = 0; t < 9; ++t) centre[i * 9 + t] = wsrc[i * 27 + 9 + t];
    wsrc = centre.data();
    taps = 9;
  }
  cw.Cout = Cout; cw.Cbi = cbi; cw.taps = taps; cw.ntile = (Cout + 63) / 64;
  size_t off = pk.reserve(conv_pack_floats(Cout, cbi, taps));
  conv_pack_host(wsrc, Cout, seg.data(), (int)seg.size(), taps, pk.buf.data() + off);
  fx.push_back({&cw.w, off});
  size_t boff = pk.reserve((size_t)cw.ntile * 64);
  const std::vector<float>& b = P(m, bkey);
  for (int i = 0; i < Cout; ++i) pk.buf[boff + i] = b[i];
  fx.push_back({&cw.bias, boff});
}
// 3x3x3 conv for the bf16 path: bf16 packed weights (tm_conv_bf16.hip layout) + the fp32 bias
static void pack_conv_h(tm_model* m, Packer& pk, std::vector<Fix>& fx, std::vector<FixH>& fxh, ConvW& cw,
                        const uint16_t** wslot, const std::string& wkey, const std::string& bkey, int Cout,
                        const std::vector<int>& seg, bool inplane = false) {
  int cbi = 0, cin = 0;
  for (int c : seg) { cbi += (c + 7) / 8; cin += c; }
  // inplane: a Conv3d(k = (1,3,3), pad (0,1,1)) run by the 3x3x3 kernel -- its 9 taps sit in the centre z slice, the z - 1
  // and z + 1 slices are zero
  std::vector<float> w27;
  if (inplane) {
    const std::vector<float>& w9 = P(m, wkey);
    w27.assign((size_t)Cout * cin * 27, 0.f);
    for (size_t i = 0; i < (size_t)Cout * cin; ++i)
      for (int t = 0; t < 9; ++t) w27[i * 27 + 9 + t] = w9[i * 9 + t];
  }
  cw.Cout = Cout; cw.Cbi = cbi; cw.taps = 27; cw.ntile = (Cout + 63) / 64; cw.w = nullptr;
  const size_t elems = conv_bf16_pack_elems(Cout, cbi);
  size_t off = pk.reserve((elems + 1) / 2);
  (m->cfg.dtype == TM_DTYPE_F16 ? conv_f16_pack_host : conv_bf16_pack_host)(inplane ? w27.data() : P(m, wkey).data(), Cout, seg.data(),
                                                                            (int)seg.size(), (uint16_t*)(pk.buf.data() + off));
  fxh.push_back({wslot, off});
  size_t boff = pk.reserve((size_t)cw.ntile * 64);
  const std::vector<float>& b = P(m, bkey);
  for (int i = 0; i < Cout; ++i) pk.buf[boff + i] = b[i];
  fx.push_back({&cw.bias, boff});
}
// rows of several [rows_i][Cin] matrices stacked into one conv1 weight
static void pack_linear_stack(tm_model* m, Packer& pk, std::vector<Fix>& fx, ConvW& cw,
                              const std::vector<std::string>& pfx, int rows_each, int Cin) {
  const int Cout = rows_each * (int)pfx.size();
  std::vector<float> w((size_t)Cout * Cin), b(Cout);
  for (size_t i = 0; i < pfx.size(); ++i) {
    const std::vector<float>& wi = P(m, pfx[i] + ".weight");
    const std::vector<float>& bi = P(m, pfx[i] + ".bias");
    memcpy(w.data() + i * (size_t)rows_each * Cin, wi.data(), (size_t)rows_each * Cin * sizeof(float));
    memcpy(b.data() + i * rows_each, bi.data(), rows_each * sizeof(float));
  }
  int seg = Cin;
  cw.Cout = Cout; cw.Cbi = (Cin + 7) / 8; cw.taps = 1; cw.ntile = (Cout + 63) / 64;
  size_t off = pk.reserve(conv_pack_floats(Cout, cw.Cbi, 1));
  conv_pack_host(w.data(), Cout, &seg, 1, 1, pk.buf.data() + off);
  fx.push_back({&cw.w, off});
  size_t boff = pk.reserve((size_t)cw.ntile * 64);
  for (int i = 0; i < Cout; ++i) pk.buf[boff + i] = b[i];
  fx.push_back({&cw.bias, boff});
}
static void pack_linear_stack_h(tm_model* m, Packer& pk, std::vector<Fix>& fx, std::vector<FixH>& fxh, ConvW& cw,
                                const uint16_t** wslot, const std::vector<std::string>& pfx, int rows_each,
                                const std::vector<int>& seg, bool is_conv_key = false) {
  int Cin = 0, cbi = 0;
  for (int c : seg) { Cin += c; cbi += (c + 7) / 8; }
  const int Cout = rows_each * (int)pfx.size();
  std::vector<float> w((size_t)Cout * Cin), b(Cout);
  for (size_t i = 0; i < pfx.size(); ++i) {
    const std::vector<float>& wi = P(m, pfx[i] + ".weight");
    const std::vector<float>& bi = P(m, pfx[i] + ".bias");
    memcpy(w.data() + i * (size_t)rows_each * Cin, wi.data(), (size_t)rows_each * Cin * sizeof(float));
    memcpy(b.data() + i * rows_each, bi.data(), rows_each * sizeof(float));
  }
  (void)is_conv_key;
  cw.Cout = Cout; cw.Cbi = cbi; cw.taps = 1; cw.ntile = (Cout + 63) / 64; cw.w = nullptr;
  const size_t elems = conv1_bf16_pack_elems(Cout, cbi);
  size_t off = pk.reserve((elems + 1) / 2);
  (m->cfg.dtype == TM_DTYPE_F16 ? conv1_f16_pack_host : conv1_bf16_pack_host)(w.data(), Cout, seg.data(), (int)seg.size(),
                                                                              (uint16_t*)(pk.buf.data() + off));
  fxh.push_back({wslot, off});
  size_t boff = pk.reserve((size_t)cw.ntile * 64);
  for (int i = 0; i < Cout; ++i) pk.buf[boff + i] = b[i];
  fx.push_back({&cw.bias, boff});
}
static void pack_vec(Packer& pk, std::vector<Fix>& fx, const float** slot, const std::vector<float>& v,
                     const std::vector<int>& seg) {
  int cbi = 0;
  for (int c : seg) cbi += (c + 7) / 8;
  size_t off = pk.reserve((size_t)cbi * 8);
  vec_pack_host(v.data(), seg.data(), (int)seg.size(), pk.buf.data() + off);
  fx.push_back({slot, off});
}
static void pack_raw(Packer& pk, std::vector<Fix>& fx, const float** slot, const std::vector<float>& v) {
  size_t off = pk.reserve(v.size());
  memcpy(pk.buf.data() + off, v.data(), v.size() * sizeof(float));
  fx.push_back({slot, off});
}
static void pack_transposed(Packer& pk, std::vector<Fix>& fx, const float** slot, const std::vector<float>& w,
                            int rows, int cols) {           // [rows][cols] -> [cols][rows]
  size_t off = pk.reserve((size_t)rows * cols);
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c) pk.buf[off + (size_t)c * rows + r] = w[(size_t)r * cols + c];
  fx.push_back({slot, off});
}
static void pack_direct(tm_model* m, Packer& pk, std::vector<Fix>& fx, DirectW& dw, const std::string& pfx, int Cout,
                        int Cin, int kz, int ky, int kx) {
  dw.Cin = Cin; dw.Cout = Cout; dw.kz = kz; dw.ky = ky; dw.kx = kx;
  const int taps = kz * ky * kx, Cop = (Cout + 7) / 8 * 8;
  const std::vector<float>& w = P(m, pfx + ".weight");
  size_t off = pk.reserve((size_t)taps * Cin * Cop);
  for (int co = 0; co < Cout; ++co)
    for (int ci = 0; ci < Cin; ++ci)
      for (int t = 0; t < taps; ++t)
        pk.buf[off + ((size_t)t * Cin + ci) * Cop + co] = w[((size_t)co * Cin + ci) * taps + t];
  fx.push_back({&dw.w, off});
  pack_raw(pk, fx, &dw.bias, P(m, pfx + ".bias"));
}

extern "C" int tm_model_finalize(tm_model* m) {
  if (!m) return fail(TM_ERR_ARG, "null model");
  if (m->finalized) return fail(TM_ERR_STATE, "already finalized");
  for (auto& kv : m->spec)
    if (!m->host[kv.first].loaded) return fail(TM_ERR_KEY, "missing key '%s' (strict load)", kv.first.c_str());
  const tm_config& c = m->cfg;
  Packer pk;
  std::vector<Fix> fx;
  std::vector<FixH> fxh;
  const bool bf16 = is_h16(c.dtype);            // 16-bit operand mode (bf16 or fp16): same packing layout
  pack_raw(pk, fx, &m->te_w1, P(m, "time_embed.time_embed.0.weight"));
  pack_raw(pk, fx, &m->te_b1, P(m, "time_embed.time_embed.0.bias"));
  pack_raw(pk, fx, &m->te_w2, P(m, "time_embed.time_embed.2.weight"));
  pack_raw(pk, fx, &m->te_b2, P(m, "time_embed.time_embed.2.bias"));
  {
    const std::string g = "rna_blocks.0.0";
    const int d = m->D;
    pack_transposed(pk, fx, &m->gene.wq_t, P(m, g + ".attn.q.weight"), d, d);
    pack_raw(pk, fx, &m->gene.bq, P(m, g + ".attn.q.bias"));
    pack_transposed(pk, fx, &m->gene.wv_t, P(m, g + ".attn.v.weight"), d, d);
    pack_raw(pk, fx, &m->gene.bv, P(m, g + ".attn.v.bias"));
    pack_raw(pk, fx, &m->gene.qnorm, P(m, g + ".attn.q_norm.weight"));
    pack_transposed(pk, fx, &m->gene.wp_t, P(m, g + ".attn.proj.weight"), d, d);
    pack_raw(pk, fx, &m->gene.bp, P(m, g + ".attn.proj.bias"));
    pack_raw(pk, fx, &m->gene.norm2, P(m, g + ".norm2.weight"));
    pack_transposed(pk, fx, &m->gene.w1_t, P(m, g + ".mlp.fc1.weight"), 4 * d, d);
    pack_raw(pk, fx, &m->gene.b1, P(m, g + ".mlp.fc1.bias"));
    pack_transposed(pk, fx, &m->gene.w2_t, P(m, g + ".mlp.fc2.weight"), d, 4 * d);
    pack_raw(pk, fx, &m->gene.b2, P(m, g + ".mlp.fc2.bias"));
    if (m->downz_mfma) pack_conv(m, pk, fx, m->downz, g + ".down_z.weight", g + ".down_z.bias", c.rna_num, {c.rna_num}, 27);
    else {
      static const int kzt[17] = {0, 1, 0, 0, 3, 0, 0, 0, 5, 0, 0, 0, 0, 0, 0, 0, 9};            // MBAblocks.py:472
      pack_direct(m, pk, fx, m->downz_d, g + ".down_z", c.rna_num, c.rna_num, kzt[c.rna_slc], 3, 3);
    }
    if (c.rna_num == 81) {                      // human-brain generalisation: the 81 shared genes (utils/__init__.py:49-57)
      static const int M2H[81] = {1, 4, 5, 11, 21, 22, 23, 24, 25, 27, 35, 38, 40, 55, 56, 57, 61, 67, 69, 70, 75, 84, 90, 91, 96,
                                  108, 111, 113, 118, 130, 134, 137, 139, 145, 152, 155, 158, 165, 170, 171, 179, 180, 189, 191,
                                  206, 215, 223, 229, 230, 235, 241, 243, 253, 288, 297, 301, 309, 329, 337, 344, 346, 370, 372,
                                  378, 380, 395, 410, 436, 441, 442, 443, 458, 465, 467, 472, 478, 487, 492, 493, 494, 496};
      size_t off = pk.reserve(81);
      memcpy(pk.buf.data() + off, M2H, sizeof(M2H));
      fx.push_back({(const float**)&m->gene_idx, off});
    }
  }
  if (!c.vis_only) {
    for (int rid = 1; rid < 4 && !is_h16(c.dtype); ++rid)
      pack_conv(m, pk, fx, m->pyr[rid - 1], "rna_blocks." + std::to_string(rid) + ".1.weight",
                "rna_blocks." + std::to_string(rid) + ".1.bias", m->rw[rid], {m->rw[rid - 1]}, 9);
    if (is_h16(c.dtype))
      for (int rid = 1; rid < 4; ++rid)
        pack_conv_h(m, pk, fx, fxh, m->pyr16[rid - 1], &m->pyrh[rid - 1], "rna_blocks." + std::to_string(rid) + ".1.weight",
                    "rna_blocks." + std::to_string(rid) + ".1.bias", m->rw[rid], {m->rw[rid - 1]}, true);
    pack_direct(m, pk, fx, m->stem, "input_blocks.0.0", c.net_ch, c.n_stain, 1, 3, 3);
    pack_direct(m, pk, fx, m->head, "out.2", c.n_stain, c.net_ch, 1, 3, 3);
    pack_raw(pk, fx, &m->out_norm, P(m, "out.0.weight"));
    // all emb_layers as one [emb_tot][E] matrix
    {
      const int E = c.embed_ch;
      size_t woff = pk.reserve((size_t)m->emb_tot * E), boff = pk.reserve(m->emb_tot);
      for (ResW& r : m->res) {
        const std::vector<float>& w = P(m, r.pfx + ".emb_layers.1.weight");
        const std::vector<float>& b = P(m, r.pfx + ".emb_layers.1.bias");
        memcpy(pk.buf.data() + woff + (size_t)r.emb_off * E, w.data(), w.size() * sizeof(float));
        memcpy(pk.buf.data() + boff + r.emb_off, b.data(), b.size() * sizeof(float));
      }
      fx.push_back({&m->emb_w, woff});
      fx.push_back({&m->emb_b, boff});
    }
    for (ResW& r : m->res) {
      if (bf16) {
        pack_conv_h(m, pk, fx, fxh, r.c1, &r.c1h, r.pfx + ".in_layers.2.weight", r.pfx + ".in_layers.2.bias", r.cout, r.seg);
        pack_conv_h(m, pk, fx, fxh, r.c2, &r.c2h, r.pfx + ".out_layers.3.weight", r.pfx + ".out_layers.3.bias", r.cout, {r.cout});
        if (r.up && m->z == 2 && r.cout % 128 == 0) {          // phase weights of the upsampled-input form (bias: c1's)
          const size_t off = pk.reserve((conv_bf16_pack_ups_elems(r.cout, r.cbi) + 1) / 2);
          (c.dtype == TM_DTYPE_F16 ? conv_f16_pack_ups_host : conv_bf16_pack_ups_host)(
              P(m, r.pfx + ".in_layers.2.weight").data(), r.cout, r.seg.data(), (int)r.seg.size(), (uint16_t*)(pk.buf.data() + off));
          fxh.push_back({&r.c1uh, off});
        }
      } else {
        pack_conv(m, pk, fx, r.c1, r.pfx + ".in_layers.2.weight", r.pfx + ".in_layers.2.bias", r.cout, r.seg, 27, m->z == 1);
        if (r.up && m->z == 2) {                             // phase weights (conv3d_mfma UPS form) + their own copy of the bias
          r.c1u = r.c1;
          r.c1u.taps = 12;
          const size_t off = pk.reserve(conv_pack_ups_floats(r.cout, r.cbi));
          conv_pack_ups_host(P(m, r.pfx + ".in_layers.2.weight").data(), r.cout, r.seg.data(), (int)r.seg.size(), pk.buf.data() + off);
          fx.push_back({&r.c1u.w, off});
          const size_t boff = pk.reserve((size_t)r.c1u.ntile * 64);
          const std::vector<float>& b = P(m, r.pfx + ".in_layers.2.bias");
          for (int i = 0; i < r.cout; ++i) pk.buf[boff + i] = b[i];
          fx.push_back({&r.c1u.bias, boff});
        }
        pack_conv(m, pk, fx, r.c2, r.pfx + ".out_layers.3.weight", r.pfx + ".out_layers.3.bias", r.cout, {r.cout}, 27, m->z == 1);
      }
      if (r.has_skip) {
        if (bf16) pack_linear_stack_h(m, pk, fx, fxh, r.skip, &r.skiph, {r.pfx + ".skip_connection"}, r.cout, r.seg);
        else pack_conv(m, pk, fx, r.skip, r.pfx + ".skip_connection.weight", r.pfx + ".skip_connection.bias", r.cout, r.seg, 1);
      }
      pack_vec(pk, fx, &r.n1, P(m, r.pfx + ".in_layers.0.weight"), r.seg);
      pack_raw(pk, fx, &r.n2, P(m, r.pfx + ".out_layers.0.weight"));
    }
    for (AttnW& a : m->attn) {
      if (bf16) {
        pack_linear_stack_h(m, pk, fx, fxh, a.ada, &a.adah, {a.pfx + ".adaLN_modulation.1"}, 7 * a.C, {a.G});
        pack_linear_stack_h(m, pk, fx, fxh, a.q, &a.qh, {a.pfx + ".attn.q"}, a.C, {a.C});
        pack_linear_stack_h(m, pk, fx, fxh, a.kv, &a.kvh, {a.pfx + ".attn.k", a.pfx + ".attn.v"}, a.C, {a.C});
        pack_linear_stack_h(m, pk, fx, fxh, a.proj, &a.projh, {a.pfx + ".attn.proj"}, a.C, {a.C});
        pack_linear_stack_h(m, pk, fx, fxh, a.fc1, &a.fc1h, {a.pfx + ".mlp.fc1"}, 4 * a.C, {a.C});
        pack_linear_stack_h(m, pk, fx, fxh, a.fc2, &a.fc2h, {a.pfx + ".mlp.fc2"}, a.C, {4 * a.C});
        pack_raw(pk, fx, &a.n1, P(m, a.pfx + ".norm1.weight"));
        pack_raw(pk, fx, &a.n2, P(m, a.pfx + ".norm2.weight"));
        pack_raw(pk, fx, &a.qn, P(m, a.pfx + ".attn.q_norm.weight"));
        pack_raw(pk, fx, &a.kn, P(m, a.pfx + ".attn.k_norm.weight"));
        continue;
      }
      pack_linear_stack(m, pk, fx, a.ada, {a.pfx + ".adaLN_modulation.1"}, 7 * a.C, a.G);
      pack_linear_stack(m, pk, fx, a.q, {a.pfx + ".attn.q"}, a.C, a.C);
      pack_linear_stack(m, pk, fx, a.kv, {a.pfx + ".attn.k", a.pfx + ".attn.v"}, a.C, a.C);
      pack_linear_stack(m, pk, fx, a.proj, {a.pfx + ".attn.proj"}, a.C, a.C);
      pack_linear_stack(m, pk, fx, a.fc1, {a.pfx + ".mlp.fc1"}, 4 * a.C, a.C);
      pack_linear_stack(m, pk, fx, a.fc2, {a.pfx + ".mlp.fc2"}, a.C, 4 * a.C);
      pack_raw(pk, fx, &a.n1, P(m, a.pfx + ".norm1.weight"));
      pack_raw(pk, fx, &a.n2, P(m, a.pfx + ".norm2.weight"));
      pack_raw(pk, fx, &a.qn, P(m, a.pfx + ".attn.q_norm.weight"));
      pack_raw(pk, fx, &a.kn, P(m, a.pfx + ".attn.k_norm.weight"));
    }
  }
  m->arena_floats = (pk.buf.size() + 15) / 16 * 16;
  pk.buf.resize(m->arena_floats, 0.f);
  HIP_TRY(hipMalloc((void**)&m->arena, m->arena_floats * sizeof(float)));
  HIP_TRY(hipMemcpy(m->arena, pk.buf.data(), m->arena_floats * sizeof(float), hipMemcpyHostToDevice));
  for (Fix& f : fx) *f.slot = m->arena + f.off;
  for (FixH& f : fxh) *f.slot = (const uint16_t*)(m->arena + f.off);
  if (bf16) HIP_TRY(c.dtype == TM_DTYPE_F16 ? init_f16_device() : init_bf16_device());
  m->host.clear();
  m->finalized = true;
  return TM_OK;
}

extern "C" size_t tm_model_arena_bytes(const tm_model* m) { return m ? m->arena_floats * sizeof(float) : 0; }
extern "C" void* tm_model_arena_ptr(tm_model* m) { return m ? (void*)m->arena : nullptr; }

extern "C" int tm_profile_enable(tm_model* m, int on) {
  if (!m) return fail(TM_ERR_ARG, "null model");
  m->prof_on = on != 0;
  return TM_OK;
}
extern "C" int tm_profile_collect(tm_model* m, tm_prof_stats* out) {
  if (!m || !out) return fail(TM_ERR_ARG, "null argument");
  memset(out, 0, sizeof(*out));
  if (m->prof_used) HIP_TRY(hipEventSynchronize(m->prof_ev[m->prof_used - 1].second));
  // TM_PROF_LAYERS=<file>: one line per counted launch (diagnosis: which layers sit furthest below the kernel's average)
  static const char* layer_log = getenv("TM_PROF_LAYERS");
  FILE* lf = (layer_log && m->prof_tag.size() == m->prof_used) ? fopen(layer_log, "a") : nullptr;
  for (size_t i = 0; i < m->prof_used; ++i) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, m->prof_ev[i].first, m->prof_ev[i].second));
    out->total_ms += ms;
    if (lf) fprintf(lf, "%zu cin %d cout %d S %d N %d nominal_gflop %.3f ms %.4f\n", i, m->prof_tag[i].cin, m->prof_tag[i].cout,
                    m->prof_tag[i].S, m->prof_tag[i].N, m->prof_tag[i].nominal * 1e-9, ms);
  }
  if (lf) fclose(lf);
  m->prof_tag.clear();
  out->launches = m->prof_used;
  out->nominal_flops = m->prof_nominal;
  // Z == 2: the z-skip form issues 18 of 27 taps; Z == 1: the centre slice only (9); Z >= 3: all 27 (zero planes staged)
  // (the 16-bit conv never stages z-padding planes: (3Z - 2) / 3Z of the taps for any Z)
  out->executed_flops = m->prof_nominal * (is_h16(m->cfg.dtype) ? (3.0 * m->z - 2.0) / (3.0 * m->z)
                                                                 : (m->z == 2 ? 18.0 / 27.0 : (m->z == 1 ? 9.0 / 27.0 : 1.0)));
  out->alg_bytes = m->prof_bytes;
  m->prof_used = 0; m->prof_nominal = 0; m->prof_bytes = 0;
  return TM_OK;
}

extern "C" int tm_model_destroy(tm_model* m) {
  if (!m) return TM_OK;
  for (auto& e : m->prof_ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  if (m->arena) (void)hipFree(m->arena);
  delete m;
  return TM_OK;
}

// ------------------------------------------------------------------------------------------
// executor
// ------------------------------------------------------------------------------------------
struct Ctx {
  tm_model* m;
  char* base = nullptr;
  size_t cap = 0, top = 0, peak = 0;
  bool dry = false;
  hipStream_t s = nullptr;
  int b = 0, p1 = 0, p2 = 0, Ne = 0, Nd = 0;
  hipError_t err = hipSuccess;
  const float* ss = nullptr;      // [b][emb_tot]

  float* alloc_f(size_t nfloats) {
    size_t off = (top + 255) / 256 * 256;
    top = off + nfloats * sizeof(float);
    if (top > peak) peak = top;
    if (dry) return (float*)(uintptr_t)256;     // never dereferenced
    if (top > cap) { if (err == hipSuccess) err = hipErrorOutOfMemory; return (float*)base; }
    return (float*)(base + off);
  }
  TV tensor(int N, int C, int Z, int S) {
    TV t;
    t.N = N; t.C = C; t.Cb = (C + 7) / 8; t.Z = Z; t.H = S; t.W = S;
    t.nstride = (long)t.Cb * t.plane();
    t.p = alloc_f((size_t)N * t.nstride);
    return t;
  }
  TVH tensor_h(int N, int Cb_even, int Z, int S) {           // bf16 CB8 (conv27_bf16 input)
    TVH t;
    t.N = N; t.Cb = Cb_even; t.C = Cb_even * 8; t.Z = Z; t.H = S; t.W = S;
    t.nstride = (long)Cb_even * Z * S * S * 8;
    t.p = (uint16_t*)alloc_f(((size_t)N * t.nstride + 1) / 2);
    return t;
  }
  // Activation-stream tensor (block outputs, skips, RNA pyramid levels): fp32 CB8 in TM_DTYPE_F32; in the 16-bit modes
  // the SAME geometry with 16-bit elements (the TV's `p` then points at uint16_t data and `nstride` counts elements).
  TV tensor_s(int N, int C, int Z, int S) {
    if (!is_h16(m->cfg.dtype)) return tensor(N, C, Z, S);
    TV t;
    t.N = N; t.C = C; t.Cb = (C + 7) / 8; t.Z = Z; t.H = S; t.W = S;
    t.nstride = (long)t.Cb * t.plane();
    t.p = alloc_f(((size_t)N * t.nstride + 1) / 2);
    return t;
  }
  void check(hipError_t e) { if (e != hipSuccess && err == hipSuccess) err = e; }
};
struct Src { TV t; bool collage; };

// 16-bit view of a stream tensor (16-bit modes only)
static TVH as_h(const TV& t) {
  TVH v;
  v.p = (uint16_t*)t.p; v.N = t.N; v.C = t.C; v.Cb = t.Cb; v.Z = t.Z; v.H = t.H; v.W = t.W; v.nstride = t.nstride;
  return v;
}


// Debug taps: with TM_DEBUG_DIR set, every block output is written (NCDHW fp32, raw) to
// $TM_DEBUG_DIR/<name>.bin after a stream sync.  Test/diagnostic aid only.
static const char* debug_dir() {
  static const char* d = getenv("TM_DEBUG_DIR");
  return (d && *d) ? d : nullptr;
}
static void dump_tv(Ctx& cx, const std::string& name, const TV& t_in) {
  if (cx.dry || !debug_dir()) return;
  TV t = t_in;
  const size_t n = (size_t)t.N * t.C * t.Z * t.H * t.W;
  float* dev = nullptr;
  float* f32copy = nullptr;
  if (hipMalloc((void**)&dev, n * sizeof(float)) != hipSuccess) return;
  if (is_h16(cx.m->cfg.dtype)) {              // 16-bit stream tensor -> fp32 CB8 copy (prep kernel, no norm / act)
    if (hipMalloc((void**)&f32copy, (size_t)t.N * t.nstride * sizeof(float)) != hipSuccess) { (void)hipFree(dev); return; }
    PrepLaunch P;
    P.nsrc = 1;
    P.src[0].p = t.p; P.src[0].nstride = t.nstride; P.src[0].Cb = t.Cb;
    P.N = t.N; P.Z = t.Z; P.S = t.H; P.src_h = 1; P.h_f16 = cx.m->cfg.dtype == TM_DTYPE_F16;
    P.out = f32copy; P.out_nstride = t.nstride;
    if (launch_prep(P, cx.s) != hipSuccess) { (void)hipFree(dev); (void)hipFree(f32copy); return; }
    t.p = f32copy;
  }
  std::vector<float> host(n);
  if (launch_from_cb8(t, dev, cx.s) == hipSuccess && hipStreamSynchronize(cx.s) == hipSuccess &&
      hipMemcpy(host.data(), dev, n * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess) {
    const std::string path = std::string(debug_dir()) + "/" + name + ".bin";
    if (FILE* f = fopen(path.c_str(), "wb")) { fwrite(host.data(), sizeof(float), n, f); fclose(f); }
  }
  (void)hipFree(dev);
  if (f32copy) (void)hipFree(f32copy);
}

static void run_conv(Ctx& cx, const TV& x, const ConvW& w, TV y, const TV* res, const TV* gate, int flags,
                     int cin_real = 0, int zmode = ZM_PAD1, bool gate_half = false, bool res_half = false) {
  if (cx.dry) return;
  ConvLaunch L;
  L.x = x; L.w = w; L.y = y; L.res = res; L.gate = gate; L.flags = flags; L.zmode = zmode; L.gate_half = gate_half ? 1 : 0;
  L.res_half = res_half ? 1 : 0;
  tm_model* m = cx.m;
  const bool prof = m->prof_on && (w.taps == 27 || (m->z == 1 && w.taps == 9)) && zmode == ZM_PAD1;
  if (prof) {
    if (m->prof_used == m->prof_ev.size()) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { cx.check(hipErrorOutOfMemory); return; }
      m->prof_ev.emplace_back(a, b);
    }
    cx.check(hipEventRecord(m->prof_ev[m->prof_used].first, cx.s));
  }
  cx.check(launch_conv_mfma(L, cx.s));
  if (prof) {
    cx.check(hipEventRecord(m->prof_ev[m->prof_used].second, cx.s));
    m->prof_used++;
    const double vox = (double)x.N * x.Z * x.H * x.W;
    m->prof_nominal += 2.0 * (cin_real ? cin_real : x.C) * w.Cout * 27.0 * vox;
    m->prof_tag.push_back({cin_real ? cin_real : x.C, w.Cout, x.H, x.N, 2.0 * (cin_real ? cin_real : x.C) * w.Cout * 27.0 * vox});
    m->prof_bytes += 4.0 * (vox * x.Cb * 8 + (double)w.ntile * w.Cbi * 27 * 512 + vox * y.Cb * 8);
  }
}

// y: geometry of the output; y16: `y.p` is a 16-bit stream tensor (written as such); res16: 16-bit stream residual
static void run_conv_h(Ctx& cx, const TVH& x, const uint16_t* w, const ConvW& cw, TV y, const TV* res16, int cin_real,
                       const TVH* fuse_a2 = nullptr, const ResW* rw = nullptr, int per_image = 1, bool y16 = false,
                       bool count = true, bool ups = false, bool res_half = false) {
  if (cx.dry) return;
  ConvLaunchH L;
  L.x = x; L.w = w; L.bias = cw.bias; L.Cout = cw.Cout; L.y = y; L.ups = ups ? 1 : 0; L.res_half = res_half ? 1 : 0;
  TVH resh;
  if (res16) { resh = as_h(*res16); L.res_h = &resh; }
  if (y16) { L.y_h = (uint16_t*)y.p; L.yh_nstride = y.nstride; }
  tm_model* m = cx.m;
  if (fuse_a2) {
    L.fuse_norm = 1; L.a2 = *fuse_a2; L.norm_w = rw->n2; L.per_image = per_image;
    L.mod_scale = cx.ss + rw->emb_off; L.mod_shift = cx.ss + rw->emb_off + rw->cout; L.mod_stride = m->emb_tot;
  }
  const bool prof = m->prof_on && count;    // count = false: not a ResBlock 3x3x3 conv (the in-plane pyramid convs: 9 real taps)
  if (prof) {
    if (m->prof_used == m->prof_ev.size()) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { cx.check(hipErrorOutOfMemory); return; }
      m->prof_ev.emplace_back(a, b);
    }
    cx.check(hipEventRecord(m->prof_ev[m->prof_used].first, cx.s));
  }
  cx.check(m->cfg.dtype == TM_DTYPE_F16 ? launch_conv27_f16(L, cx.s) : launch_conv27_bf16(L, cx.s));
  if (prof) {
    cx.check(hipEventRecord(m->prof_ev[m->prof_used].second, cx.s));
    m->prof_used++;
    const double vox = (double)x.N * x.Z * x.H * x.W;
    m->prof_nominal += 2.0 * cin_real * cw.Cout * 27.0 * vox;
    m->prof_tag.push_back({cin_real, cw.Cout, x.H, x.N, 2.0 * cin_real * cw.Cout * 27.0 * vox});
    m->prof_bytes += 2.0 * vox * x.Cb * 8 + 2.0 * (double)conv_bf16_pack_elems(cw.Cout, cw.Cbi) +
                     ((y16 || fuse_a2) ? 2.0 : 4.0) * vox * y.Cb * 8 + (res16 ? 2.0 * vox * y.Cb * 8 : 0.0);
  }
}

static void run_conv1_h(Ctx& cx, const TVH& x, const uint16_t* w, const ConvW& cw, TV y, const TV* res, const TV* gate,
                        int flags, TVH* y_h = nullptr, const TVH* gate_h = nullptr, const TVH* res_h = nullptr,
                        const std::vector<Src>* concat = nullptr, bool gate_half = false) {
  if (cx.dry) return;
  ConvLaunchH L;
  L.x = x; L.w = w; L.bias = cw.bias; L.Cout = cw.Cout; L.y = y; L.res = res; L.gate = gate; L.flags = flags;
  L.gate_h = gate_h; L.res_h = res_h; L.gate_half = gate_half ? 1 : 0;
  if (concat) {                       // x = th.cat(sources, 1) (+ to_collage) read in place; `x` carries the geometry only
    L.nsrc = (int)concat->size();
    for (int i = 0; i < L.nsrc; ++i) { L.xs[i] = as_h((*concat)[i].t); L.xs_collage[i] = (*concat)[i].collage ? 1 : 0; }
    L.p1 = cx.p1; L.p2 = cx.p2;
  }
  if (y_h) { L.y_h = y_h->p; L.yh_nstride = y_h->nstride; }
  cx.check(cx.m->cfg.dtype == TM_DTYPE_F16 ? launch_conv1_f16(L, cx.s) : launch_conv1_bf16(L, cx.s));
}

// ResBlock._forward (model/MBAblocks.py:237-299) in the 16-bit modes: every tensor that crosses a block boundary (sources,
// output, the resampled / concatenated x that feeds the skip conv or the residual add) is a 16-bit CB8 stream tensor --
// as under the reference's fp16 autocast, where every conv output and the residual sum are half tensors
// (diffusion/base.py:377).  Norm statistics, modulation, SiLU and all accumulation stay fp32.
static TV res_block_h16(Ctx& cx, const ResW& w, const std::vector<Src>& src, int N, int per_image, int S_out, int mode) {
  tm_model* m = cx.m;
  const int Z = m->z;
  const int h_f16 = m->cfg.dtype == TM_DTYPE_F16;
  TV out = cx.tensor_s(N, w.cout, Z, S_out);
  const size_t mark = cx.top;
  const int cbe = (w.cbi + 1) / 2 * 2;
  static const bool no_ups = getenv("TM_CONV_UPS") && atoi(getenv("TM_CONV_UPS")) == 0;           // A/B timing only
  if (mode == RS_UP2 && w.c1uh && !no_ups && src.size() == 1 && !src[0].collage && !w.has_skip && !(S_out & (S_out - 1)) &&
      S_out >= 16) {
    // ResBlock(up=True) (MBAblocks.py:254-261,297) without upsampling anything: norm + SiLU on the low-resolution x, the first
    // conv in its upsampled-input form (per-phase 2 x 2 in-plane weights: 8 instead of 18 taps), the second conv's epilogue
    // reads the residual Upsample(x) at (z, y >> 1, x >> 1) of x itself (see the fp32 twin in res_block)
    const int S_in = S_out / 2;
    TVH Al = cx.tensor_h(N, cbe, Z, S_in);
    if (!cx.dry) {
      PrepLaunch P;
      P.nsrc = 1; P.src_h = 1; P.h_f16 = h_f16;
      P.src[0].p = src[0].t.p; P.src[0].nstride = src[0].t.nstride; P.src[0].Cb = src[0].t.Cb;
      P.N = N; P.Z = Z; P.S = S_in; P.norm_w = w.n1; P.inv_c = 1.0f / (float)w.cin; P.act = 1; P.per_image = per_image;
      P.out_h = Al.p; P.out_h_nstride = Al.nstride; P.pad_blocks = Al.Cb - w.cbi;
      cx.check(launch_prep(P, cx.s));
    }
    const bool fuse_up = w.cout == 128;
    TVH A2u = cx.tensor_h(N, (w.cout / 8 + 1) / 2 * 2, Z, S_out);
    TV geom_u; geom_u.N = N; geom_u.C = w.cout; geom_u.Cb = w.cout / 8; geom_u.Z = Z; geom_u.H = S_out; geom_u.W = S_out;
    geom_u.nstride = (long)geom_u.Cb * geom_u.plane();
    if (fuse_up) {
      run_conv_h(cx, Al, w.c1uh, w.c1, geom_u, nullptr, w.cin, &A2u, &w, per_image, false, false, true);
    } else {
      TV H1 = cx.tensor_s(N, w.cout, Z, S_out);
      run_conv_h(cx, Al, w.c1uh, w.c1, H1, nullptr, w.cin, nullptr, nullptr, 1, true, false, true);
      if (!cx.dry) {
        PrepLaunch P;
        P.nsrc = 1;
        P.src[0].p = H1.p; P.src[0].nstride = H1.nstride; P.src[0].Cb = H1.Cb;
        P.src_h = 1;
        P.N = N; P.Z = Z; P.S = S_out;
        P.norm_w = w.n2; P.inv_c = 1.0f / (float)w.cout; P.act = 1; P.per_image = per_image;
        P.mod = MOD_IMAGE; P.mod_scale = cx.ss + w.emb_off; P.mod_shift = cx.ss + w.emb_off + w.cout;
        P.mod_stride = m->emb_tot;
        P.out_h = A2u.p; P.out_h_nstride = A2u.nstride; P.pad_blocks = A2u.Cb - w.cout / 8; P.h_f16 = h_f16;
        cx.check(launch_prep(P, cx.s));
      }
    }
    run_conv_h(cx, A2u, w.c2h, w.c2, out, &src[0].t, w.cout, nullptr, nullptr, 1, true, true, false, true);
    cx.top = mark;
    return out;
  }
  TVH Ah = cx.tensor_h(N, cbe, Z, S_out), rawh;
  // The concatenated / re-tiled x (MBAblocks.py:252-258,297) is only materialised where it is the RESIDUAL of a block
  // without skip conv (the up / down blocks: resampled x); the skip conv reads the sources in place (conv1's concat input)
  static const bool ab_raw = getenv("TM_AB_RAW") != nullptr;       // A/B switch: materialise the concat for the skip conv too
  const bool need_raw = (ab_raw || !w.has_skip) && (w.has_skip || mode != RS_SAME || src.size() > 1 || src[0].collage);
  if (w.has_skip && mode != RS_SAME) { cx.check(hipErrorInvalidValue); return out; }    // never in this model family
  if (need_raw) rawh = cx.tensor_h(N, cbe, Z, S_out);
  if (!cx.dry) {
    PrepLaunch P;
    P.nsrc = (int)src.size();
    for (size_t i = 0; i < src.size(); ++i) {
      P.src[i].p = src[i].t.p; P.src[i].nstride = src[i].t.nstride; P.src[i].Cb = src[i].t.Cb;
      P.src[i].collage = src[i].collage ? 1 : 0;
    }
    P.src_h = 1; P.h_f16 = h_f16;
    P.resample = mode; P.N = N; P.Z = Z; P.S = S_out; P.p1 = cx.p1; P.p2 = cx.p2;
    P.norm_w = w.n1; P.inv_c = 1.0f / (float)w.cin; P.act = 1; P.per_image = per_image;
    P.out_h = Ah.p; P.out_h_nstride = Ah.nstride; P.pad_blocks = Ah.Cb - w.cbi;
    if (need_raw) { P.raw_h = rawh.p; P.raw_h_nstride = rawh.nstride; }
    cx.check(launch_prep(P, cx.s));
  }
  // Cout in {64, 128}: one workgroup holds every cout of its voxels, so out_layers' norm -> modulate -> SiLU runs in the
  // first conv's epilogue and writes the second conv's 16-bit input directly
  const bool fuse_mid = w.cout == 64 || w.cout == 128;
  TVH A2h = cx.tensor_h(N, (w.cout / 8 + 1) / 2 * 2, Z, S_out);
  TV geom; geom.N = N; geom.C = w.cout; geom.Cb = w.cout / 8; geom.Z = Z; geom.H = S_out; geom.W = S_out;
  geom.nstride = (long)geom.Cb * geom.plane();
  if (fuse_mid) {
    run_conv_h(cx, Ah, w.c1h, w.c1, geom, nullptr, w.cin, &A2h, &w, per_image);
  } else {
    // Cout > 128 (several cout tiles per voxel): the conv output goes through a 16-bit tensor like every other conv
    // output of the 16-bit modes -- under the reference's autocast in_layers' Conv3d returns fp16 and out_layers'
    // LlamaRMSNorm takes its statistics from those half values (MBAblocks.py:21-43 casts the half input up)
    TV H1 = cx.tensor_s(N, w.cout, Z, S_out);
    run_conv_h(cx, Ah, w.c1h, w.c1, H1, nullptr, w.cin, nullptr, nullptr, 1, true);
    if (!cx.dry) {
      PrepLaunch P;
      P.nsrc = 1;
      P.src[0].p = H1.p; P.src[0].nstride = H1.nstride; P.src[0].Cb = H1.Cb;
      P.src_h = 1;
      P.N = N; P.Z = Z; P.S = S_out;
      P.norm_w = w.n2; P.inv_c = 1.0f / (float)w.cout; P.act = 1; P.per_image = per_image;
      P.mod = MOD_IMAGE; P.mod_scale = cx.ss + w.emb_off; P.mod_shift = cx.ss + w.emb_off + w.cout;
      P.mod_stride = m->emb_tot;
      P.out_h = A2h.p; P.out_h_nstride = A2h.nstride; P.pad_blocks = A2h.Cb - w.cout / 8; P.h_f16 = h_f16;
      cx.check(launch_prep(P, cx.s));
    }
  }
  TV rawv = geom; rawv.p = (float*)rawh.p; rawv.nstride = rawh.nstride;   // the residual view of rawh (channels < cout only)
  const TV* r = nullptr;
  if (w.has_skip) {
    TVH outh = as_h(out), xg = Ah;                       // geometry + padded block count of the virtual concat
    xg.p = nullptr;
    if (ab_raw) run_conv1_h(cx, rawh, w.skiph, w.skip, out, nullptr, nullptr, 0, &outh);
    else run_conv1_h(cx, xg, w.skiph, w.skip, out, nullptr, nullptr, 0, &outh, nullptr, nullptr, &src);
    r = &out;
  } else if (need_raw) r = &rawv;
  else r = &src[0].t;
  run_conv_h(cx, A2h, w.c2h, w.c2, out, r, w.cout, nullptr, nullptr, 1, true);
  cx.top = mark;
  return out;
}

// ResBlock._forward (model/MBAblocks.py:237-299)
static TV res_block(Ctx& cx, const ResW& w, const std::vector<Src>& src, int N, int per_image, int S_out, int mode,
                    TV* out_opt) {
  tm_model* m = cx.m;
  if (is_h16(m->cfg.dtype)) return res_block_h16(cx, w, src, N, per_image, S_out, mode);
  const int Z = m->z;
  TV out = out_opt ? *out_opt : cx.tensor(N, w.cout, Z, S_out);
  const size_t mark = cx.top;
  static const bool no_ups = getenv("TM_CONV_UPS") && atoi(getenv("TM_CONV_UPS")) == 0;           // A/B timing only
  if (mode == RS_UP2 && w.c1u.w && !no_ups && src.size() == 1 && !src[0].collage && !w.has_skip && !(S_out & (S_out - 1))) {
    // ResBlock(up=True) (MBAblocks.py:254-261,297): h = Upsample(x) feeds in_layers, x = Upsample(x) is the residual.  Nothing
    // is upsampled here: norm and SiLU act per voxel, so they run on the low-resolution x (a quarter of the bytes); the first
    // conv is the upsampled-input form of conv3d_mfma (per-phase 2 x 2 in-plane weights, 8 instead of 18 taps); the second
    // conv's epilogue reads the residual at (z, y >> 1, x >> 1) of x itself.
    const int S_in = S_out / 2;
    TV A = cx.tensor(N, w.cbi * 8, Z, S_in);
    if (!cx.dry) {
      PrepLaunch P;
      P.nsrc = 1;
      P.src[0].p = src[0].t.p; P.src[0].nstride = src[0].t.nstride; P.src[0].Cb = src[0].t.Cb;
      P.N = N; P.Z = Z; P.S = S_in; P.norm_w = w.n1; P.inv_c = 1.0f / (float)w.cin; P.act = 1; P.per_image = per_image;
      P.out = A.p; P.out_nstride = A.nstride;
      cx.check(launch_prep(P, cx.s));
    }
    TV A2 = cx.tensor(N, w.cout, Z, S_out), H1 = cx.tensor(N, w.cout, Z, S_out);
    run_conv(cx, A, w.c1u, H1, nullptr, nullptr, 0, w.cin, ZM_UPS);
    if (!cx.dry) {
      PrepLaunch P;
      P.nsrc = 1;
      P.src[0].p = H1.p; P.src[0].nstride = H1.nstride; P.src[0].Cb = H1.Cb;
      P.N = N; P.Z = Z; P.S = S_out;
      P.norm_w = w.n2; P.inv_c = 1.0f / (float)w.cout; P.act = 1; P.per_image = per_image;
      P.mod = MOD_IMAGE; P.mod_scale = cx.ss + w.emb_off; P.mod_shift = cx.ss + w.emb_off + w.cout;
      P.mod_stride = m->emb_tot;
      P.out = A2.p; P.out_nstride = A2.nstride;
      cx.check(launch_prep(P, cx.s));
    }
    run_conv(cx, A2, w.c2, out, &src[0].t, nullptr, 0, 0, ZM_PAD1, false, true);
    cx.top = mark;
    return out;
  }
  int cin_pad = w.cbi * 8;
  TV A = cx.tensor(N, cin_pad, Z, S_out), raw, H1, A2;
  // the residual / skip-conv input is the CONCATENATED (and resampled) x, MBAblocks.py:252-258,297:
  // it equals a stored tensor only for a single plain source
  const bool need_raw = w.has_skip || mode != RS_SAME || src.size() > 1 || src[0].collage;
  if (need_raw) raw = cx.tensor(N, cin_pad, Z, S_out);
  if (!cx.dry) {
    PrepLaunch P;
    P.nsrc = (int)src.size();
    for (size_t i = 0; i < src.size(); ++i) {
      P.src[i].p = src[i].t.p; P.src[i].nstride = src[i].t.nstride; P.src[i].Cb = src[i].t.Cb;
      P.src[i].collage = src[i].collage ? 1 : 0;
    }
    P.resample = mode; P.N = N; P.Z = Z; P.S = S_out; P.p1 = cx.p1; P.p2 = cx.p2;
    P.norm_w = w.n1; P.inv_c = 1.0f / (float)w.cin; P.act = 1; P.per_image = per_image;
    P.out = A.p; P.out_nstride = A.nstride;
    if (need_raw) { P.raw = raw.p; P.raw_nstride = raw.nstride; }
    cx.check(launch_prep(P, cx.s));
  }
  A2 = cx.tensor(N, w.cout, Z, S_out);
  H1 = cx.tensor(N, w.cout, Z, S_out);
  run_conv(cx, A, w.c1, H1, nullptr, nullptr, 0, w.cin);
  if (!cx.dry) {
    PrepLaunch P;
    P.nsrc = 1;
    P.src[0].p = H1.p; P.src[0].nstride = H1.nstride; P.src[0].Cb = H1.Cb;
    P.N = N; P.Z = Z; P.S = S_out;
    P.norm_w = w.n2; P.inv_c = 1.0f / (float)w.cout; P.act = 1; P.per_image = per_image;
    P.mod = MOD_IMAGE; P.mod_scale = cx.ss + w.emb_off; P.mod_shift = cx.ss + w.emb_off + w.cout;
    P.mod_stride = m->emb_tot;
    P.out = A2.p; P.out_nstride = A2.nstride;
    cx.check(launch_prep(P, cx.s));
  }
  if (w.has_skip) {
    run_conv(cx, raw, w.skip, out, nullptr, nullptr, 0);
    run_conv(cx, A2, w.c2, out, &out, nullptr, 0);
  } else if (need_raw) {
    run_conv(cx, A2, w.c2, out, &raw, nullptr, 0);
  } else {
    run_conv(cx, A2, w.c2, out, &src[0].t, nullptr, 0);
  }
  cx.top = mark;
  return out;
}

// AttnBlock._forward with cond (model/MBAblocks.py:484-489); x updated in place
static void attn_block(Ctx& cx, const AttnW& w, TV x, const Src& cond, int per_image, const TV* cond_act = nullptr) {
  const size_t mark = cx.top;
  const int N = x.N, Z = x.Z, S = x.H, C = w.C, cb = C / 8;
  if (is_h16(cx.m->cfg.dtype)) {
    const int h_f16 = cx.m->cfg.dtype == TM_DTYPE_F16;
    // bf16 operands for every Linear (fp32 accumulate); the residual stream x, q/k/v and the softmax stay fp32.
    // Activations that only feed a Linear, and the 7C modulation tensor (shift/scale/gate/cross-cond chunks), are
    // produced directly in bf16: the cross-cond chunk is the kv Linear's input as it stands.
    const int gbe = ((w.G + 7) / 8 + 1) / 2 * 2;
    const int Tw = Z * (S / 2) * (S / 2);
    const bool mfma_core = Tw == 128 || Tw == 64 || Tw == 32 || ((Tw == 256 || Tw == 512) && C <= 256);
    // The conditioning side at HALF resolution.  cond is an RNA pyramid level, and every level leaves its stage through a
    // nearest x2 Upsample (unet_ours.py:290-295, MBAblocks.py:472-479): it -- and with it SiLU(cond), the 7C adaLN modulation
    // Linear(SiLU(cond)) (MBAblocks.py:463-466,487), the cross-cond chunk, k and v (no positional term, :551-556) -- is
    // constant over aligned 2 x 2 voxel blocks, also after to_collage (a shift by S/2, even).  So these tensors are computed
    // once per block of four voxels ([.., Z, S/2, S/2]: a quarter of the Linear work and of the bytes of the largest tensor
    // of the block) and their consumers read entry (z, y >> 1, x >> 1).  Same numbers, bit for bit (TM_ATTN_HALF=0: the
    // full-resolution form, for A/B timing and the equality test).
    static const bool no_half = getenv("TM_ATTN_HALF") && atoi(getenv("TM_ATTN_HALF")) == 0;
    const bool half = !no_half && (Tw == 128 || Tw == 64 || Tw == 32) && S >= 4 && !(S & (S - 1));
    const int Sc = half ? S / 2 : S;
    // src16: the source is a 16-bit stream tensor (x, the RNA level); otherwise an fp32 scratch tensor
    auto prep_h = [&](const float* p, long ns, int Cbs, bool collage, const float* nw, const TVH* sc, const TVH* sh, int act,
                      TVH dst, int Creal, bool src16 = true, int So = 0, int resample = RS_SAME) {
      if (cx.dry) return;
      PrepLaunch P;
      P.nsrc = 1;
      P.src_h = src16 ? 1 : 0;
      P.src[0].p = p; P.src[0].nstride = ns; P.src[0].Cb = Cbs; P.src[0].collage = collage ? 1 : 0;
      P.N = N; P.Z = Z; P.S = So ? So : S; P.p1 = cx.p1; P.p2 = cx.p2; P.act = act; P.per_image = per_image;
      P.resample = resample;
      P.norm_w = nw; P.inv_c = 1.0f / (float)Creal;
      if (sc) { P.mod = MOD_VOXEL; P.mod_scale_h = sc->p; P.mod_shift_h = sh->p; P.mod_stride = sc->nstride; P.mod_half = half ? 1 : 0; }
      P.out_h = dst.p; P.out_h_nstride = dst.nstride; P.pad_blocks = dst.Cb - Cbs; P.h_f16 = h_f16;
      cx.check(launch_prep(P, cx.s));
    };
    const long cplane = (long)Z * Sc * Sc * 8;                     // elements per channel block on the conditioning side
    TVH cact = cx.tensor_h(N, gbe, Z, Sc);
    prep_h(cond.t.p, cond.t.nstride, cond.t.Cb, cond.collage, nullptr, nullptr, nullptr, 1, cact, w.G, true, Sc,
           half ? RS_PICK2 : RS_SAME);
    TVH mod = cx.tensor_h(N, 7 * cb, Z, Sc);
    TV mod_geom = x; mod_geom.H = Sc; mod_geom.W = Sc; mod_geom.Cb = 7 * cb; mod_geom.C = 7 * C; mod_geom.p = nullptr;
    mod_geom.nstride = (long)7 * cb * cplane;
    run_conv1_h(cx, cact, w.adah, w.ada, mod_geom, nullptr, nullptr, 0, &mod);
    // chunk order (MBAblocks.py:487): shift_msa, scale_msa, gate_msa, crss_cnd, shift_mlp, scale_mlp, gate_mlp
    TVH sh_a = mod.blocks(0 * cb, cb), sc_a = mod.blocks(1 * cb, cb), g_a = mod.blocks(2 * cb, cb);
    TVH crs = mod.blocks(3 * cb, cb), sh_m = mod.blocks(4 * cb, cb), sc_m = mod.blocks(5 * cb, cb), g_m = mod.blocks(6 * cb, cb);
    TVH xa = cx.tensor_h(N, cb, Z, S), oh = cx.tensor_h(N, cb, Z, S);
    prep_h(x.p, x.nstride, x.Cb, false, w.n1, &sc_a, &sh_a, 0, xa, C);
    if (mfma_core) {
      // q, k, v leave their Linears as 16-bit (the attention core's MFMA operands); softmax and accumulation are fp32
      TVH q = cx.tensor_h(N, cb, Z, S), kv = cx.tensor_h(N, 2 * cb, Z, Sc);
      TV q_geom = x; q_geom.p = nullptr;
      TV kv_geom = x; kv_geom.H = Sc; kv_geom.W = Sc; kv_geom.Cb = 2 * cb; kv_geom.C = 2 * C; kv_geom.p = nullptr;
      kv_geom.nstride = (long)2 * cb * cplane;
      q_geom.nstride = (long)cb * x.plane();
      run_conv1_h(cx, xa, w.qh, w.q, q_geom, nullptr, nullptr, 0, &q);
      run_conv1_h(cx, crs, w.kvh, w.kv, kv_geom, nullptr, nullptr, 0, &kv);
      if (!cx.dry) cx.check((h_f16 ? launch_window_attn_f16 : launch_window_attn_bf16)(q, kv.blocks(0, cb), kv.blocks(cb, cb), w.qn, w.kn, oh, cx.s));
    } else {
      // other window sizes (z_size 1 / 4 / 8, patch_size 128): fp32 q / k / v into the generic fp32 attention core, its
      // output rounded to the 16-bit type for proj
      TV q = cx.tensor(N, C, Z, S), kv = cx.tensor(N, 2 * C, Z, S), o = cx.tensor(N, C, Z, S);
      run_conv1_h(cx, xa, w.qh, w.q, q, nullptr, nullptr, 0);
      run_conv1_h(cx, crs, w.kvh, w.kv, kv, nullptr, nullptr, 0);
      if (!cx.dry) cx.check(launch_window_attn(q, kv.blocks(0, cb), kv.blocks(cb, cb), w.qn, w.kn, o, cx.s));
      prep_h(o.p, o.nstride, o.Cb, false, nullptr, nullptr, nullptr, 0, oh, C, false);
    }
    // x <- x + gate * Linear(.): the 16-bit stream tensor is updated in place (each element is read and written by one lane)
    TVH xh = as_h(x);
    run_conv1_h(cx, oh, w.projh, w.proj, x, nullptr, nullptr, 0, &xh, &g_a, &xh, nullptr, half);
    prep_h(x.p, x.nstride, x.Cb, false, w.n2, &sc_m, &sh_m, 0, xa, C);
    TVH h1 = cx.tensor_h(N, 4 * cb, Z, S);
    TV h1_geom = x; h1_geom.Cb = 4 * cb; h1_geom.C = 4 * C; h1_geom.p = nullptr; h1_geom.nstride = (long)4 * cb * x.plane();
    run_conv1_h(cx, xa, w.fc1h, w.fc1, h1_geom, nullptr, nullptr, EPI_GELU, &h1);
    run_conv1_h(cx, h1, w.fc2h, w.fc2, x, nullptr, nullptr, 0, &xh, &g_m, &xh, nullptr, half);
    cx.top = mark;
    return;
  }
  // fp32: the same structure.  The conditioning side (SiLU(cond), the 7C modulation, k, v) at half resolution where the
  // attention core has the k / v half-resolution read (T = 128 MFMA and T = 32 kernels, C a multiple of 128), see above;
  // otherwise at full resolution, reusing the activated RNA level when the cond is not re-tiled (encoder / middle).
  const int Tw = Z * (S / 2) * (S / 2);
  static const bool no_half32 = getenv("TM_ATTN_HALF") && atoi(getenv("TM_ATTN_HALF")) == 0;
  const bool half = !no_half32 && S >= 4 && !(S & (S - 1)) && C % 128 == 0 && ((Tw == 128 && C <= 512) || Tw == 32);
  const int Sc = half ? S / 2 : S;
  TV cact = (cond_act && !half) ? *cond_act : cx.tensor(N, (w.G + 7) / 8 * 8, Z, Sc);
  if (!cx.dry && (half || !cond_act)) {
    PrepLaunch P;
    P.nsrc = 1;
    P.src[0].p = cond.t.p; P.src[0].nstride = cond.t.nstride; P.src[0].Cb = cond.t.Cb;
    P.src[0].collage = cond.collage ? 1 : 0;
    P.resample = half ? RS_PICK2 : RS_SAME;
    P.N = N; P.Z = Z; P.S = Sc; P.p1 = cx.p1; P.p2 = cx.p2; P.act = 1; P.per_image = per_image;
    P.out = cact.p; P.out_nstride = cact.nstride;
    cx.check(launch_prep(P, cx.s));
  }
  TV mod = cx.tensor(N, 7 * C, Z, Sc);
  run_conv(cx, cact, w.ada, mod, nullptr, nullptr, 0);
  // chunk order (MBAblocks.py:487): shift_msa, scale_msa, gate_msa, crss_cnd, shift_mlp, scale_mlp, gate_mlp
  TV sh_a = mod.blocks(0 * cb, cb), sc_a = mod.blocks(1 * cb, cb), g_a = mod.blocks(2 * cb, cb);
  TV crs = mod.blocks(3 * cb, cb), sh_m = mod.blocks(4 * cb, cb), sc_m = mod.blocks(5 * cb, cb), g_m = mod.blocks(6 * cb, cb);
  TV xa = cx.tensor(N, C, Z, S);
  auto modulate = [&](const float* nw, const TV& sc, const TV& sh, TV dst) {
    if (cx.dry) return;
    PrepLaunch P;
    P.nsrc = 1;
    P.src[0].p = x.p; P.src[0].nstride = x.nstride; P.src[0].Cb = x.Cb;
    P.N = N; P.Z = Z; P.S = S; P.norm_w = nw; P.inv_c = 1.0f / (float)C; P.per_image = per_image;
    P.mod = MOD_VOXEL; P.mod_scale = sc.p; P.mod_shift = sh.p; P.mod_stride = mod.nstride; P.mod_half = half ? 1 : 0;
    P.out = dst.p; P.out_nstride = dst.nstride;
    cx.check(launch_prep(P, cx.s));
  };
  modulate(w.n1, sc_a, sh_a, xa);
  TV q = cx.tensor(N, C, Z, S), kv = cx.tensor(N, 2 * C, Z, Sc), o = cx.tensor(N, C, Z, S);
  run_conv(cx, xa, w.q, q, nullptr, nullptr, 0);
  run_conv(cx, crs, w.kv, kv, nullptr, nullptr, 0);
  if (!cx.dry) cx.check(launch_window_attn(q, kv.blocks(0, cb), kv.blocks(cb, cb), w.qn, w.kn, o, cx.s));
  run_conv(cx, o, w.proj, x, &x, &g_a, 0, 0, ZM_PAD1, half);
  modulate(w.n2, sc_m, sh_m, xa);
  TV h1 = cx.tensor(N, 4 * C, Z, S);
  run_conv(cx, xa, w.fc1, h1, nullptr, nullptr, EPI_GELU);
  run_conv(cx, h1, w.fc2, x, &x, &g_m, 0, 0, ZM_PAD1, half);
  cx.top = mark;
}

static void run_direct(Ctx& cx, const DirectW& w, const float* x, Acc5 ax, float* y, Acc5 ay, int N, int Zin, int Zout,
                       int S, int pz, int silu_in, int up2) {
  if (cx.dry) return;
  DirectLaunch L;
  L.x = x; L.ax = ax; L.y = y; L.ay = ay; L.w = w.w; L.bias = w.bias;
  L.N = N; L.Cin = w.Cin; L.Cout = w.Cout; L.Zin = Zin; L.Zout = Zout; L.S = S;
  L.kz = w.kz; L.ky = w.ky; L.kx = w.kx; L.pz = pz; L.py = 1; L.px = 1;
  L.silu_in = silu_in; L.up2_out = up2;
  cx.check(launch_conv_direct(L, cx.s));
}

// The RNA conditioning of a call (get_rna, model/unet_ours.py:298-323): four pyramid levels (stream tensors: fp32, or 16-bit
// in the 16-bit modes) and, in fp32 mode, SiLU(level) of the first three (pyramid conv input and adaLN input of the
// non-re-tiled AttnBlocks).  It depends on the genes only, not on t or x: a sampler that runs many steps on the same genes
// (mode A, LitModel.gen_sample) computes it once (tm_rna_pyramid) and hands it to every step (tm_unet_forward_rna).
struct RnaOut { TV rl[4]; TV rs[3]; };

// Allocates the persistent outputs FIRST (so that their layout inside a caller-provided pyramid buffer is a pure function
// of (b, p1, p2)), computes them, and releases the scratch.  rna == nullptr: layout only.
// l0_in: level 0 (gene attention -> down_z -> Upsample, the part that reads the gene counts) was computed before
// (tm_rna_level0) and is taken from that buffer -- a tile sweep recomputes the conditioning of the SAME genes at every diffusion
// step (test_brn.py:232-255), and level 0 is both its costly part and small enough to keep (59 KB per patch).
// l0_out: compute level 0 only, into that buffer.
static void rna_stage(Ctx& cx, const float* rna, RnaOut& R, const void* l0_in = nullptr, void* l0_out = nullptr) {
  tm_model* m = cx.m;
  const tm_config& c = m->cfg;
  const int Z = m->z, Ne = cx.Ne;
  const bool h16 = is_h16(c.dtype), run = !cx.dry && (rna != nullptr || l0_in != nullptr);
  const bool head = l0_in == nullptr;                            // gene attention + down_z run in this call
  int S = m->gn * 2;
  for (int i = 0; i < 4; ++i) { R.rl[i] = cx.tensor_s(Ne, m->rw[i], Z, S); S *= 2; }
  if (l0_in) R.rl[0].p = (float*)const_cast<void*>(l0_in);
  if (l0_out) R.rl[0].p = (float*)l0_out;
  S = m->gn * 2;
  if (!h16) for (int i = 0; i < 3; ++i) { R.rs[i] = cx.tensor(Ne, m->rw[i], Z, S); S *= 2; }
  const size_t rna_mark = cx.top;
  TV tok = cx.tensor(Ne, c.rna_num, c.rna_slc, m->gn);           // gene-attention output, CB8 [Ne][Gb][zs][gn][gn][8]
  if (run && head) {
    cx.check(hipMemsetAsync(tok.p, 0, (size_t)Ne * tok.nstride * sizeof(float), cx.s));      // pad gene slots
    if (m->gene_mfma)
      cx.check(launch_gene_attn(rna, Ne, m->gn, c.rna_slc, c.rna_num, m->gene, tok.p, nullptr, 0, c.rna_slc, cx.s));
  }
  if (!m->gene_mfma) {
    const size_t mark = cx.top;
    float* gws = cx.alloc_f((size_t)Ne * gene_generic_split(Ne) * gene_generic_ws_floats(c.rna_num, m->D));
    if (run && head)
      cx.check(launch_gene_attn_generic(rna, Ne, m->gn, c.rna_slc, c.rna_num, m->D, m->gene, m->gene_idx, tok.p, nullptr, 0,
                                        c.rna_slc, gws, cx.s));
    cx.top = mark;                                               // scratch only (the stream orders its reuse)
  }
  const bool was_dry = cx.dry;
  cx.dry = !run;
  if (h16) {
    // 16-bit modes: gene attention and down_z stay fp32 (they read the fp32 gene counts; < 0.3 % of the FLOPs), level 0
    // enters the 16-bit stream behind down_z, and the three SiLU -> Conv3d(1,3,3) -> Upsample stages
    // (model/unet_ours.py:290-295) run as 16-bit convs like every other conv under the reference's autocast: the in-plane
    // conv is the 3x3x3 kernel on weights whose z - 1 / z + 1 slices are zero, its output leaves at the conv's resolution
    // and ONE pass of the block-input kernel writes both the x2 level (raw copy) and SiLU of it (the next conv's input).
    const int h_f16 = c.dtype == TM_DTYPE_F16;
    int S0 = m->gn * 2;
    TV rl0 = cx.tensor(Ne, m->rw[0], Z, S0);
    if (head) {
      if (m->downz_mfma) run_conv(cx, tok, m->downz, rl0, nullptr, nullptr, EPI_UP2, 0, ZM_VALID);      // down_z + Upsample
      else {
        const Acc5 ax = acc_cb8(tok), ay = acc_cb8(rl0);
        if (run) cx.check(hipMemsetAsync(rl0.p, 0, (size_t)Ne * rl0.nstride * sizeof(float), cx.s));     // pad channels
        run_direct(cx, m->downz_d, tok.p, ax, rl0.p, ay, Ne, c.rna_slc, Z, m->gn, 0, 0, 1);
      }
      if (run) {
        PrepLaunch P;                                              // fp32 level 0 -> the 16-bit stream tensor
        P.nsrc = 1;
        P.src[0].p = rl0.p; P.src[0].nstride = rl0.nstride; P.src[0].Cb = rl0.Cb;
        P.N = Ne; P.Z = Z; P.S = S0; P.h_f16 = h_f16;
        P.out_h = (uint16_t*)R.rl[0].p; P.out_h_nstride = R.rl[0].nstride;
        cx.check(launch_prep(P, cx.s));
      }
    }
    if (l0_out) { cx.dry = was_dry; cx.top = rna_mark; return; }
    const int cbe0 = (rl0.Cb + 1) / 2 * 2;
    TVH act = cx.tensor_h(Ne, cbe0, Z, S0);                        // SiLU(level) = the conv input (even block count)
    if (run) {
      PrepLaunch Q;                                                // SiLU of it, pair-padded
      Q.nsrc = 1; Q.src_h = 1; Q.h_f16 = h_f16;
      Q.src[0].p = R.rl[0].p; Q.src[0].nstride = R.rl[0].nstride; Q.src[0].Cb = R.rl[0].Cb;
      Q.N = Ne; Q.Z = Z; Q.S = S0; Q.act = 1;
      Q.out_h = act.p; Q.out_h_nstride = act.nstride; Q.pad_blocks = act.Cb - R.rl[0].Cb;
      cx.check(launch_prep(Q, cx.s));
    }
    int S = S0;
    for (int i = 1; i < 4; ++i) {
      TV y = cx.tensor_s(Ne, m->rw[i], Z, S);                      // conv output at the conv's resolution (16-bit)
      run_conv_h(cx, act, m->pyrh[i - 1], m->pyr16[i - 1], y, nullptr, m->rw[i - 1], nullptr, nullptr, 1, true, false);
      S *= 2;
      TVH nxt;
      if (i < 3) nxt = cx.tensor_h(Ne, (R.rl[i].Cb + 1) / 2 * 2, Z, S);
      if (run) {
        PrepLaunch U;                                              // Upsample: level i (raw) and SiLU(level i)
        U.nsrc = 1; U.src_h = 1; U.h_f16 = h_f16; U.resample = RS_UP2;
        U.src[0].p = y.p; U.src[0].nstride = y.nstride; U.src[0].Cb = y.Cb;
        U.N = Ne; U.Z = Z; U.S = S;
        if (i < 3 && nxt.Cb == R.rl[i].Cb) {
          U.act = 1; U.out_h = nxt.p; U.out_h_nstride = nxt.nstride;
          U.raw_h = (uint16_t*)R.rl[i].p; U.raw_h_nstride = R.rl[i].nstride;
          cx.check(launch_prep(U, cx.s));
        } else {
          U.out_h = (uint16_t*)R.rl[i].p; U.out_h_nstride = R.rl[i].nstride;
          cx.check(launch_prep(U, cx.s));
          if (i < 3) {                                             // odd block count: the pair-padded SiLU copy on its own
            PrepLaunch Q;
            Q.nsrc = 1; Q.src_h = 1; Q.h_f16 = h_f16;
            Q.src[0].p = R.rl[i].p; Q.src[0].nstride = R.rl[i].nstride; Q.src[0].Cb = R.rl[i].Cb;
            Q.N = Ne; Q.Z = Z; Q.S = S; Q.act = 1;
            Q.out_h = nxt.p; Q.out_h_nstride = nxt.nstride; Q.pad_blocks = nxt.Cb - R.rl[i].Cb;
            cx.check(launch_prep(Q, cx.s));
          }
        }
      }
      act = nxt;
    }
    cx.dry = was_dry;
    cx.top = rna_mark;
    return;
  }
  TV* rl = R.rl;
  TV* rs = R.rs;
  if (head) {
    if (m->downz_mfma) run_conv(cx, tok, m->downz, rl[0], nullptr, nullptr, EPI_UP2, 0, ZM_VALID);      // down_z + Upsample
    else {
      // generic (kz, gn): direct conv straight from / to the CB8 tensors, nearest x2 fused into the store
      const Acc5 ax = acc_cb8(tok), ay = acc_cb8(rl[0]);
      if (run) cx.check(hipMemsetAsync(rl[0].p, 0, (size_t)Ne * rl[0].nstride * sizeof(float), cx.s));   // pad channels
      run_direct(cx, m->downz_d, tok.p, ax, rl[0].p, ay, Ne, c.rna_slc, Z, m->gn, 0, 0, 1);
    }
  }
  if (l0_out) { cx.dry = was_dry; cx.top = rna_mark; return; }
  for (int i = 1; i < 4; ++i) {
    if (run) {
      PrepLaunch P;
      P.nsrc = 1;
      P.src[0].p = rl[i - 1].p; P.src[0].nstride = rl[i - 1].nstride; P.src[0].Cb = rl[i - 1].Cb;
      P.N = Ne; P.Z = Z; P.S = rl[i - 1].H; P.act = 1;
      P.out = rs[i - 1].p; P.out_nstride = rs[i - 1].nstride;
      cx.check(launch_prep(P, cx.s));
    }
    run_conv(cx, rs[i - 1], m->pyr[i - 1], rl[i], nullptr, nullptr, EPI_UP2, 0, ZM_INPLANE);   // SiLU -> conv -> Upsample
  }
  cx.dry = was_dry;
  cx.top = rna_mark;                                              // tok / fp32 scratch are dead (the stream orders reuse)
}

// rna != nullptr: compute the conditioning inside this call's workspace; otherwise R_in (tm_rna_pyramid) is used
static int forward_impl(Ctx& cx, const float* x, const int64_t* t, const float* rna, const RnaOut* R_in, float* pred,
                        float* pred2, const void* l0_in = nullptr) {
  tm_model* m = cx.m;
  const tm_config& c = m->cfg;
  const int Z = m->z, L = m->L, ps = c.patch_size, Ne = cx.Ne, Nd = cx.Nd, b = cx.b;
  const int ne_img = cx.p1 * cx.p2, nd_img = (cx.p1 - 1) * (cx.p2 - 1);
  const bool h16 = is_h16(c.dtype);
  // ---- time embedding + all emb_layers ----
  float* te = cx.alloc_f((size_t)b * c.embed_ch);
  float* ss = cx.alloc_f((size_t)b * m->emb_tot);
  cx.ss = ss;
  if (!cx.dry) {
    cx.check(launch_time_embed(t, b, c.net_ch, c.embed_ch, m->te_w1, m->te_b1, m->te_w2, m->te_b2, te, cx.s));
    cx.check(launch_emb_all(te, b, c.embed_ch, m->emb_w, m->emb_b, m->emb_tot, ss, cx.s));
  }
  // ---- RNA pyramid ----
  RnaOut Rloc;
  if (!R_in) { rna_stage(cx, rna, Rloc, l0_in); R_in = &Rloc; }
  TV rl[4], rs[3];
  for (int i = 0; i < 4; ++i) rl[i] = R_in->rl[i];
  for (int i = 0; i < 3; ++i) rs[i] = R_in->rs[i];
  for (int i = 0; i < 4; ++i) { TV v = rl[i]; v.C = m->rw[i]; dump_tv(cx, "rna." + std::to_string(i), v); }
  // ---- stem ----
  std::vector<std::vector<TV>> skips(L);
  TV h = cx.tensor_s(Ne, c.net_ch, Z, ps);
  if (!cx.dry) cx.check(launch_stem(x, h, m->stem.w, m->stem.bias, c.n_stain, cx.s, h16 ? (uint16_t*)h.p : nullptr, h.nstride,
                                    c.dtype == TM_DTYPE_F16));
  skips[0].push_back(h);
  dump_tv(cx, "stem", h);
  // ---- encoder ----
  for (const EncEntry& e : m->enc) {
    const TV& cond = rl[L - 1 - e.lvl];
    const int So = ps >> e.lvl;
    std::vector<Src> src;
    src.push_back({h, false});
    if (e.cat) src.push_back({cond, false});
    for (const Op& op : e.ops) {
      if (op.kind == 0) h = res_block(cx, m->res[op.idx], src, Ne, ne_img, So, op.mode, nullptr);
      else attn_block(cx, m->attn[op.idx], h, {cond, false}, ne_img, (L - 1 - e.lvl) < 3 ? &rs[L - 1 - e.lvl] : nullptr);
    }
    skips[e.lvl].push_back(h);
    const Op& last = e.ops.back();
    dump_tv(cx, last.kind == 0 ? m->res[last.idx].pfx : m->attn[last.idx].pfx, h);
  }
  // ---- middle ----
  {
    std::vector<Src> src = {{h, false}, {rl[0], false}};
    const int So = ps >> (L - 1);
    h = res_block(cx, m->res[m->mid[0].idx], src, Ne, ne_img, So, RS_SAME, nullptr);
    attn_block(cx, m->attn[m->mid[1].idx], h, {rl[0], false}, ne_img, &rs[0]);
    std::vector<Src> src2 = {{h, false}};
    h = res_block(cx, m->res[m->mid[2].idx], src2, Ne, ne_img, So, RS_SAME, nullptr);
    dump_tv(cx, "middle_block", h);
  }
  // ---- decoder(s) ----
  auto decode = [&](bool col, float* outp) {
    const size_t mark = cx.top;
    const int N = col ? Nd : Ne, per = col ? nd_img : ne_img;
    std::vector<std::vector<TV>> st = skips;
    TV hd = h;
    bool hd_col = col;                 // only the middle output still lives on the encoder grid
    for (const DecEntry& e : m->dec) {
      const int So = ps >> e.lvl;
      Src cond = {rl[L - 1 - e.lvl], col};
      std::vector<Src> src = {{hd, hd_col}, {st[e.lvl].back(), col}, cond};
      st[e.lvl].pop_back();
      for (const Op& op : e.ops) {
        if (op.kind == 0) {
          if (op.mode == RS_SAME) hd = res_block(cx, m->res[op.idx], src, N, per, So, RS_SAME, nullptr);
          else { std::vector<Src> s1 = {{hd, false}}; hd = res_block(cx, m->res[op.idx], s1, N, per, So * 2, RS_UP2, nullptr); }
        } else attn_block(cx, m->attn[op.idx], hd, cond, per);
      }
      hd_col = false;
      if (col) { const std::string& p0 = m->res[e.ops[0].idx].pfx; dump_tv(cx, p0.substr(0, p0.size() - 2), hd); }
    }
    // head: RMSNorm -> SiLU -> Conv3d(1,3,3) -> 'b s z h w -> b (s z) h w'  (unet_ours.py:271-275,423-424)
    // (16-bit modes: the normalised, activated tensor is a 16-bit tensor like every other conv input)
    TV A = cx.tensor_s(N, c.net_ch, Z, ps);
    if (!cx.dry) {
      PrepLaunch P;
      P.nsrc = 1;
      P.src[0].p = hd.p; P.src[0].nstride = hd.nstride; P.src[0].Cb = hd.Cb;
      P.src_h = h16 ? 1 : 0; P.h_f16 = c.dtype == TM_DTYPE_F16;
      P.N = N; P.Z = Z; P.S = ps; P.norm_w = m->out_norm; P.inv_c = 1.0f / (float)c.net_ch; P.act = 1; P.per_image = per;
      if (h16) { P.out_h = (uint16_t*)A.p; P.out_h_nstride = A.nstride; }
      else { P.out = A.p; P.out_nstride = A.nstride; }
      cx.check(launch_prep(P, cx.s));
    }
    if (!cx.dry) cx.check(launch_head(A, outp, m->head.w, m->head.bias, c.n_stain, cx.s, h16 ? (c.dtype == TM_DTYPE_F16 ? 2 : 1) : 0));
    cx.top = mark;
  };
  decode(true, pred);
  if (pred2) decode(false, pred2);
  return TM_OK;
}

static int check_fwd_args(const tm_model* m, int b, int p1, int p2) {
  if (!m) return fail(TM_ERR_ARG, "null model");
  if (!m->finalized) return fail(TM_ERR_STATE, "tm_model_finalize has not been called");
  if (m->cfg.vis_only) return fail(TM_ERR_STATE, "attention-map model has no UNet forward");
  if (b < 1 || p1 < 2 || p2 < 2) return fail(TM_ERR_ARG, "need b >= 1 and p1, p2 >= 2 (got %d, %d, %d)", b, p1, p2);
  return TM_OK;
}

extern "C" size_t tm_workspace_bytes(const tm_model* m, int b, int p1, int p2, int want_pred2) {
  if (check_fwd_args(m, b, p1, p2) != TM_OK) return 0;
  Ctx cx;
  cx.m = const_cast<tm_model*>(m); cx.dry = true;
  cx.b = b; cx.p1 = p1; cx.p2 = p2; cx.Ne = b * p1 * p2; cx.Nd = b * (p1 - 1) * (p2 - 1);
  forward_impl(cx, nullptr, nullptr, nullptr, nullptr, (float*)256, want_pred2 ? (float*)256 : nullptr);
  return cx.peak + 256;
}

extern "C" int tm_unet_forward(tm_model* m, const void* x, const int64_t* t, const void* rna_dense, int b, int p1,
                               int p2, void* pred, void* pred2_or_null, void* workspace, size_t workspace_bytes,
                               void* stream) {
  int rc = check_fwd_args(m, b, p1, p2);
  if (rc != TM_OK) return rc;
  if (!x || !t || !rna_dense || !pred || !workspace) return fail(TM_ERR_ARG, "null tensor argument");
  Ctx cx;
  cx.m = m; cx.s = (hipStream_t)stream;
  cx.base = (char*)(((uintptr_t)workspace + 255) / 256 * 256);
  cx.cap = workspace_bytes - (size_t)(cx.base - (char*)workspace);
  cx.b = b; cx.p1 = p1; cx.p2 = p2; cx.Ne = b * p1 * p2; cx.Nd = b * (p1 - 1) * (p2 - 1);
  const size_t need = tm_workspace_bytes(m, b, p1, p2, pred2_or_null != nullptr);
  if (workspace_bytes < need) return fail(TM_ERR_WORKSPACE, "workspace %zu B < required %zu B", workspace_bytes, need);
  forward_impl(cx, (const float*)x, t, (const float*)rna_dense, nullptr, (float*)pred, (float*)pred2_or_null);
  if (cx.err != hipSuccess) return fail(TM_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(cx.err));
  return TM_OK;
}

// ---- RNA conditioning computed once for many steps (mode A: the genes do not change over the reverse loop) ----
static void pyramid_ctx(Ctx& cx, tm_model* m, int b, int p1, int p2, void* buf, size_t bytes, hipStream_t s) {
  cx.m = m; cx.s = s;
  cx.b = b; cx.p1 = p1; cx.p2 = p2; cx.Ne = b * p1 * p2; cx.Nd = b * (p1 - 1) * (p2 - 1);
  if (buf) {
    cx.base = (char*)(((uintptr_t)buf + 255) / 256 * 256);
    cx.cap = bytes - (size_t)(cx.base - (char*)buf);
  } else cx.dry = true;
}
extern "C" size_t tm_rna_pyramid_bytes(const tm_model* m, int b, int p1, int p2) {
  if (check_fwd_args(m, b, p1, p2) != TM_OK) return 0;
  Ctx cx;
  pyramid_ctx(cx, const_cast<tm_model*>(m), b, p1, p2, nullptr, 0, nullptr);
  RnaOut R;
  rna_stage(cx, nullptr, R);
  return cx.peak + 512;
}
extern "C" int tm_rna_pyramid(tm_model* m, const void* rna_dense, int b, int p1, int p2, void* pyramid, size_t pyramid_bytes,
                              void* stream) {
  int rc = check_fwd_args(m, b, p1, p2);
  if (rc != TM_OK) return rc;
  if (!rna_dense || !pyramid) return fail(TM_ERR_ARG, "null tensor argument");
  const size_t need = tm_rna_pyramid_bytes(m, b, p1, p2);
  if (pyramid_bytes < need) return fail(TM_ERR_WORKSPACE, "pyramid buffer %zu B < required %zu B", pyramid_bytes, need);
  Ctx cx;
  pyramid_ctx(cx, m, b, p1, p2, pyramid, pyramid_bytes, (hipStream_t)stream);
  RnaOut R;
  rna_stage(cx, (const float*)rna_dense, R);
  if (cx.err != hipSuccess) return fail(TM_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(cx.err));
  return TM_OK;
}
// ---- level 0 of the RNA conditioning on its own (the sweep's per-window cache) ----
static TV level0_geom(const tm_model* m, int Ne) {
  TV t;
  t.N = Ne; t.C = m->rw[0]; t.Cb = (m->rw[0] + 7) / 8; t.Z = m->z; t.H = m->gn * 2; t.W = m->gn * 2;
  t.nstride = (long)t.Cb * t.plane();
  return t;
}
extern "C" size_t tm_rna_level0_bytes(const tm_model* m, int b, int p1, int p2) {
  if (check_fwd_args(m, b, p1, p2) != TM_OK) return 0;
  const TV t = level0_geom(m, b * p1 * p2);
  return (size_t)t.N * t.nstride * (is_h16(m->cfg.dtype) ? 2 : 4);
}
static void plain_ctx(Ctx& cx, tm_model* m, int b, int p1, int p2, void* workspace, size_t workspace_bytes, hipStream_t s) {
  cx.m = m; cx.s = s;
  cx.base = (char*)(((uintptr_t)workspace + 255) / 256 * 256);
  cx.cap = workspace_bytes - (size_t)(cx.base - (char*)workspace);
  cx.b = b; cx.p1 = p1; cx.p2 = p2; cx.Ne = b * p1 * p2; cx.Nd = b * (p1 - 1) * (p2 - 1);
}
extern "C" int tm_rna_level0(tm_model* m, const void* rna_dense, int b, int p1, int p2, void* level0, size_t level0_bytes,
                             void* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_fwd_args(m, b, p1, p2);
  if (rc != TM_OK) return rc;
  if (!rna_dense || !level0 || !workspace) return fail(TM_ERR_ARG, "null tensor argument");
  if (((uintptr_t)level0 & 15) != 0) return fail(TM_ERR_ARG, "level0 buffer must be 16-byte aligned");
  if (level0_bytes < tm_rna_level0_bytes(m, b, p1, p2)) return fail(TM_ERR_WORKSPACE, "level-0 buffer too small for (b, p1, p2)");
  const size_t need = tm_workspace_bytes(m, b, p1, p2, 0);
  if (workspace_bytes < need) return fail(TM_ERR_WORKSPACE, "workspace %zu B < required %zu B", workspace_bytes, need);
  Ctx cx;
  plain_ctx(cx, m, b, p1, p2, workspace, workspace_bytes, (hipStream_t)stream);
  RnaOut R;
  rna_stage(cx, (const float*)rna_dense, R, nullptr, level0);
  if (cx.err != hipSuccess) return fail(TM_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(cx.err));
  return TM_OK;
}
extern "C" int tm_unet_forward_level0(tm_model* m, const void* x, const int64_t* t, const void* level0, size_t level0_bytes,
                                      int b, int p1, int p2, void* pred, void* pred2_or_null, void* workspace,
                                      size_t workspace_bytes, void* stream) {
  int rc = check_fwd_args(m, b, p1, p2);
  if (rc != TM_OK) return rc;
  if (!x || !t || !level0 || !pred || !workspace) return fail(TM_ERR_ARG, "null tensor argument");
  if (((uintptr_t)level0 & 15) != 0) return fail(TM_ERR_ARG, "level0 buffer must be 16-byte aligned");
  if (level0_bytes < tm_rna_level0_bytes(m, b, p1, p2)) return fail(TM_ERR_WORKSPACE, "level-0 buffer too small for (b, p1, p2)");
  const size_t need = tm_workspace_bytes(m, b, p1, p2, pred2_or_null != nullptr);
  if (workspace_bytes < need) return fail(TM_ERR_WORKSPACE, "workspace %zu B < required %zu B", workspace_bytes, need);
  Ctx cx;
  plain_ctx(cx, m, b, p1, p2, workspace, workspace_bytes, (hipStream_t)stream);
  forward_impl(cx, (const float*)x, t, nullptr, nullptr, (float*)pred, (float*)pred2_or_null, level0);
  if (cx.err != hipSuccess) return fail(TM_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(cx.err));
  return TM_OK;
}
extern "C" int tm_unet_forward_rna(tm_model* m, const void* x, const int64_t* t, const void* pyramid, size_t pyramid_bytes,
                                   int b, int p1, int p2, void* pred, void* pred2_or_null, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  int rc = check_fwd_args(m, b, p1, p2);
  if (rc != TM_OK) return rc;
  if (!x || !t || !pyramid || !pred || !workspace) return fail(TM_ERR_ARG, "null tensor argument");
  if (pyramid_bytes < tm_rna_pyramid_bytes(m, b, p1, p2)) return fail(TM_ERR_WORKSPACE, "pyramid buffer too small for (b, p1, p2)");
  // the layout of the persistent tensors inside the pyramid buffer is a pure function of (b, p1, p2): replay it
  Ctx px;
  pyramid_ctx(px, m, b, p1, p2, const_cast<void*>(pyramid), pyramid_bytes, (hipStream_t)stream);
  RnaOut R;
  rna_stage(px, nullptr, R);
  Ctx cx;
  cx.m = m; cx.s = (hipStream_t)stream;
  cx.base = (char*)(((uintptr_t)workspace + 255) / 256 * 256);
  cx.cap = workspace_bytes - (size_t)(cx.base - (char*)workspace);
  cx.b = b; cx.p1 = p1; cx.p2 = p2; cx.Ne = b * p1 * p2; cx.Nd = b * (p1 - 1) * (p2 - 1);
  const size_t need = tm_workspace_bytes(m, b, p1, p2, pred2_or_null != nullptr);      // upper bound (it includes the RNA scratch)
  if (workspace_bytes < need) return fail(TM_ERR_WORKSPACE, "workspace %zu B < required %zu B", workspace_bytes, need);
  forward_impl(cx, (const float*)x, t, nullptr, &R, (float*)pred, (float*)pred2_or_null);
  if (cx.err != hipSuccess) return fail(TM_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(cx.err));
  return TM_OK;
}

// ------------------------------------------------------------------------------------------
extern "C" int tm_sampler_step(const tm_step_coefs* c, const void* x_patches, const void* eps, const void* noise,
                               void* x_prev_img, int b, int P1, int P2, int C, int ps, int mode, void* stream) {
  if (!c || !x_patches || !eps || !x_prev_img) return fail(TM_ERR_ARG, "null argument");
  if (b < 1 || P1 < 1 || P2 < 1 || C < 1 || ps < 2 || (ps & 1)) return fail(TM_ERR_ARG, "bad geometry");
  if (mode != TM_SAMPLE_DDPM && mode != TM_SAMPLE_DDIM) return fail(TM_ERR_ARG, "mode must be DDPM or DDIM");
  StepCoefs k = {c->sqrt_recip_alphas_cumprod, c->sqrt_recipm1_alphas_cumprod, c->posterior_mean_coef1,
                 c->posterior_mean_coef2, c->sigma, c->sqrt_alpha_bar_prev, c->sqrt_one_minus_alpha_bar_prev};
  HIP_TRY(launch_sampler_step(k, (const float*)x_patches, (const float*)eps, (const float*)noise, (float*)x_prev_img,
                              b, P1, P2, C, ps, mode, (hipStream_t)stream));
  return TM_OK;
}

extern "C" int tm_pad_patchify(const void* img, void* patches, int b, int C, int P1, int P2, int ps, float pad,
                               void* stream) {
  if (!img || !patches || b < 1 || P1 < 1 || P2 < 1 || C < 1 || ps < 2 || (ps & 1)) return fail(TM_ERR_ARG, "bad argument");
  HIP_TRY(launch_pad_patchify((const float*)img, (float*)patches, b, C, P1, P2, ps, pad, (hipStream_t)stream));
  return TM_OK;
}

extern "C" size_t tm_gene_attn_workspace_bytes(const tm_model* m, int B) {
  if (!m || B < 1 || m->gene_mfma) return 256;
  return 256 + (size_t)B * gene_generic_split(B) * gene_generic_ws_floats(m->cfg.rna_num, m->D) * sizeof(float);
}

extern "C" int tm_gene_attn(tm_model* m, const void* rna_dense, int B, void* attn_out, void* rna_mid, void* workspace,
                            size_t workspace_bytes, void* stream) {
  if (!m || !rna_dense || !attn_out || B < 1) return fail(TM_ERR_ARG, "bad argument");
  if (!m->finalized) return fail(TM_ERR_STATE, "tm_model_finalize has not been called");
  const tm_config& c = m->cfg;
  const int G = c.rna_num, zs = c.rna_slc;
  if (zs != 4) return fail(TM_ERR_ARG, "the attention-map read-out is defined for rna_slc = 4 (three slice pairs, model/unet_attn.py:162-172)");
  if (workspace_bytes < tm_gene_attn_workspace_bytes(m, B) || (!m->gene_mfma && !workspace))
    return fail(TM_ERR_WORKSPACE, "workspace %zu B < required %zu B", workspace_bytes, tm_gene_attn_workspace_bytes(m, B));
  float* ao = (float*)attn_out;
  hipStream_t s = (hipStream_t)stream;
  // three slice-pair masks then the unmasked map (model/unet_attn.py:162-172)
  for (int i = 0; i < 4; ++i) {
    const int lo = (i < 3) ? i : 0, hi = (i < 3) ? i + 2 : zs;
    if (m->gene_mfma)
      HIP_TRY(launch_gene_attn((const float*)rna_dense, B, m->gn, zs, G, m->gene, nullptr, ao + (size_t)i * B * G * G, lo, hi, s));
    else
      HIP_TRY(launch_gene_attn_generic((const float*)rna_dense, B, m->gn, zs, G, m->D, m->gene, m->gene_idx, nullptr,
                                       ao + (size_t)i * B * G * G, lo, hi, (float*)workspace, s));
  }
  if (rna_mid) HIP_TRY(launch_rna_mid((const float*)rna_dense, B, m->gn, zs, G, (float*)rna_mid, s));
  return TM_OK;
}

// ------------------------------------------------------------------------------------------
// single-operator entry points (tests)
// ------------------------------------------------------------------------------------------
static TV view_cb8(void* p, int N, int C, int Z, int H, int W) {
  TV t;
  t.p = (float*)p; t.N = N; t.C = C; t.Cb = (C + 7) / 8; t.Z = Z; t.H = H; t.W = W;
  t.nstride = (long)t.Cb * t.plane();
  return t;
}
extern "C" int tm_op_to_cb8(const void* x, void* y, int N, int C, int Z, int H, int W, void* stream) {
  HIP_TRY(launch_to_cb8((const float*)x, view_cb8(y, N, C, Z, H, W), (hipStream_t)stream));
  return TM_OK;
}
extern "C" int tm_op_from_cb8(const void* x, void* y, int N, int C, int Z, int H, int W, void* stream) {
  HIP_TRY(launch_from_cb8(view_cb8(const_cast<void*>(x), N, C, Z, H, W), (float*)y, (hipStream_t)stream));
  return TM_OK;
}
extern "C" int tm_op_conv_mfma(const void* x_cb8, const void* w_host, const void* bias_host, void* y_cb8, int N, int Cin,
                               int Cout, int Z, int S, int ksize, int zmode, int up2, int tile_variant, void* stream) {
  if (ksize != 1 && ksize != 3) return fail(TM_ERR_ARG, "ksize must be 1 or 3");
  if (zmode != ZM_PAD1 && zmode != ZM_INPLANE && zmode != ZM_VALID && zmode != ZM_UPS) return fail(TM_ERR_ARG, "bad zmode");
  const bool ups = ksize == 3 && zmode == ZM_UPS;       // w [Cout][Cin][27]: conv of the nearest-x2 upsampled x (y at 2S)
  if (zmode == ZM_UPS && (ksize != 3 || up2)) return fail(TM_ERR_ARG, "ZM_UPS: ksize 3, no fused upsample of the output");
  const int taps = ksize == 1 ? 1 : (zmode == ZM_INPLANE ? 9 : (ups ? 12 : 27));
  const int Zout = (ksize == 3 && zmode == ZM_VALID) ? Z - 2 : Z;
  const int So = (up2 || ups) ? 2 * S : S;
  ConvW cw;
  cw.Cout = Cout; cw.Cbi = (Cin + 7) / 8; cw.taps = taps; cw.ntile = (Cout + 63) / 64;
  std::vector<float> pk(ups ? conv_pack_ups_floats(Cout, cw.Cbi) : conv_pack_floats(Cout, cw.Cbi, taps)), bp((size_t)cw.ntile * 64, 0.f);
  if (ups) conv_pack_ups_host((const float*)w_host, Cout, &Cin, 1, pk.data());
  else conv_pack_host((const float*)w_host, Cout, &Cin, 1, taps, pk.data());
  memcpy(bp.data(), bias_host, Cout * sizeof(float));
  float *dw = nullptr, *db = nullptr;
  HIP_TRY(hipMalloc((void**)&dw, pk.size() * sizeof(float)));
  HIP_TRY(hipMalloc((void**)&db, bp.size() * sizeof(float)));
  HIP_TRY(hipMemcpy(dw, pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(db, bp.data(), bp.size() * sizeof(float), hipMemcpyHostToDevice));
  cw.w = dw; cw.bias = db;
  ConvLaunch L;
  L.x = view_cb8(const_cast<void*>(x_cb8), N, Cin, Z, S, S);
  L.w = cw;
  L.y = view_cb8(y_cb8, N, Cout, Zout, So, So);
  L.tile_variant = tile_variant;
  L.zmode = zmode;
  L.flags = up2 ? EPI_UP2 : 0;
  hipError_t e = launch_conv_mfma(L, (hipStream_t)stream);
  hipError_t e2 = hipStreamSynchronize((hipStream_t)stream);
  (void)hipFree(dw); (void)hipFree(db);
  if (e != hipSuccess) return fail(TM_ERR_HIP, "launch_conv_mfma: %s", hipGetErrorString(e));
  if (e2 != hipSuccess) return fail(TM_ERR_HIP, "conv_mfma execution: %s", hipGetErrorString(e2));
  return TM_OK;
}
// shared body of the 16-bit 3x3x3 conv test entry points: fp32 CB8 input -> 16-bit CB8 (prep kernel), then the conv
static int op_conv27_h16(const void* x_cb8, const void* w_host, const void* bias_host, void* y_cb8, int N, int Cin, int Cout,
                         int S, int dtype, int waves, const void* norm_w_host, const void* scale_host, const void* shift_host,
                         int per_image, void* a2_out, void* stream, const void* res_h16 = nullptr, void* y_h16 = nullptr,
                         int ups = 0, int res_half = 0) {
  if (!is_h16(dtype)) return fail(TM_ERR_ARG, "dtype must be TM_DTYPE_BF16 or TM_DTYPE_F16");
  if (waves != 0 && waves != 4 && waves != 8 && waves != 9) return fail(TM_ERR_ARG, "waves must be 0 (auto), 4, 8 or 9 (lockstep 8-wave form)");
  const bool f16 = dtype == TM_DTYPE_F16, fused = norm_w_host != nullptr;
  if (fused && (!scale_host || !shift_host || !a2_out || per_image < 1 || (Cout != 64 && Cout != 128)))
    return fail(TM_ERR_ARG, "fused epilogue needs Cout in {64, 128}, scale / shift / a2 and per_image >= 1");
  hipStream_t st = (hipStream_t)stream;
  const int Cbi = (Cin + 7) / 8, Cbe = (Cbi + 1) / 2 * 2, nt64 = (Cout + 63) / 64;
  if (ups && (Cout % 128 || res_h16)) return fail(TM_ERR_ARG, "upsampled-input form: Cout a multiple of 128, no residual");
  const int So = ups ? 2 * S : S;                       // output plane size
  std::vector<uint16_t> pk(ups ? conv_bf16_pack_ups_elems(Cout, Cbi) : conv_bf16_pack_elems(Cout, Cbi));
  if (ups) (f16 ? conv_f16_pack_ups_host : conv_bf16_pack_ups_host)((const float*)w_host, Cout, &Cin, 1, pk.data());
  else (f16 ? conv_f16_pack_host : conv_bf16_pack_host)((const float*)w_host, Cout, &Cin, 1, pk.data());
  const int nimg = (N + per_image - 1) / per_image;
  // one device buffer of floats: bias | norm_w | scale [nimg][Cout] | shift [nimg][Cout]
  std::vector<float> fp((size_t)nt64 * 64 + (fused ? (size_t)Cout * (1 + 2 * nimg) : 0), 0.f);
  memcpy(fp.data(), bias_host, Cout * sizeof(float));
  if (fused) {
    memcpy(fp.data() + nt64 * 64, norm_w_host, Cout * sizeof(float));
    memcpy(fp.data() + nt64 * 64 + Cout, scale_host, (size_t)nimg * Cout * sizeof(float));
    memcpy(fp.data() + nt64 * 64 + Cout + (size_t)nimg * Cout, shift_host, (size_t)nimg * Cout * sizeof(float));
  }
  uint16_t *dw = nullptr, *dx = nullptr;
  float* df = nullptr;
  const long vox = (long)2 * S * S;
  HIP_TRY(hipMalloc((void**)&dw, pk.size() * sizeof(uint16_t)));
  HIP_TRY(hipMalloc((void**)&df, fp.size() * sizeof(float)));
  HIP_TRY(hipMalloc((void**)&dx, (size_t)N * Cbe * vox * 8 * sizeof(uint16_t)));
  HIP_TRY(hipMemcpy(dw, pk.data(), pk.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(df, fp.data(), fp.size() * sizeof(float), hipMemcpyHostToDevice));
  TV x = view_cb8(const_cast<void*>(x_cb8), N, Cin, 2, S, S);
  PrepLaunch P;                       // fp32 CB8 -> 16-bit CB8 (no norm / act), pair padding
  P.nsrc = 1;
  P.src[0].p = x.p; P.src[0].nstride = x.nstride; P.src[0].Cb = x.Cb;
  P.N = N; P.Z = 2; P.S = S; P.h_f16 = f16 ? 1 : 0;
  P.out_h = dx; P.out_h_nstride = (long)Cbe * vox * 8; P.pad_blocks = Cbe - Cbi;
  hipError_t e0 = launch_prep(P, st);
  ConvLaunchH L;
  L.x.p = dx; L.x.N = N; L.x.Cb = Cbe; L.x.C = Cbe * 8; L.x.Z = 2; L.x.H = S; L.x.W = S; L.x.nstride = P.out_h_nstride;
  L.w = dw; L.bias = df; L.Cout = Cout; L.force_waves = waves;
  L.y = view_cb8(y_cb8, N, Cout, 2, So, So);
  L.ups = ups; L.res_half = res_half;
  TVH resh = as_h(L.y);
  if (res_h16) {
    resh.p = (uint16_t*)const_cast<void*>(res_h16); L.res_h = &resh;
    if (res_half) { resh.H = So / 2; resh.W = So / 2; resh.nstride = L.y.nstride / 4; }
  }
  if (y_h16) { L.y_h = (uint16_t*)y_h16; L.yh_nstride = L.y.nstride; }
  if (fused) {
    L.fuse_norm = 1; L.norm_w = df + nt64 * 64; L.mod_scale = L.norm_w + Cout; L.mod_shift = L.mod_scale + (size_t)nimg * Cout;
    L.mod_stride = Cout; L.per_image = per_image;
    L.a2.p = (uint16_t*)a2_out; L.a2.N = N; L.a2.Cb = Cout / 8; L.a2.C = Cout; L.a2.Z = 2; L.a2.H = So; L.a2.W = So;
    L.a2.nstride = (long)(Cout / 8) * 2 * So * So * 8;
  }
  hipError_t e = (f16 ? launch_conv27_f16 : launch_conv27_bf16)(L, st);
  hipError_t e2 = hipStreamSynchronize(st);
  (void)hipFree(dw); (void)hipFree(df); (void)hipFree(dx);
  if (e0 != hipSuccess) return fail(TM_ERR_HIP, "launch_prep: %s", hipGetErrorString(e0));
  if (e != hipSuccess) return fail(TM_ERR_HIP, "launch_conv27 (16-bit): %s", hipGetErrorString(e));
  if (e2 != hipSuccess) return fail(TM_ERR_HIP, "conv27 (16-bit) execution: %s", hipGetErrorString(e2));
  return TM_OK;
}
extern "C" int tm_op_conv27_bf16(const void* x_cb8, const void* w_host, const void* bias_host, void* y_cb8, int N, int Cin,
                                 int Cout, int S, int dtype, int waves, const void* res_h16, void* y_h16, int ups, int res_half,
                                 void* stream) {
  if (!x_cb8 || !w_host || !bias_host || (!y_cb8 && !y_h16)) return fail(TM_ERR_ARG, "null argument");
  return op_conv27_h16(x_cb8, w_host, bias_host, y_cb8 ? y_cb8 : y_h16, N, Cin, Cout, S, dtype, waves, nullptr, nullptr, nullptr, 1,
                       nullptr, stream, res_h16, y_h16, ups, res_half);
}
extern "C" int tm_op_conv27_fused(const void* x_cb8, const void* w_host, const void* bias_host, const void* norm_w_host,
                                  const void* scale_host, const void* shift_host, void* a2_out, int N, int Cin, int Cout,
                                  int S, int per_image, int dtype, int waves, void* stream) {
  if (!x_cb8 || !w_host || !bias_host || !norm_w_host || !a2_out) return fail(TM_ERR_ARG, "null argument");
  // the launcher takes the output geometry from `y`; the fused form never writes it
  return op_conv27_h16(x_cb8, w_host, bias_host, a2_out, N, Cin, Cout, S, dtype, waves, norm_w_host, scale_host, shift_host,
                       per_image, a2_out, stream);
}
// Timing hook of the 16-bit 3x3x3 conv on random device data (uniform in [-1, 1): the clock the chip holds depends on the
// operand bits, cdna guide rule 25): the model's launch forms -- 16-bit stream output with an optional 16-bit residual, the
// fused norm epilogue, the upsampled-input form -- `iters` launches between two events after one warm-up launch.
__global__ void fill_h16_kernel(uint16_t* p, size_t n, unsigned seed, int f16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned hsh = (unsigned)i * 2654435761u ^ (unsigned)(i >> 32) * 40503u ^ seed;
    hsh ^= hsh >> 15; hsh *= 2246822519u; hsh ^= hsh >> 13; hsh *= 3266489917u; hsh ^= hsh >> 16;
    const float v = (float)(hsh >> 8) * (1.0f / 8388608.0f) - 1.0f;
    uint16_t u;
    if (f16) { const _Float16 hf = (_Float16)v; u = __builtin_bit_cast(uint16_t, hf); }
    else { const __bf16 bf = (__bf16)v; u = __builtin_bit_cast(uint16_t, bf); }
    p[i] = u;
  }
}
extern "C" int tm_op_conv27_time(int N, int Cin, int Cout, int S, int dtype, int waves, int ups, int with_res, int fused,
                                 int iters, float* ms_per_launch, void* stream) {
  if (!is_h16(dtype) || iters < 1 || !ms_per_launch || N < 1) return fail(TM_ERR_ARG, "bad argument");
  if (waves != 0 && waves != 4 && waves != 8 && waves != 9) return fail(TM_ERR_ARG, "waves must be 0 (auto), 4, 8 or 9 (lockstep 8-wave form)");
  if (fused && (Cout != 64 && Cout != 128)) return fail(TM_ERR_ARG, "fused epilogue needs Cout in {64, 128}");
  if (ups && (Cout % 128 || with_res)) return fail(TM_ERR_ARG, "upsampled-input form: Cout a multiple of 128, no residual");
  const bool f16 = dtype == TM_DTYPE_F16;
  hipStream_t st = (hipStream_t)stream;
  const int Cbi = (Cin + 7) / 8, Cbe = (Cbi + 1) / 2 * 2, nt64 = (Cout + 63) / 64, So = ups ? 2 * S : S;
  const size_t nw = ups ? conv_bf16_pack_ups_elems(Cout, Cbi) : conv_bf16_pack_elems(Cout, Cbi);
  const long vox = (long)2 * S * S, voxo = (long)2 * So * So;
  const size_t nx = (size_t)N * Cbe * vox * 8, ny = (size_t)N * ((Cout + 7) / 8) * voxo * 8;
  uint16_t *dw = nullptr, *dx = nullptr, *dy = nullptr, *dr = nullptr;
  float* df = nullptr;
  const size_t nf = (size_t)nt64 * 64 + (size_t)Cout * 3;
  HIP_TRY(hipMalloc((void**)&dw, nw * 2));
  HIP_TRY(hipMalloc((void**)&dx, nx * 2));
  HIP_TRY(hipMalloc((void**)&dy, ny * 2));
  if (with_res) HIP_TRY(hipMalloc((void**)&dr, ny * 2));
  HIP_TRY(hipMalloc((void**)&df, nf * sizeof(float)));
  hipLaunchKernelGGL(fill_h16_kernel, dim3(2048), dim3(256), 0, st, dw, nw, 11u, f16 ? 1 : 0);
  hipLaunchKernelGGL(fill_h16_kernel, dim3(2048), dim3(256), 0, st, dx, nx, 23u, f16 ? 1 : 0);
  if (with_res) hipLaunchKernelGGL(fill_h16_kernel, dim3(2048), dim3(256), 0, st, dr, ny, 37u, f16 ? 1 : 0);
  std::vector<float> fp(nf, 0.f);
  for (size_t i = 0; i < nf; ++i) fp[i] = 0.01f * (float)((int)(i * 37 % 101) - 50);
  for (int i = 0; i < Cout; ++i) fp[(size_t)nt64 * 64 + i] = 1.0f + 0.001f * (float)(i % 17);      // norm_w
  HIP_TRY(hipMemcpyAsync(df, fp.data(), nf * sizeof(float), hipMemcpyHostToDevice, st));
  HIP_TRY(hipStreamSynchronize(st));
  ConvLaunchH L;
  L.x.p = dx; L.x.N = N; L.x.Cb = Cbe; L.x.C = Cbe * 8; L.x.Z = 2; L.x.H = S; L.x.W = S; L.x.nstride = (long)Cbe * vox * 8;
  L.w = dw; L.bias = df; L.Cout = Cout; L.force_waves = waves;
  L.y = view_cb8(dy, N, Cout, 2, So, So);
  L.ups = ups;
  TVH resh = as_h(L.y);
  if (with_res) { resh.p = dr; L.res_h = &resh; }
  L.y_h = dy; L.yh_nstride = L.y.nstride;
  if (fused) {
    L.fuse_norm = 1; L.norm_w = df + nt64 * 64; L.mod_scale = L.norm_w + Cout; L.mod_shift = L.mod_scale + Cout;
    L.mod_stride = 0; L.per_image = N;
    L.a2.p = dy; L.a2.N = N; L.a2.Cb = Cout / 8; L.a2.C = Cout; L.a2.Z = 2; L.a2.H = So; L.a2.W = So;
    L.a2.nstride = (long)(Cout / 8) * voxo * 8;
  }
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  hipError_t e = (f16 ? launch_conv27_f16 : launch_conv27_bf16)(L, st);
  (void)hipEventRecord(e0, st);
  for (int i = 0; i < iters && e == hipSuccess; ++i) e = (f16 ? launch_conv27_f16 : launch_conv27_bf16)(L, st);
  (void)hipEventRecord(e1, st);
  hipError_t e2 = hipStreamSynchronize(st);
  *ms_per_launch = 0.f;
  if (e == hipSuccess && e2 == hipSuccess) { (void)hipEventElapsedTime(ms_per_launch, e0, e1); *ms_per_launch /= (float)iters; }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(dw); (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(df);
  if (dr) (void)hipFree(dr);
  if (e != hipSuccess) return fail(TM_ERR_HIP, "launch_conv27 (16-bit): %s", hipGetErrorString(e));
  if (e2 != hipSuccess) return fail(TM_ERR_HIP, "conv27 (16-bit) execution: %s", hipGetErrorString(e2));
  return TM_OK;
}
extern "C" int tm_op_conv1_bf16(const void* x_cb8, const void* w_host, const void* bias_host, void* y_cb8, int N, int Cin,
                                int Cout, int Z, int S, int gelu, int dtype, int waves, const void* res_h16, const void* gate_h16,
                                void* y_h16, void* stream) {
  if (!x_cb8 || !w_host || !bias_host || (!y_cb8 && !y_h16)) return fail(TM_ERR_ARG, "null argument");
  if (!y_cb8) y_cb8 = y_h16;                              // geometry carrier only
  if (!is_h16(dtype)) return fail(TM_ERR_ARG, "dtype must be TM_DTYPE_BF16 or TM_DTYPE_F16");
  if (waves != 0 && waves != 4 && waves != 8) return fail(TM_ERR_ARG, "waves must be 0 (auto), 4 or 8");
  const bool f16 = dtype == TM_DTYPE_F16;
  hipStream_t st = (hipStream_t)stream;
  const int Cbi = (Cin + 7) / 8, Cbe = (Cbi + 1) / 2 * 2, nt64 = (Cout + 63) / 64;
  std::vector<uint16_t> pk(conv1_bf16_pack_elems(Cout, Cbi));
  (f16 ? conv1_f16_pack_host : conv1_bf16_pack_host)((const float*)w_host, Cout, &Cin, 1, pk.data());
  std::vector<float> bp((size_t)nt64 * 64, 0.f);
  memcpy(bp.data(), bias_host, Cout * sizeof(float));
  uint16_t *dw = nullptr, *dx = nullptr;
  float* db = nullptr;
  const long vox = (long)Z * S * S;
  HIP_TRY(hipMalloc((void**)&dw, pk.size() * sizeof(uint16_t)));
  HIP_TRY(hipMalloc((void**)&db, bp.size() * sizeof(float)));
  HIP_TRY(hipMalloc((void**)&dx, (size_t)N * Cbe * vox * 8 * sizeof(uint16_t)));
  HIP_TRY(hipMemcpy(dw, pk.data(), pk.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(db, bp.data(), bp.size() * sizeof(float), hipMemcpyHostToDevice));
  TV x = view_cb8(const_cast<void*>(x_cb8), N, Cin, Z, S, S);
  PrepLaunch P;
  P.nsrc = 1;
  P.src[0].p = x.p; P.src[0].nstride = x.nstride; P.src[0].Cb = x.Cb;
  P.N = N; P.Z = Z; P.S = S; P.h_f16 = f16 ? 1 : 0;
  P.out_h = dx; P.out_h_nstride = (long)Cbe * vox * 8; P.pad_blocks = Cbe - Cbi;
  hipError_t e0 = launch_prep(P, st);
  ConvLaunchH L;
  L.x.p = dx; L.x.N = N; L.x.Cb = Cbe; L.x.C = Cbe * 8; L.x.Z = Z; L.x.H = S; L.x.W = S; L.x.nstride = P.out_h_nstride;
  L.w = dw; L.bias = db; L.Cout = Cout; L.flags = gelu ? EPI_GELU : 0; L.force_waves = waves;
  L.y = view_cb8(y_cb8, N, Cout, Z, S, S);
  TVH resh = as_h(L.y), gateh = as_h(L.y);
  if (res_h16) { resh.p = (uint16_t*)const_cast<void*>(res_h16); L.res_h = &resh; }
  if (gate_h16) { gateh.p = (uint16_t*)const_cast<void*>(gate_h16); L.gate_h = &gateh; }
  if (y_h16) { L.y_h = (uint16_t*)y_h16; L.yh_nstride = L.y.nstride; }
  hipError_t e = (f16 ? launch_conv1_f16 : launch_conv1_bf16)(L, st);
  hipError_t e2 = hipStreamSynchronize(st);
  (void)hipFree(dw); (void)hipFree(db); (void)hipFree(dx);
  if (e0 != hipSuccess) return fail(TM_ERR_HIP, "launch_prep: %s", hipGetErrorString(e0));
  if (e != hipSuccess) return fail(TM_ERR_HIP, "launch_conv1 (16-bit): %s", hipGetErrorString(e));
  if (e2 != hipSuccess) return fail(TM_ERR_HIP, "conv1 (16-bit) execution: %s", hipGetErrorString(e2));
  return TM_OK;
}
extern "C" int tm_op_conv1_concat(const void* const* x_cb8, const int* cin, const int* collage, int nsrc, const void* w_host,
                                  const void* bias_host, void* y_cb8, int N, int Cout, int Z, int S, int p1, int p2,
                                  int dtype, int waves, void* stream) {
  if (!x_cb8 || !cin || !collage || !w_host || !bias_host || !y_cb8 || nsrc < 1 || nsrc > 3) return fail(TM_ERR_ARG, "bad argument");
  if (!is_h16(dtype)) return fail(TM_ERR_ARG, "dtype must be TM_DTYPE_BF16 or TM_DTYPE_F16");
  const bool f16 = dtype == TM_DTYPE_F16;
  hipStream_t st = (hipStream_t)stream;
  bool any_col = false;
  for (int i = 0; i < nsrc; ++i) any_col = any_col || collage[i];
  const int q = any_col ? (p1 - 1) * (p2 - 1) : 1;
  if (any_col && (p1 < 2 || p2 < 2 || N % q)) return fail(TM_ERR_ARG, "collage needs N = b * (p1-1) * (p2-1)");
  const int Nsrc_col = any_col ? N / q * p1 * p2 : N;      // a collaged source lives on the (p1 x p2) grid
  std::vector<int> seg(cin, cin + nsrc);
  int Cbi = 0;
  for (int c : seg) Cbi += (c + 7) / 8;
  const int Cbe = (Cbi + 1) / 2 * 2, nt64 = (Cout + 63) / 64;
  std::vector<uint16_t> pk(conv1_bf16_pack_elems(Cout, Cbi));
  (f16 ? conv1_f16_pack_host : conv1_bf16_pack_host)((const float*)w_host, Cout, seg.data(), nsrc, pk.data());
  std::vector<float> bp((size_t)nt64 * 64, 0.f);
  memcpy(bp.data(), bias_host, Cout * sizeof(float));
  const long vox = (long)Z * S * S;
  uint16_t *dw = nullptr, *dx[3] = {nullptr, nullptr, nullptr};
  float* db = nullptr;
  HIP_TRY(hipMalloc((void**)&dw, pk.size() * sizeof(uint16_t)));
  HIP_TRY(hipMalloc((void**)&db, bp.size() * sizeof(float)));
  HIP_TRY(hipMemcpy(dw, pk.data(), pk.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(db, bp.data(), bp.size() * sizeof(float), hipMemcpyHostToDevice));
  ConvLaunchH L;
  hipError_t e0 = hipSuccess;
  for (int i = 0; i < nsrc && e0 == hipSuccess; ++i) {
    const int Ni = collage[i] ? Nsrc_col : N, cb = (seg[i] + 7) / 8;
    HIP_TRY(hipMalloc((void**)&dx[i], (size_t)Ni * cb * vox * 8 * sizeof(uint16_t)));
    TV x = view_cb8(const_cast<void*>(x_cb8[i]), Ni, seg[i], Z, S, S);
    PrepLaunch P;
    P.nsrc = 1;
    P.src[0].p = x.p; P.src[0].nstride = x.nstride; P.src[0].Cb = x.Cb;
    P.N = Ni; P.Z = Z; P.S = S; P.h_f16 = f16 ? 1 : 0;
    P.out_h = dx[i]; P.out_h_nstride = (long)cb * vox * 8;
    e0 = launch_prep(P, st);
    L.xs[i].p = dx[i]; L.xs[i].N = Ni; L.xs[i].Cb = cb; L.xs[i].C = cb * 8; L.xs[i].Z = Z; L.xs[i].H = S; L.xs[i].W = S;
    L.xs[i].nstride = P.out_h_nstride;
    L.xs_collage[i] = collage[i] ? 1 : 0;
  }
  L.nsrc = nsrc; L.p1 = p1; L.p2 = p2;
  L.x.p = nullptr; L.x.N = N; L.x.Cb = Cbe; L.x.C = Cbe * 8; L.x.Z = Z; L.x.H = S; L.x.W = S; L.x.nstride = 0;
  L.w = dw; L.bias = db; L.Cout = Cout; L.force_waves = waves;
  L.y = view_cb8(y_cb8, N, Cout, Z, S, S);
  hipError_t e = e0 == hipSuccess ? (f16 ? launch_conv1_f16 : launch_conv1_bf16)(L, st) : e0;
  hipError_t e2 = hipStreamSynchronize(st);
  (void)hipFree(dw); (void)hipFree(db);
  for (int i = 0; i < 3; ++i) if (dx[i]) (void)hipFree(dx[i]);
  if (e != hipSuccess) return fail(TM_ERR_HIP, "conv1 concat launch: %s", hipGetErrorString(e));
  if (e2 != hipSuccess) return fail(TM_ERR_HIP, "conv1 concat execution: %s", hipGetErrorString(e2));
  return TM_OK;
}
extern "C" int tm_op_prep_h16(const void* const* src_h16, const int* src_c, const int* collage, int nsrc, int N, int Z, int S,
                              int p1, int p2, int up2, const void* norm_w_dev, int c_real, int mod, const void* mod_scale,
                              const void* mod_shift, long mod_stride, int per_image, int act, int dtype, int variant,
                              void* out_h16, void* raw_h16, int iters, float* elapsed_ms, void* stream) {
  if (!src_h16 || !src_c || !collage || !out_h16 || nsrc < 1 || nsrc > 3 || iters < 1) return fail(TM_ERR_ARG, "bad argument");
  if (!is_h16(dtype)) return fail(TM_ERR_ARG, "dtype must be TM_DTYPE_BF16 or TM_DTYPE_F16");
  if (mod != MOD_NONE && (!mod_scale || !mod_shift)) return fail(TM_ERR_ARG, "modulation tensors missing");
  hipStream_t st = (hipStream_t)stream;
  bool any_col = false;
  for (int i = 0; i < nsrc; ++i) any_col = any_col || collage[i];
  const int q = any_col ? (p1 - 1) * (p2 - 1) : 1;
  if (any_col && (p1 < 2 || p2 < 2 || N % q)) return fail(TM_ERR_ARG, "collage needs N = b * (p1-1) * (p2-1)");
  if (up2 < 0 || up2 > 2) return fail(TM_ERR_ARG, "up2: 0 same, 1 nearest x2, 2 = 2 x 2 average (Downsample)");
  if (up2 == 1 && (any_col || (S & 1))) return fail(TM_ERR_ARG, "up2 takes plain sources and an even S");
  if (up2 == 2 && (any_col || nsrc != 1 || mod != MOD_NONE)) return fail(TM_ERR_ARG, "the downsample form takes one plain source, no modulation");
  const int Ss = up2 == 1 ? S / 2 : (up2 == 2 ? 2 * S : S);
  PrepLaunch P;
  P.nsrc = nsrc;
  int cbtot = 0;
  for (int i = 0; i < nsrc; ++i) {
    const int cb = (src_c[i] + 7) / 8;
    P.src[i].p = (const float*)src_h16[i]; P.src[i].Cb = cb; P.src[i].collage = collage[i] ? 1 : 0;
    P.src[i].nstride = (long)cb * Z * Ss * Ss * 8;
    cbtot += cb;
  }
  const int cbe = (cbtot + 1) / 2 * 2;
  P.src_h = 1; P.h_f16 = dtype == TM_DTYPE_F16;
  P.resample = up2 == 1 ? RS_UP2 : (up2 == 2 ? RS_DOWN2 : RS_SAME); P.N = N; P.Z = Z; P.S = S; P.p1 = p1; P.p2 = p2;
  P.norm_w = (const float*)norm_w_dev; P.inv_c = 1.0f / (float)c_real; P.act = act; P.per_image = per_image > 0 ? per_image : 1;
  P.mod = mod; P.mod_stride = mod_stride;
  if (mod == MOD_IMAGE) { P.mod_scale = (const float*)mod_scale; P.mod_shift = (const float*)mod_shift; }
  if (mod == MOD_VOXEL) { P.mod_scale_h = (const uint16_t*)mod_scale; P.mod_shift_h = (const uint16_t*)mod_shift; }
  P.out_h = (uint16_t*)out_h16; P.out_h_nstride = (long)cbe * Z * S * S * 8; P.pad_blocks = cbe - cbtot;
  if (raw_h16) { P.raw_h = (uint16_t*)raw_h16; P.raw_h_nstride = P.out_h_nstride; }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (elapsed_ms) { HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1)); }
  set_prep_variant(variant);
  hipError_t e = launch_prep(P, st);                                   // warm-up / the result
  if (elapsed_ms && e == hipSuccess) e = hipEventRecord(e0, st);
  for (int i = 1; i < iters && e == hipSuccess; ++i) e = launch_prep(P, st);
  if (elapsed_ms && e == hipSuccess) e = hipEventRecord(e1, st);
  set_prep_variant(0);
  hipError_t e2 = hipStreamSynchronize(st);
  if (elapsed_ms) {
    if (e == hipSuccess && e2 == hipSuccess && iters > 1) { (void)hipEventElapsedTime(elapsed_ms, e0, e1); *elapsed_ms /= (float)(iters - 1); }
    else *elapsed_ms = 0.f;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  }
  if (e != hipSuccess) return fail(TM_ERR_HIP, "prep launch: %s", hipGetErrorString(e));
  if (e2 != hipSuccess) return fail(TM_ERR_HIP, "prep execution: %s", hipGetErrorString(e2));
  return TM_OK;
}
extern "C" int tm_op_window_attn(const void* q_cb8, const void* k_cb8, const void* v_cb8, const void* qw_dev,
                                 const void* kw_dev, void* out, int N, int C, int Z, int S, int dtype, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (C % 64) return fail(TM_ERR_ARG, "C must be a multiple of 64");
  TV q = view_cb8(const_cast<void*>(q_cb8), N, C, Z, S, S), k = view_cb8(const_cast<void*>(k_cb8), N, C, Z, S, S);
  TV v = view_cb8(const_cast<void*>(v_cb8), N, C, Z, S, S);
  if (dtype == TM_DTYPE_F32) {
    TV o = view_cb8(out, N, C, Z, S, S);
    HIP_TRY(launch_window_attn(q, k, v, (const float*)qw_dev, (const float*)kw_dev, o, st));
    HIP_TRY(hipStreamSynchronize(st));
    return TM_OK;
  }
  const int cb = C / 8;
  const long ns = (long)cb * Z * S * S * 8;
  uint16_t* buf = nullptr;
  HIP_TRY(hipMalloc((void**)&buf, (size_t)3 * N * ns * sizeof(uint16_t)));
  TVH h[3];
  const TV* src[3] = {&q, &k, &v};
  hipError_t e = hipSuccess;
  for (int i = 0; i < 3 && e == hipSuccess; ++i) {
    h[i].p = buf + (size_t)i * N * ns; h[i].N = N; h[i].C = C; h[i].Cb = cb; h[i].Z = Z; h[i].H = S; h[i].W = S; h[i].nstride = ns;
    PrepLaunch P;
    P.nsrc = 1;
    P.src[0].p = src[i]->p; P.src[0].nstride = src[i]->nstride; P.src[0].Cb = cb;
    P.N = N; P.Z = Z; P.S = S;
    P.out_h = h[i].p; P.out_h_nstride = ns;
    e = launch_prep(P, st);
  }
  TVH o = h[0];
  o.p = (uint16_t*)out;
  if (e == hipSuccess) e = launch_window_attn_bf16(h[0], h[1], h[2], (const float*)qw_dev, (const float*)kw_dev, o, st);
  hipError_t e2 = hipStreamSynchronize(st);
  (void)hipFree(buf);
  if (e != hipSuccess) return fail(TM_ERR_HIP, "window attention launch: %s", hipGetErrorString(e));
  if (e2 != hipSuccess) return fail(TM_ERR_HIP, "window attention execution: %s", hipGetErrorString(e2));
  return TM_OK;
}
extern "C" int tm_op_conv_direct(const void* x, const void* w_host, const void* bias_host, void* y, int N, int Cin,
                                 int Cout, int Zin, int S, int kz, int ky, int kx, int pz, int py, int px, int silu_in,
                                 int up2_out, void* stream) {
  const int taps = kz * ky * kx, Cop = (Cout + 7) / 8 * 8;
  const int Zout = Zin + 2 * pz - kz + 1;
  if (Zout < 1 || py != ky / 2 || px != kx / 2) return fail(TM_ERR_ARG, "unsupported geometry");
  std::vector<float> wt((size_t)taps * Cin * Cop, 0.f);
  const float* w = (const float*)w_host;
  for (int co = 0; co < Cout; ++co)
    for (int ci = 0; ci < Cin; ++ci)
      for (int t = 0; t < taps; ++t) wt[((size_t)t * Cin + ci) * Cop + co] = w[((size_t)co * Cin + ci) * taps + t];
  float *dw = nullptr, *db = nullptr;
  HIP_TRY(hipMalloc((void**)&dw, wt.size() * sizeof(float)));
  HIP_TRY(hipMalloc((void**)&db, Cout * sizeof(float)));
  HIP_TRY(hipMemcpy(dw, wt.data(), wt.size() * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(db, bias_host, Cout * sizeof(float), hipMemcpyHostToDevice));
  DirectLaunch L;
  L.x = (const float*)x; L.ax = acc_ncdhw(Cin, Zin, S, S);
  const int So = up2_out ? 2 * S : S;
  L.y = (float*)y; L.ay = acc_ncdhw(Cout, Zout, So, So);
  L.w = dw; L.bias = db; L.N = N; L.Cin = Cin; L.Cout = Cout; L.Zin = Zin; L.Zout = Zout; L.S = S;
  L.kz = kz; L.ky = ky; L.kx = kx; L.pz = pz; L.py = py; L.px = px; L.silu_in = silu_in; L.up2_out = up2_out;
  hipError_t e = launch_conv_direct(L, (hipStream_t)stream);
  hipError_t e2 = hipStreamSynchronize((hipStream_t)stream);
  (void)hipFree(dw); (void)hipFree(db);
  if (e != hipSuccess) return fail(TM_ERR_HIP, "launch_conv_direct: %s", hipGetErrorString(e));
  if (e2 != hipSuccess) return fail(TM_ERR_HIP, "conv_direct execution: %s", hipGetErrorString(e2));
  return TM_OK;
}

// ------------------------------------------------------------------------------------------
// training slice: forward with dropout + backward of one ResBlock's pieces
// ------------------------------------------------------------------------------------------
// host vector -> device copy living until `free_all`
struct DevTmp {
  std::vector<void*> ptrs;
  float* up(const float* host, size_t n, size_t n_alloc = 0) {
    float* d = nullptr;
    if (n_alloc < n) n_alloc = n;
    if (hipMalloc((void**)&d, n_alloc * sizeof(float)) != hipSuccess) return nullptr;
    ptrs.push_back(d);
    if (n_alloc > n && hipMemset(d, 0, n_alloc * sizeof(float)) != hipSuccess) return nullptr;
    if (n && hipMemcpy(d, host, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
  }
  ~DevTmp() { for (void* p : ptrs) (void)hipFree(p); }
};
// per-channel vector [rows][C] -> [rows][Cb*8] zero padded (host)
static std::vector<float> pad_rows(const float* v, int rows, int C, int Cp) {
  std::vector<float> o((size_t)rows * Cp, 0.f);
  for (int r = 0; r < rows; ++r) memcpy(o.data() + (size_t)r * Cp, v + (size_t)r * C, C * sizeof(float));
  return o;
}

extern "C" int tm_op_prep_train(const void* x_cb8, const void* norm_w_host, const void* scale_host, const void* shift_host,
                                const void* mask_cb8, float drop_scale, int per_image, void* y_cb8, int N, int C, int Z, int S,
                                void* stream) {
  if (!x_cb8 || !norm_w_host || !y_cb8 || per_image < 1) return fail(TM_ERR_ARG, "bad argument");
  const int Cb = (C + 7) / 8, Cp = Cb * 8, nimg = (N + per_image - 1) / per_image;
  DevTmp tmp;
  const std::vector<float> wp = pad_rows((const float*)norm_w_host, 1, C, Cp);
  const float* dw = tmp.up(wp.data(), Cp);
  const float *dsc = nullptr, *dsh = nullptr;
  if (scale_host) {
    const std::vector<float> a = pad_rows((const float*)scale_host, nimg, C, Cp), b = pad_rows((const float*)shift_host, nimg, C, Cp);
    dsc = tmp.up(a.data(), a.size()); dsh = tmp.up(b.data(), b.size());
    if (!dsc || !dsh) return fail(TM_ERR_HIP, "device allocation failed");
  }
  if (!dw) return fail(TM_ERR_HIP, "device allocation failed");
  TV x = view_cb8(const_cast<void*>(x_cb8), N, C, Z, S, S), y = view_cb8(y_cb8, N, C, Z, S, S);
  PrepLaunch P;
  P.nsrc = 1;
  P.src[0].p = x.p; P.src[0].nstride = x.nstride; P.src[0].Cb = x.Cb;
  P.N = N; P.Z = Z; P.S = S; P.norm_w = dw; P.inv_c = 1.0f / (float)C; P.act = 1; P.per_image = per_image;
  if (dsc) { P.mod = MOD_IMAGE; P.mod_scale = dsc; P.mod_shift = dsh; P.mod_stride = Cp; }
  if (mask_cb8) { P.drop_mask = (const float*)mask_cb8; P.drop_ns = x.nstride; P.drop_scale = drop_scale; }
  P.out = y.p; P.out_nstride = y.nstride;
  hipError_t e = launch_prep(P, (hipStream_t)stream);
  hipError_t e2 = hipStreamSynchronize((hipStream_t)stream);
  if (e != hipSuccess || e2 != hipSuccess) return fail(TM_ERR_HIP, "prep (training forward): %s", hipGetErrorString(e != hipSuccess ? e : e2));
  return TM_OK;
}

extern "C" int tm_op_prep_bwd(const void* x_cb8, const void* g_cb8, const void* norm_w_host, const void* scale_host,
                              const void* shift_host, const void* mask_cb8, float drop_scale, int per_image, void* dx_cb8,
                              void* dw_host, void* dscale_host, void* dshift_host, int N, int C, int Z, int S, void* stream) {
  if (!x_cb8 || !g_cb8 || !norm_w_host || !dx_cb8 || !dw_host || per_image < 1) return fail(TM_ERR_ARG, "bad argument");
  if (scale_host && (!shift_host || !dscale_host || !dshift_host)) return fail(TM_ERR_ARG, "scale without shift / gradient outputs");
  const int Cb = (C + 7) / 8, Cp = Cb * 8, nimg = (N + per_image - 1) / per_image;
  DevTmp tmp;
  const std::vector<float> wp = pad_rows((const float*)norm_w_host, 1, C, Cp);
  const float* dwt = tmp.up(wp.data(), Cp);
  float* ddw = tmp.up(nullptr, 0, Cp);
  const float *dsc = nullptr, *dsh = nullptr;
  float *ddsc = nullptr, *ddsh = nullptr;
  if (scale_host) {
    const std::vector<float> a = pad_rows((const float*)scale_host, nimg, C, Cp), b = pad_rows((const float*)shift_host, nimg, C, Cp);
    dsc = tmp.up(a.data(), a.size()); dsh = tmp.up(b.data(), b.size());
    ddsc = tmp.up(nullptr, 0, (size_t)(nimg + 1) * Cp); ddsh = tmp.up(nullptr, 0, (size_t)(nimg + 1) * Cp);
    if (!dsc || !dsh || !ddsc || !ddsh) return fail(TM_ERR_HIP, "device allocation failed");
  }
  if (!dwt || !ddw) return fail(TM_ERR_HIP, "device allocation failed");
  TV x = view_cb8(const_cast<void*>(x_cb8), N, C, Z, S, S);
  hipStream_t st = (hipStream_t)stream;
  float* scratch = tmp.up(nullptr, 0, prep_bwd_scratch_floats(N, Cb, Z, S, scale_host != nullptr));
  if (!scratch) return fail(TM_ERR_HIP, "device allocation failed");
  hipError_t e = launch_prep_bwd(x.p, x.nstride, (const float*)g_cb8, x.nstride, (const float*)mask_cb8, x.nstride, drop_scale, dwt,
                                 dsc, dsh, Cp, per_image, (float*)dx_cb8, x.nstride, ddw, ddsc, ddsh, N, Cb, C, Z, S, scratch, st);
  hipError_t e2 = hipStreamSynchronize(st);
  if (e != hipSuccess || e2 != hipSuccess) return fail(TM_ERR_HIP, "prep backward: %s", hipGetErrorString(e != hipSuccess ? e : e2));
  std::vector<float> h(Cp);
  HIP_TRY(hipMemcpy(h.data(), ddw, Cp * sizeof(float), hipMemcpyDeviceToHost));
  memcpy(dw_host, h.data(), C * sizeof(float));
  if (scale_host) {
    std::vector<float> hs((size_t)nimg * Cp), hh((size_t)nimg * Cp);
    HIP_TRY(hipMemcpy(hs.data(), ddsc, hs.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(hh.data(), ddsh, hh.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (int r = 0; r < nimg; ++r) {
      memcpy((float*)dscale_host + (size_t)r * C, hs.data() + (size_t)r * Cp, C * sizeof(float));
      memcpy((float*)dshift_host + (size_t)r * C, hh.data() + (size_t)r * Cp, C * sizeof(float));
    }
  }
  return TM_OK;
}

// dX of Conv3d(k = 3x3x3 pad 1 | 1x1x1), stride 1: the forward MFMA conv of dY with the kernel flipped in z, y, x and
// cin <-> cout transposed (w_host [Cout][Cin][taps] as in the reference state_dict)
// ---- AttnBlock training pieces (teramind_amd.training.AttnBlockTrain composes them) ----
extern "C" int tm_op_ew(int op, const void* a, const void* b, const void* c, void* o1, void* o2, long n, void* stream) {
  if (op < 0 || op > 8 || !a || !o1 || n < 0) return fail(TM_ERR_ARG, "bad argument");
  if ((op == 0 || op == 1) && (!b || !c)) return fail(TM_ERR_ARG, "op %d needs b and c", op);
  if ((op == 3 || op == 5 || op == 6) && !b) return fail(TM_ERR_ARG, "op %d needs b", op);
  if (op == 1 && !o2) return fail(TM_ERR_ARG, "op 1 needs two outputs");
  hipError_t e = launch_ew(op, (const float*)a, (const float*)b, (const float*)c, (float*)o1, (float*)o2, n, (hipStream_t)stream);
  hipError_t e2 = hipStreamSynchronize((hipStream_t)stream);
  if (e != hipSuccess || e2 != hipSuccess) return fail(TM_ERR_HIP, "elementwise op: %s", hipGetErrorString(e != hipSuccess ? e : e2));
  return TM_OK;
}

extern "C" int tm_op_modnorm(const void* x_cb8, const void* norm_w_host, const void* scale_cb8, const void* shift_cb8, void* y_cb8, int N,
                             int C, int Z, int S, void* stream) {
  if (!x_cb8 || !norm_w_host || !scale_cb8 || !shift_cb8 || !y_cb8) return fail(TM_ERR_ARG, "bad argument");
  const int Cb = (C + 7) / 8, Cp = Cb * 8;
  DevTmp tmp;
  const std::vector<float> wp = pad_rows((const float*)norm_w_host, 1, C, Cp);
  const float* dw = tmp.up(wp.data(), Cp);
  if (!dw) return fail(TM_ERR_HIP, "device allocation failed");
  TV x = view_cb8(const_cast<void*>(x_cb8), N, C, Z, S, S), y = view_cb8(y_cb8, N, C, Z, S, S);
  PrepLaunch P;
  P.nsrc = 1;
  P.src[0].p = x.p; P.src[0].nstride = x.nstride; P.src[0].Cb = x.Cb;
  P.N = N; P.Z = Z; P.S = S; P.norm_w = dw; P.inv_c = 1.0f / (float)C; P.act = 0;
  P.mod = MOD_VOXEL; P.mod_scale = (const float*)scale_cb8; P.mod_shift = (const float*)shift_cb8; P.mod_stride = x.nstride;
  P.out = y.p; P.out_nstride = y.nstride;
  hipError_t e = launch_prep(P, (hipStream_t)stream);
  hipError_t e2 = hipStreamSynchronize((hipStream_t)stream);
  if (e != hipSuccess || e2 != hipSuccess) return fail(TM_ERR_HIP, "modulate(norm): %s", hipGetErrorString(e != hipSuccess ? e : e2));
  return TM_OK;
}

extern "C" int tm_op_modnorm_bwd(const void* x_cb8, const void* g_cb8, const void* norm_w_host, const void* scale_cb8, void* dx_cb8,
                                 void* dscale_cb8, void* dshift_cb8, void* dw_host, int N, int C, int Z, int S, void* stream) {
  if (!x_cb8 || !g_cb8 || !norm_w_host || !scale_cb8 || !dx_cb8 || !dscale_cb8 || !dshift_cb8 || !dw_host)
    return fail(TM_ERR_ARG, "bad argument");
  const int Cb = (C + 7) / 8, Cp = Cb * 8;
  DevTmp tmp;
  const std::vector<float> wp = pad_rows((const float*)norm_w_host, 1, C, Cp);
  const float* dwt = tmp.up(wp.data(), Cp);
  float* ddw = tmp.up(nullptr, 0, Cp);
  const long vox = (long)N * Z * S * S;
  float* scratch = tmp.up(nullptr, 0, (size_t)((vox + 63) / 64) * Cp);
  if (!dwt || !ddw || !scratch) return fail(TM_ERR_HIP, "device allocation failed");
  TV x = view_cb8(const_cast<void*>(x_cb8), N, C, Z, S, S);
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = launch_modnorm_bwd(x, (const float*)g_cb8, dwt, (const float*)scale_cb8, (float*)dx_cb8, (float*)dscale_cb8,
                                    (float*)dshift_cb8, ddw, C, scratch, st);
  hipError_t e2 = hipStreamSynchronize(st);
  if (e != hipSuccess || e2 != hipSuccess) return fail(TM_ERR_HIP, "modulate(norm) backward: %s", hipGetErrorString(e != hipSuccess ? e : e2));
  std::vector<float> h(Cp);
  HIP_TRY(hipMemcpy(h.data(), ddw, Cp * sizeof(float), hipMemcpyDeviceToHost));
  memcpy(dw_host, h.data(), C * sizeof(float));
  return TM_OK;
}

extern "C" int tm_op_window_attn_train(const void* q_cb8, const void* k_cb8, const void* v_cb8, const void* qw_host, const void* kw_host,
                                       const void* dout_cb8, void* o_cb8, void* dq_cb8, void* dk_cb8, void* dv_cb8, void* dqw_host,
                                       void* dkw_host, int N, int C, int Z, int S, void* stream) {
  const bool bwd = dout_cb8 != nullptr;
  if (!q_cb8 || !k_cb8 || !v_cb8 || !qw_host || !kw_host) return fail(TM_ERR_ARG, "bad argument");
  if (bwd ? (!dq_cb8 || !dk_cb8 || !dv_cb8 || !dqw_host || !dkw_host) : !o_cb8) return fail(TM_ERR_ARG, "missing output");
  const int T = Z * (S / 2) * (S / 2);
  if ((S & 1) || (T != 32 && T != 64 && T != 128) || C > 512 || C < 1)
    return fail(TM_ERR_ARG, "window of %d tokens / C = %d: the training attention core takes 32, 64 or 128 tokens and C <= 512", T, C);
  const int Cb = (C + 7) / 8, Cp = Cb * 8;
  DevTmp tmp;
  const std::vector<float> qp = pad_rows((const float*)qw_host, 1, C, Cp), kp = pad_rows((const float*)kw_host, 1, C, Cp);
  const float *dqw_in = tmp.up(qp.data(), Cp), *dkw_in = tmp.up(kp.data(), Cp);
  float *gq = nullptr, *gk = nullptr, *scratch = nullptr;
  if (bwd) {
    gq = tmp.up(nullptr, 0, Cp); gk = tmp.up(nullptr, 0, Cp);
    scratch = tmp.up(nullptr, 0, (size_t)2 * N * 4 * Cp);
    if (!gq || !gk || !scratch) return fail(TM_ERR_HIP, "device allocation failed");
  }
  if (!dqw_in || !dkw_in) return fail(TM_ERR_HIP, "device allocation failed");
  TV q = view_cb8(const_cast<void*>(q_cb8), N, C, Z, S, S), k = view_cb8(const_cast<void*>(k_cb8), N, C, Z, S, S),
     v = view_cb8(const_cast<void*>(v_cb8), N, C, Z, S, S);
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = launch_attn_train(q, k, v, dqw_in, dkw_in, (const float*)dout_cb8, (float*)o_cb8, (float*)dq_cb8, (float*)dk_cb8,
                                   (float*)dv_cb8, gq, gk, scratch, bwd, st);
  hipError_t e2 = hipStreamSynchronize(st);
  if (e != hipSuccess || e2 != hipSuccess) return fail(TM_ERR_HIP, "window attention (training): %s", hipGetErrorString(e != hipSuccess ? e : e2));
  if (bwd) {
    std::vector<float> h(Cp);
    HIP_TRY(hipMemcpy(h.data(), gq, Cp * sizeof(float), hipMemcpyDeviceToHost));
    memcpy(dqw_host, h.data(), C * sizeof(float));
    HIP_TRY(hipMemcpy(h.data(), gk, Cp * sizeof(float), hipMemcpyDeviceToHost));
    memcpy(dkw_host, h.data(), C * sizeof(float));
  }
  return TM_OK;
}

extern "C" int tm_op_gemm_f32(const void* A_dev, const void* B_dev, const void* bias_dev, void* C_dev, int M, int N, int K,
                              const long* strides9_host, int batch, int bias_mode, int accumulate, float alpha, void* stream) {
  if (!A_dev || !B_dev || !C_dev || !strides9_host || M < 1 || N < 1 || K < 1 || batch < 1 || bias_mode < 0 || bias_mode > 2 ||
      (bias_mode && !bias_dev))
    return fail(TM_ERR_ARG, "bad argument");
  hipError_t e = launch_gemm_f32((const float*)A_dev, (const float*)B_dev, (const float*)bias_dev, (float*)C_dev, M, N, K, strides9_host,
                                 batch, bias_mode, accumulate, alpha, (hipStream_t)stream);
  hipError_t e2 = hipStreamSynchronize((hipStream_t)stream);
  if (e != hipSuccess || e2 != hipSuccess) return fail(TM_ERR_HIP, "gemm: %s", hipGetErrorString(e != hipSuccess ? e : e2));
  return TM_OK;
}

extern "C" int tm_op_rows(int op, const void* x_dev, const void* w_dev, const void* g_dev, void* y_dev, void* dw_dev, long rows, int D,
                          void* stream) {
  if (op < 0 || op > 3 || !x_dev || !y_dev || rows < 1 || D < 1 || D > 8192) return fail(TM_ERR_ARG, "bad argument");
  if ((op <= 1 && !w_dev) || ((op == 1 || op == 3) && !g_dev) || (op == 1 && !dw_dev)) return fail(TM_ERR_ARG, "op %d: missing operand", op);
  DevTmp tmp;
  float* scratch = nullptr;
  if (op == 1) {
    scratch = tmp.up(nullptr, 0, (size_t)((rows + 3) / 4) * D);
    if (!scratch) return fail(TM_ERR_HIP, "device allocation failed");
  }
  hipError_t e = launch_rows(op, (const float*)x_dev, (const float*)w_dev, (const float*)g_dev, (float*)y_dev, (float*)dw_dev, scratch, rows, D,
                             (hipStream_t)stream);
  hipError_t e2 = hipStreamSynchronize((hipStream_t)stream);
  if (e != hipSuccess || e2 != hipSuccess) return fail(TM_ERR_HIP, "row op: %s", hipGetErrorString(e != hipSuccess ? e : e2));
  return TM_OK;
}

extern "C" int tm_op_resample(const void* x_cb8, void* y_cb8, int N, int C, int Z, int S_out, int mode, void* stream) {
  if (!x_cb8 || !y_cb8 || (mode != 1 && mode != 2) || (mode == 1 && (S_out & 1))) return fail(TM_ERR_ARG, "bad argument");
  const int S_in = mode == 1 ? S_out / 2 : S_out * 2;
  TV x = view_cb8(const_cast<void*>(x_cb8), N, C, Z, S_in, S_in), y = view_cb8(y_cb8, N, C, Z, S_out, S_out);
  PrepLaunch P;
  P.nsrc = 1;
  P.src[0].p = x.p; P.src[0].nstride = x.nstride; P.src[0].Cb = x.Cb;
  P.resample = mode == 1 ? RS_UP2 : RS_DOWN2;
  P.N = N; P.Z = Z; P.S = S_out; P.inv_c = 1.0f / (float)C;
  P.out = y.p; P.out_nstride = y.nstride;
  hipError_t e = launch_prep(P, (hipStream_t)stream);
  hipError_t e2 = hipStreamSynchronize((hipStream_t)stream);
  if (e != hipSuccess || e2 != hipSuccess) return fail(TM_ERR_HIP, "resample: %s", hipGetErrorString(e != hipSuccess ? e : e2));
  return TM_OK;
}

extern "C" int tm_op_sumsq(const void* x_dev, long n, float* out_host, void* stream) {
  if (!x_dev || n < 1 || !out_host) return fail(TM_ERR_ARG, "bad argument");
  const int nwg = (int)std::min<long>(1024, (n + 255) / 256);
  DevTmp tmp;
  float* scratch = tmp.up(nullptr, 0, (size_t)nwg + 8);
  if (!scratch) return fail(TM_ERR_HIP, "device allocation failed");
  hipError_t e = launch_sumsq((const float*)x_dev, n, scratch + nwg, scratch, nwg, (hipStream_t)stream);
  hipError_t e2 = hipStreamSynchronize((hipStream_t)stream);
  if (e != hipSuccess || e2 != hipSuccess) return fail(TM_ERR_HIP, "sum of squares: %s", hipGetErrorString(e != hipSuccess ? e : e2));
  HIP_TRY(hipMemcpy(out_host, scratch + nwg, sizeof(float), hipMemcpyDeviceToHost));
  return TM_OK;
}

extern "C" int tm_op_adam(void* p_dev, const void* g_dev, void* m_dev, void* v_dev, long n, float lr, float beta1, float beta2, float eps,
                          float weight_decay, int step, float grad_scale, void* stream) {
  if (!p_dev || !g_dev || !m_dev || !v_dev || n < 1 || step < 1) return fail(TM_ERR_ARG, "bad argument");
  hipError_t e = launch_adam((float*)p_dev, (const float*)g_dev, (float*)m_dev, (float*)v_dev, n, lr, beta1, beta2, eps, weight_decay, step,
                             grad_scale, (hipStream_t)stream);
  hipError_t e2 = hipStreamSynchronize((hipStream_t)stream);
  if (e != hipSuccess || e2 != hipSuccess) return fail(TM_ERR_HIP, "adam: %s", hipGetErrorString(e != hipSuccess ? e : e2));
  return TM_OK;
}

extern "C" int tm_op_conv_dgrad(const void* dy_cb8, const void* w_host, void* dx_cb8, int N, int Cin, int Cout, int Z, int S,
                                int ksize, void* stream) {
  if (!dy_cb8 || !w_host || !dx_cb8 || (ksize != 1 && ksize != 3)) return fail(TM_ERR_ARG, "bad argument");
  const int taps = ksize == 1 ? 1 : 27;
  const float* w = (const float*)w_host;
  std::vector<float> wt((size_t)Cin * Cout * taps);
  for (int co = 0; co < Cout; ++co)
    for (int ci = 0; ci < Cin; ++ci)
      for (int t = 0; t < taps; ++t) wt[((size_t)ci * Cout + co) * taps + (taps - 1 - t)] = w[((size_t)co * Cin + ci) * taps + t];
  std::vector<float> zb(Cin, 0.f);
  return tm_op_conv_mfma(dy_cb8, wt.data(), zb.data(), dx_cb8, N, Cout, Cin, Z, S, ksize, ZM_PAD1, 0, 0, stream);
}

// dW [Cout][Cin][taps] and db [Cout] (HOST outputs) of the same convs from the forward input x and dY
extern "C" int tm_op_conv_wgrad(const void* x_cb8, const void* dy_cb8, void* dw_host, void* db_host_or_null, int N, int Cin,
                                int Cout, int Z, int S, int ksize, void* stream) {
  if (!x_cb8 || !dy_cb8 || !dw_host || (ksize != 1 && ksize != 3)) return fail(TM_ERR_ARG, "bad argument");
  if (Z < 1 || Z > 4) return fail(TM_ERR_ARG, "weight gradient: Z must be 1 .. 4 (the kernel stages up to four z planes)");
  const int taps = ksize == 1 ? 1 : 27;
  hipStream_t st = (hipStream_t)stream;
  TV x = view_cb8(const_cast<void*>(x_cb8), N, Cin, Z, S, S), dy = view_cb8(const_cast<void*>(dy_cb8), N, Cout, Z, S, S);
  DevTmp tmp;
  const size_t nw = (size_t)Cout * Cin * taps;
  float* ddw = tmp.up(nullptr, 0, nw);
  float* ddb = tmp.up(nullptr, 0, (size_t)dy.Cb * 8);
  if (!ddw || !ddb) return fail(TM_ERR_HIP, "device allocation failed");
  hipError_t e = launch_conv_wgrad(x, dy, ddw, Cin, Cout, taps, st);
  if (e == hipSuccess && db_host_or_null) e = launch_chan_sum(dy, ddb, Cout, st);
  hipError_t e2 = hipStreamSynchronize(st);
  if (e != hipSuccess || e2 != hipSuccess) return fail(TM_ERR_HIP, "conv wgrad: %s", hipGetErrorString(e != hipSuccess ? e : e2));
  HIP_TRY(hipMemcpy(dw_host, ddw, nw * sizeof(float), hipMemcpyDeviceToHost));
  if (db_host_or_null) HIP_TRY(hipMemcpy(db_host_or_null, ddb, Cout * sizeof(float), hipMemcpyDeviceToHost));
  return TM_OK;
}
