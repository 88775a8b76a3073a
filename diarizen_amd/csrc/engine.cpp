// engine.cpp — the C ABI of include/dzn.h: weight ingest (fold + pack + pad), workspace, and the
// host-side orchestration of the gfx950 kernels for the two device stages of the DiariZen
// pipeline: the WavLM + Conformer segmentation forward and the ResNet34 embedding forward.
//
// Data layout in HBM (all activations fp32, channels-last, feature dims zero-padded to
// multiples of 32 so every contraction has K % 32 == 0 and 16-byte aligned rows):
//   conv stack   [B, T_i, Cp_i]         — conv i reads conv i-1 as overlapping rows (lda = s*Cp)
//   encoder      x, y, ws [B*L, D]      — residual stream, LN output, layer-weighted sum
//   attention    qkv [B*L, 3*h*64], gate [B*L, H], bias table [H, 2L-1]
//   ResNet       zero-bordered NHWC images [B, H+2, W+2, C]; 3x3 convs address them through
//                per-geometry row-offset tables (no im2col)
// Weights are folded on the host at finalize: weight-norm of the positional conv
// (W2V/components.py:344), eval BatchNorm of the Conformer conv module (conformer.py:205) and of
// every ResNet conv (resnet.py:139-144), q/k/v concatenation, conv weights re-ordered to
// [out][tap][in].
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include "common.h"

#define HIPCHK(x)                                                                        \
  do {                                                                                   \
    hipError_t e_ = (x);                                                                 \
    if (e_ != hipSuccess) {                                                              \
      throw EngineError(e_ == hipErrorOutOfMemory ? DZN_E_NOMEM : DZN_E_HIP,             \
                        std::string(#x) + ": " + hipGetErrorString(e_));                 \
    }                                                                                    \
  } while (0)

namespace {

struct EngineError : std::runtime_error {
  int code;
  EngineError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

// Makes the handle's device current for the duration of a C-ABI call and restores the caller's
// (lazy allocations, table uploads and kernel launches must not land on whatever device the calling
// thread happens to have current — e.g. DiariZenPipeline(device="cuda:1") in a process at device 0).

struct HostT {
  std::vector<float> v;
  std::vector<int64_t> shape;
};

struct Lin {       // y = x W^T + b with W [N, K] (rows zero-padded to Np, cols to Kp)
  float* W = nullptr;
  u16* W16 = nullptr;
  u16* W3 = nullptr;   // exact 3-way bf16 split planes (DZN_PREC_F32_SPLIT), gemm_split.hip layout
  float* b = nullptr;
  u16* W2h = nullptr;  // two-term fp16 planes of W * 2^e_row (DZN_PREC_F32_H2) ...
  float* wsc = nullptr;   // ... and 2^-e_row per output row (dzn_gemm_desc.col_scale)
  float* csum = nullptr;  // LayerNorm folded into W (make_lin_ln): column sums of W diag(gamma), dzn_gemm_desc.ln_colsum
  unsigned char* Wmx = nullptr;   // DZN_PREC_F16: planes of the reduced-precision contraction (gemm_mx.hip: fp16 hi | fp8 hi, fp8 lo) ...
  float* wsc_mx = nullptr;        // ... and their inverse row scales (dzn_gemm_desc.col_scale_mx)
  int N = 0, K = 0;    // padded sizes
  int Nt = 0, Kt = 0;  // reference (un-padded) sizes, for algorithmic flop accounting
};

struct LNp {
  float* g = nullptr;
  float* b = nullptr;
  int C = 0;
};

struct EncLayer {
  bool attn = false, ffn = false;
  int h = 0, F = 0, Fp = 0;
  LNp ln1, ln2;
  Lin qkv, out, f1, f2;
  float *Wg = nullptr, *bg = nullptr, *cst = nullptr;
  int32_t* head_idx = nullptr;
};

struct ConfLayer {
  LNp ffn1_ln, ffn2_ln, mha_ln, conv_ln, out_ln;
  Lin ffn1_w1, ffn1_w2, ffn2_w1, ffn2_w2, qkv, o, pw1, pw2;
  float *dw = nullptr, *dwb = nullptr;
};

struct ResConv {
  Lin l;
  int cin = 0, cout = 0, ksz = 3, stride = 1;
  float l1max = 0.f, bmax = 0.f;   // max_oc sum_k |W[oc][k]| and max_oc |shift[oc]| of the folded conv: |out| <= amax(in) l1max + bmax
};
struct ResBlock {
  ResConv c1, c2, sc;
  bool has_sc = false;
};

std::string last_create_error;

// contraction classes of the segmentation model, by the name their call site passes to gemm(): bit index of the
// DZN_F16_KEEP2 mask (DZN_PREC_F16: classes whose bit is set keep two fp16 terms per operand)
const char* const F16_CLASSES[] = {"conv gemm", "feature projection", "pos conv", "qkv", "out_proj", "ffn1", "ffn2", "proj",
                                   "conf ffn w1", "conf ffn w2", "conf qkv", "conf out", "conf pw1", "conf pw2"};
int f16_class(const char* what) {
  for (int i = 0; i < (int)(sizeof(F16_CLASSES) / sizeof(F16_CLASSES[0])); ++i)
    if (!strcmp(what, F16_CLASSES[i])) return i;
  return -1;
}

}  // namespace

struct dzn_handle {
  dzn_config cfg{};
  int device = 0;  // HIP device current at dzn_create: every entry point makes it current again
  std::string err;
  std::map<std::string, HostT> sd;
  std::set<std::string> used;
  int ignored = 0;
  bool finalized = false;
  bool debug = false;
  std::vector<void*> allocs;
  int64_t bytes = 0;
  std::map<std::string, std::vector<float>> taps;

  // ---- segmentation: geometry ----
  int nconv = 0;
  int C[DZN_MAX_CONV]{}, Cp[DZN_MAX_CONV]{};
  int D = 0, H = 0, A = 0, Fh = 0;
  // ---- segmentation: weights ----
  float* conv0_w = nullptr;
  float* conv0_lnq = nullptr;
  u16* conv1_W2n = nullptr;     // conv1's fp16 planes in natural k order + row scales: the fused conv0 -> conv1 kernel
  float* conv1_wscn = nullptr;   // [10 + 100]: mean and covariance over channels of conv0's taps (frontend.hip)
  LNp conv_ln[DZN_MAX_CONV];
  Lin conv[DZN_MAX_CONV];
  float* dummy_w = nullptr;
  LNp fp_ln, enc_ln;
  Lin fp, posconv;
  int pos_rowD = 0;               // row width the row-offset table of the positional conv was built for
  int pos_cgp = 0;                // channels per positional-conv group in the pre-split planes (cg rounded up to 32)
  std::vector<EncLayer> layers;
  std::vector<float> rel_embed;  // [num_buckets, H] host
  std::vector<float> wsum_w;
  Lin proj;
  LNp lnorm;
  std::vector<ConfLayer> conf;
  float *cls_w = nullptr, *cls_b = nullptr;
  uint8_t* mapping = nullptr;
  // rel-pos table cache
  int table_L = -1;
  int pos_L = -1;               // geometry baked into pos_rowoff
  int32_t* pos_rowoff = nullptr;  // row m = b*L + t of the positional conv -> element offset into xpad
  u16* xpad3 = nullptr;           // f32s mode: the padded copy as three bf16 planes (split once, read by 128 taps)
  int64_t xpad3_plane = 0;
  float* table = nullptr;
  // ---- segmentation: workspace ----
  int maxT[DZN_MAX_CONV]{};
  int maxL = 0;
  float *stats = nullptr, *gn_stats = nullptr, *bufA = nullptr, *bufB = nullptr, *craw = nullptr;
  float *x = nullptr, *xpad = nullptr, *y = nullptr, *ws = nullptr, *qkv = nullptr, *ao = nullptr,
        *gate = nullptr, *mid = nullptr;
  float *hz = nullptr, *ht = nullptr, *hmid = nullptr, *hv = nullptr;
  // (r6) K / V of the encoder's attention as pre-split fp16 planes + per-(row, head slot) inverse scales, written by the q/k/v
  // contraction's epilogue (csrc/attention_planes.hip); nullptr = fp32 K / V and the in-kernel split of attention_split.hip
  void* kvp = nullptr;
  float* kvs = nullptr;
  int64_t kvp_rows = 0;
  int kvp_ldmax = 0;
  // (r4) pre-norm encoders: every layer writes its output rows into ITS OWN buffer (xl + layer * rows * D) and the
  // layer-weighted sum (model_wavlm_conformer.py:236,253-254) is ONE pass over those buffers after the last layer, instead of
  // a read-modify-write of `ws` in every FFN-output epilogue: 25 reads + 1 write of [rows, D] where there were 25 reads and
  // 25 writes.  nullptr: the fused read-modify-write (post-norm encoders, DZN_NO_WS_DEFER, not enough free memory).
  float* xl = nullptr;
  int64_t xl_elems = 0;     // size xl would have (0: not applicable to this model / switched off)
  float* spart = nullptr;   // [max_batch * maxL][32][2] per-row partial sums left by a producing epilogue (stat_partial)
  float* rstat = nullptr;   // [max_batch * maxL][2] (mean, rstd) of the LayerNorm folded into the next contraction
  // |max| trackers of activation tensors (DZN_PREC_F32_H2), ONE PER WINDOW of the batch (slot * max_batch + b): written
  // by the producer's epilogue / LayerNorm, read by the consuming contraction to scale its fp16 split
  // (gemm_split.hip).  Per window, so a window's result is independent of the batch it runs in.  Zeroed per forward.
  enum { AM_CONVA, AM_CONVB, AM_X, AM_XPAD, AM_Y, AM_MID, AM_QKV, AM_HZ, AM_HMID, AM_IMG0, AM_COUNT = AM_IMG0 + 12 };
  float conv0_bound = 0.f;  // sqrt(C0) max|gamma| + max|beta| >= |GELU(LN(conv0))|: static |max| of conv0's output
  float* amax = nullptr;
  bool fold_ln = false;     // fp32 engine modes: LayerNorms that feed only linears are folded (make_lin_ln)

  // ---- embedding ----
  bool has_emb = false;
  float *hamming = nullptr;
  Lin dft, mel, seg1;
  float *stem_w = nullptr, *stem_b = nullptr;
  std::vector<std::vector<ResBlock>> stages;
  int emb_T = -1;  // geometry currently baked into tables / borders
  int maxTf = 0;
  int sH[4]{}, sW[4]{}, sC[4]{};
  float* sbuf[4][3]{};
  int64_t sbuf_elems[4]{};  // allocation per image at the largest geometry
  int64_t simg[4]{};        // elements per image at the current geometry
  int32_t* tab1[4]{};   // tables of the CURRENT geometry (owned by geoms)
  int32_t* tab2[4]{};
  struct EmbGeom {      // per fbank-frame-count T: row-offset tables stay resident, so alternating window
    int32_t* t1[4]{};   // lengths (full windows / a ragged tail) neither re-upload nor synchronise
    int32_t* t2[4]{};
  };
  std::map<int, EmbGeom> geoms;
  float *frames = nullptr, *spec = nullptr, *pw = nullptr, *fb = nullptr, *pool = nullptr;
  // trunk skip for windows without any active speaker (r3; r4: decided on the device): flag per window, the ascending
  // list of active windows + its length, running totals (windows seen, windows skipped) for dzn_embed_skip_stats
  int *win_flag = nullptr, *win_idx = nullptr, *win_cnt = nullptr;
  long long* emb_totals = nullptr;
  long long emb_dense_windows = 0;   // windows that went through the trunk on the dense path (no subset: nothing to skip)
  bool emb_skip = true;       // DZN_EMB_NO_SKIP (read once, at dzn_create) switches the subset off
  // DZN_PREC_F16 (reduced precision): which contraction classes keep two fp16 terms (F16_CLASSES bit mask; bit 14 = the
  // ResNet stages 2-4), and whether LayerNorm-folded single-term contractions subtract the row mean BEFORE rounding
  // defaults from profiles/r4_f16_sensitivity.json (64 non-degenerate windows vs the f32h engine): the error of the
  // single-term mode is NOT localised — no one class brings max |dlogp| under SURVEY 8d's 5e-2, only two terms everywhere
  // (= f32h) does; the conv stack alone carries ~70 % of the error variance for 4 % of the flops, so it keeps two terms
  // (0.247 -> 0.133, flips 0.52 % -> 0.33 %); centring the LayerNorm-folded split changes nothing (0.228 vs 0.247: the
  // error is plain operand rounding, not the mean * colsum cancellation) and stays off.
  // (r5) bit 14 governs the whole trunk: the stride-2 / shortcut / stage 3-4 contractions (gemm_split.hip NP = 1)
  // AND the fused stride-1 BasicBlocks of stages 1-2 (resblock_fused.hip / resblock_ws.hip, which have a single-term form
  // since r5; in r4 they were two-term whatever the bit said).
  bool fuse_resblock = true;  // DZN_NO_RESBLOCK_FUSION (read once, at dzn_create)
  int resblock_ws = 2;        // DZN_RESBLOCK_WS bit mask: 1 = 32-plane blocks, 2 = 64-plane blocks on the producer / consumer form
  // (r5) with the cross terms in fp8 everywhere else, the Conformer head (bits 8-13, 8 % of the segmentation flops) keeps two
  // fp16 terms as well: max |dlogp| does not move (1.01e-2 vs 1.03e-2) but the decisions next to the classifier do — DER of the
  // mode's RTTM against the fp32 RTTM on the 30 s fixture 0.05 % (one 20 ms frame) instead of 0.17 % (three), inside SURVEY 8d's
  // 0.1 abs; device step 1872 -> 1856 audio-s/s on one box (profiles/r5_reduced_mode_keep2_probe.txt)
  unsigned f16_keep2 = 0x3f01;
  bool f16_center = false;
  // (r5) DZN_PREC_F16: the classes whose bit is set run fp16 hi*hi + the two cross terms in fp8 (gemm_mx.hip) — every linear
  // contraction of the segmentation model: the class sweep of profiles/r4_reduced_mode_emulation.txt shows that any one of them
  // at a single fp16 term costs 0.04-0.21 of max |dlogp| against SURVEY 8d's 5e-2, only the positional conv (bit 2) can stay at
  // one term.  Bit 0 (the conv stack) keeps two fp16 terms (f16_keep2).  DZN_F16_MX overrides (0 = r4's single-term mode).
  unsigned f16_mx = 0x3ffa;
  bool mx_pack = false;   // make_lin also packs the MX planes (set while the segmentation weights are finalized)
};

namespace {

using H = dzn_handle;

// ------------------------------------------------------------------ device memory helpers
template <typename T>
T* dalloc(H* h, int64_t n, bool zero = true) {
  if (n <= 0) n = 1;
  void* p = nullptr;
  HIPCHK(hipMalloc(&p, n * sizeof(T)));
  h->allocs.push_back(p);
  h->bytes += n * (int64_t)sizeof(T);
  if (zero) HIPCHK(hipMemset(p, 0, n * sizeof(T)));
  return reinterpret_cast<T*>(p);
}

float* upload(H* h, const std::vector<float>& v) {
  float* p = dalloc<float>(h, (int64_t)v.size(), false);
  if (!v.empty()) HIPCHK(hipMemcpy(p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
  return p;
}

u16* upload_bf16(H* h, const std::vector<float>& v) {
  std::vector<u16> t(v.size());
  for (size_t i = 0; i < v.size(); ++i) t[i] = f32_to_bf16_host(v[i]);
  u16* p = dalloc<u16>(h, (int64_t)t.size(), false);
  if (!t.empty()) HIPCHK(hipMemcpy(p, t.data(), t.size() * sizeof(u16), hipMemcpyHostToDevice));
  return p;
}

int pad32(int v) { return round_up(v, 32); }

const HostT& need(H* h, const std::string& key) {
  auto it = h->sd.find(key);
  if (it == h->sd.end()) throw EngineError(DZN_E_MISSING, "missing state_dict key: " + key);
  h->used.insert(key);
  return it->second;
}

void expect_numel(const HostT& t, int64_t n, const std::string& key) {
  if ((int64_t)t.v.size() != n)
    throw EngineError(DZN_E_INVALID, "shape mismatch for " + key + ": got " +
                                         std::to_string(t.v.size()) + " elements, expected " +
                                         std::to_string(n));
}

// W given as [N][K] row-major (already in the contraction's k order); pads to [Np][Kp]
Lin make_lin(H* h, const std::vector<float>& W, const float* bias, int N, int K, int Np, int Kp) {
  std::vector<float> wp((size_t)Np * Kp, 0.f);
  for (int n = 0; n < N; ++n)
    std::copy(W.begin() + (size_t)n * K, W.begin() + (size_t)(n + 1) * K, wp.begin() + (size_t)n * Kp);
  Lin l;
  l.N = Np;
  l.K = Kp;
  l.Nt = N;
  l.Kt = K;
  l.W = upload(h, wp);
  if (h->cfg.precision == DZN_PREC_BF16) l.W16 = upload_bf16(h, wp);
  if (prec_is_split(h->cfg.precision) && Kp % 32 == 0) {
    l.W3 = dalloc<u16>(h, (int64_t)3 * Np * Kp, false);
    if (launch_split_weights(l.W, Np, Kp, Kp, l.W3, nullptr) != DZN_OK)
      throw EngineError(DZN_E_HIP, "split_weights launch failed");
    if (prec_is_h2(h->cfg.precision)) {
      l.W2h = dalloc<u16>(h, (int64_t)2 * Np * Kp, false);
      l.wsc = dalloc<float>(h, Np, false);
      if (launch_split_weights_h2(l.W, Np, Kp, Kp, l.W2h, l.wsc, nullptr) != DZN_OK)
        throw EngineError(DZN_E_HIP, "split_weights_h2 launch failed");
    }
    if (h->cfg.precision == DZN_PREC_F16 && h->mx_pack && h->f16_mx) {
      l.Wmx = dalloc<unsigned char>(h, (int64_t)4 * Np * Kp, false);
      l.wsc_mx = dalloc<float>(h, Np, false);
      if (launch_split_weights_mx(l.W, Np, Kp, Kp, l.Wmx, l.wsc_mx, nullptr) != DZN_OK)
        throw EngineError(DZN_E_HIP, "split_weights_mx launch failed");
    }
    HIPCHK(hipDeviceSynchronize());
  }
  if (bias) {
    std::vector<float> bp(Np, 0.f);
    std::copy(bias, bias + N, bp.begin());
    l.b = upload(h, bp);
  }
  return l;
}

Lin linear_from_sd(H* h, const std::string& prefix, int N, int K, int Np, int Kp, bool bias = true) {
  const HostT& w = need(h, prefix + ".weight");
  expect_numel(w, (int64_t)N * K, prefix + ".weight");
  const float* b = nullptr;
  if (bias) {
    const HostT& bt = need(h, prefix + ".bias");
    expect_numel(bt, N, prefix + ".bias");
    b = bt.v.data();
  }
  return make_lin(h, w.v, b, N, K, Np, Kp);
}

// y = LN(x; gamma, beta) W^T + bias  ==  rstd (x W'^T - mean colsum(W')) + (bias + W beta),  W' = W diag(gamma):
// the contraction reads the RAW rows and its epilogue finishes the norm from (mean, rstd) per row
// (dzn_gemm_desc.ln_stats).  Used where a LayerNorm feeds only linears (pre-norm encoder layers
// W2V/components.py:920-935, feature projection :305-306, Conformer sub-blocks conformer.py:136-214).
Lin make_lin_ln(H* h, const std::vector<float>& W, const float* bias, const std::vector<float>& gamma,
                const std::vector<float>& beta, int N, int K, int Np, int Kp) {
  std::vector<float> wf((size_t)N * K), bf(N), cs(Np, 0.f);
  for (int n = 0; n < N; ++n) {
    double sb = bias ? (double)bias[n] : 0.0, sc = 0.0;
    for (int k = 0; k < K; ++k) {
      const float w = W[(size_t)n * K + k];
      const float wg = w * gamma[k];
      wf[(size_t)n * K + k] = wg;
      sc += (double)wg;
      sb += (double)w * (double)beta[k];
    }
    bf[n] = (float)sb;
    cs[n] = (float)sc;
  }
  Lin l = make_lin(h, wf, bf.data(), N, K, Np, Kp);
  l.csum = upload(h, cs);
  return l;
}

Lin linear_ln_from_sd(H* h, const std::string& prefix, const std::string& ln_prefix, int N, int K, int Np, int Kp) {
  const HostT& w = need(h, prefix + ".weight");
  const HostT& bt = need(h, prefix + ".bias");
  const HostT& g = need(h, ln_prefix + ".weight");
  const HostT& b = need(h, ln_prefix + ".bias");
  expect_numel(w, (int64_t)N * K, prefix + ".weight");
  expect_numel(bt, N, prefix + ".bias");
  expect_numel(g, K, ln_prefix + ".weight");
  expect_numel(b, K, ln_prefix + ".bias");
  return make_lin_ln(h, w.v, bt.v.data(), g.v, b.v, N, K, Np, Kp);
}

LNp ln_from_sd(H* h, const std::string& prefix, int Cn) {
  const HostT& g = need(h, prefix + ".weight");
  const HostT& b = need(h, prefix + ".bias");
  expect_numel(g, Cn, prefix + ".weight");
  expect_numel(b, Cn, prefix + ".bias");
  LNp l;
  l.C = Cn;
  l.g = upload(h, g.v);
  l.b = upload(h, b.v);
  return l;
}

int conv_out(int n, int k, int s) { return n < k ? 0 : (n - k) / s + 1; }

// W2V/components.py:641-666
int relpos_bucket(int rel, int num_buckets, int max_distance) {
  const int nb = num_buckets / 2;
  int ret = rel > 0 ? nb : 0;
  const int a = rel < 0 ? -rel : rel;
  const int max_exact = nb / 2;
  if (a < max_exact) return ret + a;
  const float v = logf((float)a / (float)max_exact) /
                  (float)std::log((double)max_distance / (double)max_exact) *
                  (float)(nb - max_exact);
  long large = max_exact + (long)v;
  if (large > nb - 1) large = nb - 1;
  return ret + (int)large;
}

int64_t lnx_raw_elems(H* h, int64_t B) {
  const dzn_config& c = h->cfg;
  int64_t need = 1;
  if (c.extractor_layer_norm) {
    for (int i = 1; i < c.n_conv; ++i) need = std::max(need, B * h->maxT[i] * h->Cp[i]);
  } else {
    need = B * h->maxT[0] * h->Cp[0];
  }
  return need;
}

// ------------------------------------------------------------------ segmentation: finalize
void finalize_seg(H* h) {
  const dzn_config& c = h->cfg;
  const std::string P = "wavlm_model.";
  h->fold_ln = c.precision != DZN_PREC_BF16 && !getenv("DZN_NO_LN_FOLD");
  h->nconv = c.n_conv;
  h->D = c.embed_dim;
  h->H = c.total_heads;
  h->A = c.attention_in;
  h->Fh = c.ffn_hidden;
  if (c.n_conv < 1 || c.n_conv > DZN_MAX_CONV || c.n_layers < 1 || c.n_layers > DZN_MAX_LAYERS ||
      c.total_heads > DZN_MAX_HEADS || c.embed_dim != c.total_heads * 64 || (c.attention_in % 32) ||
      (c.ffn_hidden % 32) || c.attention_in != c.conf_heads * 64 ||
      c.embed_dim % c.pos_conv_groups || ((c.embed_dim / c.pos_conv_groups) % 8))
    throw EngineError(DZN_E_INVALID, "unsupported architecture (head_dim must be 64, dims % 32)");
  if (c.max_speakers_per_chunk > 8 || c.n_classes > 16)
    throw EngineError(DZN_E_INVALID, "powerset too large");
  for (int i = 0; i < c.n_conv; ++i) {
    h->C[i] = c.conv_ch[i];
    h->Cp[i] = pad32(c.conv_ch[i]);
  }
  // conv0
  {
    const std::string pre = P + "feature_extractor.conv_layers.0";
    const HostT& w = need(h, pre + ".conv.weight");
    expect_numel(w, (int64_t)h->C[0] * c.conv_k[0], pre + ".conv.weight");
    h->conv0_w = upload(h, w.v);
    h->conv_ln[0] = ln_from_sd(h, pre + ".layer_norm", h->C[0]);
    if (c.extractor_layer_norm && c.conv_k[0] <= 10 && !getenv("DZN_CONV0_WAVE_STATS")) {
      // LayerNorm statistics of conv0's frames as a quadratic form of the input samples (frontend.hip)
      const int C0 = h->C[0], k0 = c.conv_k[0];
      std::vector<double> wb(10, 0.0), Q(100, 0.0);
      for (int ch = 0; ch < C0; ++ch)
        for (int i = 0; i < k0; ++i) wb[i] += w.v[(size_t)ch * k0 + i] / C0;
      for (int ch = 0; ch < C0; ++ch)
        for (int i = 0; i < k0; ++i)
          for (int j = 0; j < k0; ++j)
            Q[i * 10 + j] += (w.v[(size_t)ch * k0 + i] - wb[i]) * (w.v[(size_t)ch * k0 + j] - wb[j]) / C0;
      // var = x^T Q x as a plain sum of products cancels when the frame lies near Q's small-eigenvalue directions (trained
      // DC-blocking / band-pass conv0 taps on low-frequency or quiet audio: ADVICE r2).  Q is PSD: factor it once, in
      // double, as Q = sum_e lambda_e v_e v_e^T (cyclic Jacobi) and ship F[e][j] = sqrt(lambda_e) v_e[j]; the kernels
      // evaluate var = sum_e (F[e] . x)^2 — a sum of squares, nothing to cancel.
      std::vector<double> Aj(Q), V(100, 0.0);
      for (int i = 0; i < 10; ++i) V[i * 10 + i] = 1.0;
      for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 10; ++p)
          for (int q = p + 1; q < 10; ++q) off += Aj[p * 10 + q] * Aj[p * 10 + q];
        if (off < 1e-300) break;
        for (int p = 0; p < 10; ++p)
          for (int q = p + 1; q < 10; ++q) {
            const double apq = Aj[p * 10 + q];
            if (apq == 0.0) continue;
            const double theta = (Aj[q * 10 + q] - Aj[p * 10 + p]) / (2.0 * apq);
            const double tt = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
            const double cs = 1.0 / std::sqrt(tt * tt + 1.0), sn = tt * cs;
            for (int r = 0; r < 10; ++r) {        // A <- A J (columns p, q)
              const double arp = Aj[r * 10 + p], arq = Aj[r * 10 + q];
              Aj[r * 10 + p] = cs * arp - sn * arq;
              Aj[r * 10 + q] = sn * arp + cs * arq;
            }
            for (int r = 0; r < 10; ++r) {        // A <- J^T A (rows p, q)
              const double apr = Aj[p * 10 + r], aqr = Aj[q * 10 + r];
              Aj[p * 10 + r] = cs * apr - sn * aqr;
              Aj[q * 10 + r] = sn * apr + cs * aqr;
            }
            for (int r = 0; r < 10; ++r) {        // V <- V J
              const double vrp = V[r * 10 + p], vrq = V[r * 10 + q];
              V[r * 10 + p] = cs * vrp - sn * vrq;
              V[r * 10 + q] = sn * vrp + cs * vrq;
            }
          }
      }
      std::vector<float> lq(110);
      for (int i = 0; i < 10; ++i) lq[i] = (float)wb[i];
      for (int e = 0; e < 10; ++e) {
        const double lam = std::max(Aj[e * 10 + e], 0.0);
        for (int j = 0; j < 10; ++j) lq[10 + e * 10 + j] = (float)(std::sqrt(lam) * V[j * 10 + e]);
      }
      if (k0 == 10 && C0 % 2 == 0) {
        // behind the coefficients (from float 112 on): conv0's taps + LayerNorm affine, two channels per 24-float record
        // [w c[10], w c+1[10], gamma c, gamma c+1, beta c, beta c+1] — what a producer wavefront of frontend_fused.hip's
        // conv01_ws_kernel needs per channel pair, as ONE contiguous scalar load
        const std::vector<float>& g0 = need(h, pre + ".layer_norm.weight").v;
        const std::vector<float>& b0 = need(h, pre + ".layer_norm.bias").v;
        lq.resize(112 + (size_t)(C0 / 2) * 24, 0.f);
        for (int cp = 0; cp < C0 / 2; ++cp) {
          float* r = lq.data() + 112 + (size_t)cp * 24;
          for (int t = 0; t < 10; ++t) { r[t] = w.v[(size_t)(2 * cp) * 10 + t]; r[10 + t] = w.v[(size_t)(2 * cp + 1) * 10 + t]; }
          r[20] = g0[2 * cp]; r[21] = g0[2 * cp + 1]; r[22] = b0[2 * cp]; r[23] = b0[2 * cp + 1];
        }
      }
      h->conv0_lnq = upload(h, lq);
    }
    if (c.extractor_layer_norm) {   // |LN(x)_c| <= sqrt(C - 1), |GELU(t)| <= |t|
      float mg = 0.f, mb = 0.f;
      for (float v : need(h, pre + ".layer_norm.weight").v) mg = std::max(mg, std::fabs(v));
      for (float v : need(h, pre + ".layer_norm.bias").v) mb = std::max(mb, std::fabs(v));
      h->conv0_bound = std::sqrt((float)h->C[0]) * mg + mb;
    }
  }
  for (int i = 1; i < c.n_conv; ++i) {
    const std::string pre = P + "feature_extractor.conv_layers." + std::to_string(i);
    const HostT& w = need(h, pre + ".conv.weight");
    const int ci = h->C[i - 1], co = h->C[i], k = c.conv_k[i], cip = h->Cp[i - 1];
    expect_numel(w, (int64_t)co * ci * k, pre + ".conv.weight");
    std::vector<float> wp((size_t)co * k * cip, 0.f);  // [co][j*cip + ci] = w[co][ci][j]
    for (int o = 0; o < co; ++o)
      for (int ii = 0; ii < ci; ++ii)
        for (int j = 0; j < k; ++j)
          wp[((size_t)o * k + j) * cip + ii] = w.v[((size_t)o * ci + ii) * k + j];
    h->conv[i] = make_lin(h, wp, nullptr, co, k * cip, h->Cp[i], k * cip);
    h->conv[i].Kt = k * ci;
    if (c.extractor_layer_norm) h->conv_ln[i] = ln_from_sd(h, pre + ".layer_norm", co);
    if (i == 1 && prec_is_h2(c.precision) && c.extractor_layer_norm && h->conv0_lnq && c.conv_k[0] == 10 &&
        c.conv_s[0] == 5 && k == 3 && c.conv_s[1] == 2 && h->C[0] % 64 == 0 && h->Cp[0] == h->C[0] &&
        h->Cp[1] == 160 && !getenv("DZN_NO_CONV01_FUSION")) {
      // two copies of the planes: [row][K/32][2][32] (phase-alternating kernel), then fragment-major (producer / consumer kernel)
      h->conv1_W2n = dalloc<u16>(h, (int64_t)4 * h->Cp[1] * k * cip, false);
      h->conv1_wscn = dalloc<float>(h, h->Cp[1], false);
      if (launch_split_weights_h2_natural(h->conv[1].W, h->Cp[1], k * cip, h->conv1_W2n, h->conv1_wscn, nullptr) != DZN_OK)
        throw EngineError(DZN_E_HIP, "split_weights_h2_natural launch failed");
      if (launch_fragment_major(h->conv1_W2n, h->Cp[1], k * cip, h->conv1_W2n + (int64_t)2 * h->Cp[1] * k * cip, nullptr) != DZN_OK)
        throw EngineError(DZN_E_HIP, "fragment_major launch failed");
      HIPCHK(hipDeviceSynchronize());
    }
  }
  const int last = c.n_conv - 1;
  {
    const HostT& dw = need(h, P + "feature_extractor.dummy_weight");
    expect_numel(dw, h->C[last], "dummy_weight");
    h->dummy_w = upload(h, dw.v);
  }
  const int D = h->D;
  h->fp_ln = ln_from_sd(h, P + "encoder.feature_projection.layer_norm", h->C[last]);
  if (h->fold_ln)
    h->fp = linear_ln_from_sd(h, P + "encoder.feature_projection.projection",
                              P + "encoder.feature_projection.layer_norm", D, h->C[last], D, h->Cp[last]);
  else
    h->fp = linear_from_sd(h, P + "encoder.feature_projection.projection", D, h->C[last], D,
                           h->Cp[last]);
  // positional conv: fold weight norm  W = g * v / ||v||_{dims 0,1}   (components.py:344)
  {
    const std::string pc = P + "encoder.transformer.pos_conv_embed.conv";
    const HostT& g = need(h, pc + ".parametrizations.weight.original0");
    const HostT& v = need(h, pc + ".parametrizations.weight.original1");
    const HostT& b = need(h, pc + ".bias");
    const int K = c.pos_conv_kernel, G = c.pos_conv_groups, cg = D / G;
    expect_numel(g, K, pc + ".original0");
    expect_numel(v, (int64_t)D * cg * K, pc + ".original1");
    expect_numel(b, D, pc + ".bias");
    std::vector<double> nrm(K, 0.0);
    for (int o = 0; o < D; ++o)
      for (int ii = 0; ii < cg; ++ii)
        for (int j = 0; j < K; ++j) {
          const double t = v.v[((size_t)o * cg + ii) * K + j];
          nrm[j] += t * t;
        }
    // split modes: groups whose width is not a multiple of 32 (base: 768 / 16 = 48) are zero-padded to the next one
    // (64) in the weights AND in the pre-split padded copy of x, so the contraction keeps whole 32-k tiles per tap and
    // runs on the fp16 / bf16 planes instead of the fp32 MFMA fallback (r2: 15 % of BASELINE configs[1])
    const int cgp = prec_is_split(c.precision) ? round_up(cg, 32) : cg;
    h->pos_cgp = cgp;
    std::vector<float> wp((size_t)D * K * cgp, 0.f);
    for (int o = 0; o < D; ++o)
      for (int ii = 0; ii < cg; ++ii)
        for (int j = 0; j < K; ++j) {
          const float nj = (float)std::sqrt(nrm[j]);
          wp[((size_t)o * K + j) * cgp + ii] = g.v[j] * v.v[((size_t)o * cg + ii) * K + j] / nj;
        }
    h->posconv = make_lin(h, wp, b.v.data(), D, K * cgp, D, K * cgp);
    h->posconv.Kt = K * cg;
  }
  if (!c.layer_norm_first) h->enc_ln = ln_from_sd(h, P + "encoder.transformer.layer_norm", D);
  h->layers.resize(c.n_layers);
  int maxQ = 64, maxF = 32;
  for (int i = 0; i < c.n_layers; ++i) {
    EncLayer& L = h->layers[i];
    const std::string lp = P + "encoder.transformer.layers." + std::to_string(i);
    L.attn = c.use_attention[i] && c.n_heads[i] > 0;
    L.ffn = c.use_ffn[i] != 0;
    L.ln1 = ln_from_sd(h, lp + ".layer_norm", D);
    L.ln2 = ln_from_sd(h, lp + ".final_layer_norm", D);
    if (L.attn) {
      L.h = c.n_heads[i];
      const int hd = L.h * 64;
      maxQ = std::max(maxQ, hd);
      std::vector<float> wq((size_t)3 * hd * D), bq((size_t)3 * hd);
      const char* names[3] = {".attention.q_proj", ".attention.k_proj", ".attention.v_proj"};
      for (int t = 0; t < 3; ++t) {
        const HostT& w = need(h, lp + names[t] + ".weight");
        const HostT& b = need(h, lp + names[t] + ".bias");
        expect_numel(w, (int64_t)hd * D, lp + names[t] + ".weight");
        expect_numel(b, hd, lp + names[t] + ".bias");
        std::copy(w.v.begin(), w.v.end(), wq.begin() + (size_t)t * hd * D);
        std::copy(b.v.begin(), b.v.end(), bq.begin() + (size_t)t * hd);
      }
      if (h->fold_ln && c.layer_norm_first)
        L.qkv = make_lin_ln(h, wq, bq.data(), need(h, lp + ".layer_norm.weight").v,
                            need(h, lp + ".layer_norm.bias").v, 3 * hd, D, 3 * hd, D);
      else
        L.qkv = make_lin(h, wq, bq.data(), 3 * hd, D, 3 * hd, D);
      L.out = linear_from_sd(h, lp + ".attention.out_proj", D, hd, D, hd);
      const HostT& wg = need(h, lp + ".attention.gru_rel_pos_linear.weight");
      const HostT& bg = need(h, lp + ".attention.gru_rel_pos_linear.bias");
      const HostT& cs = need(h, lp + ".attention.gru_rel_pos_const");
      expect_numel(wg, 8 * 64, "gru_rel_pos_linear.weight");
      expect_numel(bg, 8, "gru_rel_pos_linear.bias");
      expect_numel(cs, h->H, "gru_rel_pos_const");
      L.Wg = upload(h, wg.v);
      L.bg = upload(h, bg.v);
      L.cst = upload(h, cs.v);
      std::vector<int32_t> hi(c.head_idx[i], c.head_idx[i] + L.h);
      for (int v : hi)
        if (v < 0 || v >= h->H) throw EngineError(DZN_E_INVALID, "head index out of range");
      L.head_idx = dalloc<int32_t>(h, L.h, false);
      HIPCHK(hipMemcpy(L.head_idx, hi.data(), L.h * sizeof(int32_t), hipMemcpyHostToDevice));
      if (i == 0) {
        const HostT& e = need(h, lp + ".attention.rel_attn_embed.weight");
        expect_numel(e, (int64_t)c.num_buckets * h->H, "rel_attn_embed.weight");
        h->rel_embed = e.v;
      }
    }
    if (L.ffn) {
      L.F = c.ffn_dim[i];
      L.Fp = pad32(L.F);
      maxF = std::max(maxF, L.Fp);
      if (h->fold_ln && c.layer_norm_first)
        L.f1 = linear_ln_from_sd(h, lp + ".feed_forward.intermediate_dense", lp + ".final_layer_norm", L.F, D,
                                 L.Fp, D);
      else
        L.f1 = linear_from_sd(h, lp + ".feed_forward.intermediate_dense", L.F, D, L.Fp, D);
      L.f2 = linear_from_sd(h, lp + ".feed_forward.output_dense", D, L.F, D, L.Fp);
    }
  }
  if (h->rel_embed.empty())
    throw EngineError(DZN_E_INVALID, "layer 0 must carry attention (rel_attn_embed lives there)");
  {
    const HostT& w = need(h, "weight_sum.weight");
    expect_numel(w, c.n_layers + 1, "weight_sum.weight");
    h->wsum_w = w.v;
  }
  const int A = h->A, Fh = h->Fh;
  h->proj = linear_from_sd(h, "proj", A, D, A, D);
  h->lnorm = ln_from_sd(h, "lnorm", A);
  h->conf.resize(c.conf_layers);
  for (int i = 0; i < c.conf_layers; ++i) {
    ConfLayer& L = h->conf[i];
    const std::string cp = "conformer.conformer_layer." + std::to_string(i);
    L.ffn1_ln = ln_from_sd(h, cp + ".ffn1.ln_norm", A);
    L.ffn1_w1 = h->fold_ln ? linear_ln_from_sd(h, cp + ".ffn1.w_1", cp + ".ffn1.ln_norm", Fh, A, Fh, A)
                           : linear_from_sd(h, cp + ".ffn1.w_1", Fh, A, Fh, A);
    L.ffn1_w2 = linear_from_sd(h, cp + ".ffn1.w_2", A, Fh, A, Fh);
    L.ffn2_ln = ln_from_sd(h, cp + ".ffn2.ln_norm", A);
    L.ffn2_w1 = h->fold_ln ? linear_ln_from_sd(h, cp + ".ffn2.w_1", cp + ".ffn2.ln_norm", Fh, A, Fh, A)
                           : linear_from_sd(h, cp + ".ffn2.w_1", Fh, A, Fh, A);
    L.ffn2_w2 = linear_from_sd(h, cp + ".ffn2.w_2", A, Fh, A, Fh);
    L.mha_ln = ln_from_sd(h, cp + ".mha.ln_norm", A);
    {
      std::vector<float> wq((size_t)3 * A * A), bq((size_t)3 * A);
      const char* names[3] = {".mha.mha.linearQ", ".mha.mha.linearK", ".mha.mha.linearV"};
      for (int t = 0; t < 3; ++t) {
        const HostT& w = need(h, cp + names[t] + ".weight");
        const HostT& b = need(h, cp + names[t] + ".bias");
        expect_numel(w, (int64_t)A * A, cp + names[t] + ".weight");
        expect_numel(b, A, cp + names[t] + ".bias");
        std::copy(w.v.begin(), w.v.end(), wq.begin() + (size_t)t * A * A);
        std::copy(b.v.begin(), b.v.end(), bq.begin() + (size_t)t * A);
      }
      if (h->fold_ln)
        L.qkv = make_lin_ln(h, wq, bq.data(), need(h, cp + ".mha.ln_norm.weight").v,
                            need(h, cp + ".mha.ln_norm.bias").v, 3 * A, A, 3 * A, A);
      else
        L.qkv = make_lin(h, wq, bq.data(), 3 * A, A, 3 * A, A);
    }
    L.o = linear_from_sd(h, cp + ".mha.mha.linearO", A, A, A, A);
    L.conv_ln = ln_from_sd(h, cp + ".conv.ln_norm", A);
    L.pw1 = h->fold_ln ? linear_ln_from_sd(h, cp + ".conv.pointwise_conv1", cp + ".conv.ln_norm", 2 * A, A, 2 * A, A)
                       : linear_from_sd(h, cp + ".conv.pointwise_conv1", 2 * A, A, 2 * A, A);
    L.pw2 = linear_from_sd(h, cp + ".conv.pointwise_conv2", A, A, A, A);
    {
      // fold eval BatchNorm1d into the depthwise taps (conformer.py:205)
      const int ks = c.conf_kernel;
      const HostT& w = need(h, cp + ".conv.depthwise_conv.weight");
      const HostT& b = need(h, cp + ".conv.depthwise_conv.bias");
      const HostT& g = need(h, cp + ".conv.bn_norm.weight");
      const HostT& be = need(h, cp + ".conv.bn_norm.bias");
      const HostT& rm = need(h, cp + ".conv.bn_norm.running_mean");
      const HostT& rv = need(h, cp + ".conv.bn_norm.running_var");
      expect_numel(w, (int64_t)A * ks, cp + ".conv.depthwise_conv.weight");
      std::vector<float> wf((size_t)A * ks), bf(A);
      for (int ch = 0; ch < A; ++ch) {
        const float s = g.v[ch] / std::sqrt(rv.v[ch] + 1e-5f);
        for (int j = 0; j < ks; ++j) wf[(size_t)ch * ks + j] = w.v[(size_t)ch * ks + j] * s;
        bf[ch] = (b.v[ch] - rm.v[ch]) * s + be.v[ch];
      }
      L.dw = upload(h, wf);
      L.dwb = upload(h, bf);
    }
    L.out_ln = ln_from_sd(h, cp + ".ln_norm", A);
  }
  {
    const HostT& w = need(h, "classifier.weight");
    const HostT& b = need(h, "classifier.bias");
    expect_numel(w, (int64_t)c.n_classes * A, "classifier.weight");
    expect_numel(b, c.n_classes, "classifier.bias");
    h->cls_w = upload(h, w.v);
    h->cls_b = upload(h, b.v);
    // PA/utils/powerset.py:68-97: classes ordered by set size, then lexicographic combinations
    const int S = c.max_speakers_per_chunk;
    std::vector<uint8_t> map;
    int rows = 0;
    for (int size = 0; size <= c.max_speakers_per_frame; ++size) {
      std::vector<int> idx(size);
      for (int i = 0; i < size; ++i) idx[i] = i;
      while (true) {
        std::vector<uint8_t> r(S, 0);
        for (int v : idx) r[v] = 1;
        map.insert(map.end(), r.begin(), r.end());
        ++rows;
        int i = size - 1;
        while (i >= 0 && idx[i] == S - size + i) --i;
        if (i < 0) break;
        ++idx[i];
        for (int j = i + 1; j < size; ++j) idx[j] = idx[j - 1] + 1;
      }
    }
    if (rows != c.n_classes) throw EngineError(DZN_E_INVALID, "n_classes != powerset size");
    h->mapping = dalloc<uint8_t>(h, (int64_t)map.size(), false);
    HIPCHK(hipMemcpy(h->mapping, map.data(), map.size(), hipMemcpyHostToDevice));
  }

  // ---- workspace for (max_batch, max_samples) ----
  const int64_t B = c.max_batch;
  int n = c.max_samples;
  int64_t maxbuf = 0;
  for (int i = 0; i < c.n_conv; ++i) {
    n = conv_out(n, c.conv_k[i], c.conv_s[i]);
    h->maxT[i] = n;
    maxbuf = std::max(maxbuf, (int64_t)B * n * h->Cp[i]);
  }
  h->maxL = n;
  if (n < 1) throw EngineError(DZN_E_INVALID, "max_samples too short");
  const int64_t ML = B * n;
  h->stats = dalloc<float>(h, B * 2);
  if (!c.extractor_layer_norm) h->gn_stats = dalloc<float>(h, gn_stats_floats((int)B, h->maxT[0], h->C[0], h->Cp[0]));
  h->bufA = dalloc<float>(h, maxbuf);
  {
    // bufB holds every odd conv output (and the feature-projection LN output)
    int64_t need_b = B * n * h->Cp[c.n_conv - 1];
    for (int i = 1; i < c.n_conv; i += 2) need_b = std::max(need_b, B * h->maxT[i] * h->Cp[i]);
    h->bufB = dalloc<float>(h, need_b);
  }
  if (c.precision == DZN_PREC_BF16) {
    // fp32 scratch for raw conv outputs in front of their LayerNorm / GroupNorm (bf16 mode only)
    int64_t need_r = lnx_raw_elems(h, B);
    h->craw = dalloc<float>(h, need_r);
  }
  h->x = dalloc<float>(h, ML * D);
  h->xpad = dalloc<float>(h, B * (n + c.pos_conv_kernel) * D);
  if (prec_is_split(c.precision)) {
    h->xpad3_plane = B * (n + c.pos_conv_kernel) * (int64_t)(c.pos_conv_groups * round_up(D / c.pos_conv_groups, 32));
    h->xpad3 = dalloc<u16>(h, 3 * h->xpad3_plane);
  }
  h->y = dalloc<float>(h, ML * D);
  h->ws = dalloc<float>(h, ML * D);
  // the per-layer buffers of the deferred layer-weighted sum are OPTIONAL: they are taken last, by alloc_optional_workspace()
  h->xl_elems = (c.layer_norm_first && h->fold_ln && c.n_layers > 0 && c.n_layers < WS_SUM_MAX && D % 4 == 0 &&
                 !getenv("DZN_NO_WS_DEFER")) ? (int64_t)c.n_layers * ML * D : 0;
  h->qkv = dalloc<float>(h, ML * 3 * maxQ);
  if (c.precision == DZN_PREC_F32_H2 && maxQ > 0 && !getenv("DZN_NO_ATT_PLANES")) {
    // two fp16 planes of [rows + 64][2 * maxQ] (the bytes of the fp32 K / V they replace) and the slot scales; zero-filled once:
    // the 64 rows past a batch are only ever read for keys that are masked, but they must be finite
    h->kvp_rows = ML + 64;
    h->kvp_ldmax = 2 * maxQ;
    h->kvp = dalloc<uint16_t>(h, 2 * h->kvp_rows * h->kvp_ldmax);
    h->kvs = dalloc<float>(h, h->kvp_rows * (h->kvp_ldmax / 64));
  }
  h->ao = dalloc<float>(h, ML * maxQ);
  h->gate = dalloc<float>(h, ML * h->H);
  h->mid = dalloc<float>(h, ML * maxF);
  h->hz = dalloc<float>(h, ML * A);
  h->ht = dalloc<float>(h, ML * A);
  h->hmid = dalloc<float>(h, ML * std::max(Fh, 3 * A));
  h->hv = dalloc<float>(h, ML * A);
  h->rstat = dalloc<float>(h, ML * 2);
  h->spart = dalloc<float>(h, ML * 32 * 2);
  h->amax = dalloc<float>(h, (int64_t)dzn_handle::AM_COUNT * c.max_batch);
}

// Optional workspace, taken AFTER everything a forward needs (segmentation + embedding) and only against a reserve, so that
// several handles on one device (DiariZenPipeline num_streams, bench config1) cannot starve what is allocated later — torch's
// waveform / chunk copies, VBx, RCCL (the r4 gate was `need < free / 2` per handle at a point where the embedding
// workspace did not exist yet).  A failed hipMalloc leaves the read-modify-write path in place instead of failing the handle.
void alloc_optional_workspace(H* h) {
  if (h->xl_elems <= 0) return;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return;
  const int64_t need = h->xl_elems * (int64_t)sizeof(float);
  const int64_t reserve = std::max<int64_t>((int64_t)32 << 30, (int64_t)(total_b / 4));   // stays free after the allocation
  if (need + reserve > (int64_t)free_b) return;
  void* p = nullptr;
  if (hipMalloc(&p, (size_t)need) != hipSuccess) {
    (void)hipGetLastError();
    return;
  }
  h->allocs.push_back(p);
  h->bytes += need;
  h->xl = reinterpret_cast<float*>(p);
}

// ------------------------------------------------------------------ embedding: finalize
void fold_bn(H* h, const std::string& bn, int Cn, std::vector<float>& scale, std::vector<float>& shift) {
  const HostT& g = need(h, bn + ".weight");
  const HostT& b = need(h, bn + ".bias");
  const HostT& rm = need(h, bn + ".running_mean");
  const HostT& rv = need(h, bn + ".running_var");
  expect_numel(g, Cn, bn + ".weight");
  scale.resize(Cn);
  shift.resize(Cn);
  for (int i = 0; i < Cn; ++i) {
    scale[i] = g.v[i] / std::sqrt(rv.v[i] + 1e-5f);
    shift[i] = b.v[i] - rm.v[i] * scale[i];
  }
}

ResConv make_resconv(H* h, const std::string& conv, const std::string& bn, int cin, int cout, int ksz,
                     int stride) {
  const HostT& w = need(h, conv + ".weight");
  expect_numel(w, (int64_t)cout * cin * ksz * ksz, conv + ".weight");
  std::vector<float> sc, sh;
  fold_bn(h, bn, cout, sc, sh);
  const int K = ksz * ksz * cin;
  std::vector<float> wp((size_t)cout * K);
  for (int o = 0; o < cout; ++o)
    for (int ci = 0; ci < cin; ++ci)
      for (int dh = 0; dh < ksz; ++dh)
        for (int dw = 0; dw < ksz; ++dw)
          wp[(size_t)o * K + (dh * ksz + dw) * cin + ci] =
              w.v[(((size_t)o * cin + ci) * ksz + dh) * ksz + dw] * sc[o];
  ResConv r;
  r.cin = cin;
  r.cout = cout;
  r.ksz = ksz;
  r.stride = stride;
  r.l = make_lin(h, wp, sh.data(), cout, K, cout, K);
  for (int o = 0; o < cout; ++o) {
    double l1 = 0.0;
    for (int k = 0; k < K; ++k) l1 += std::fabs((double)wp[(size_t)o * K + k]);
    r.l1max = std::max(r.l1max, (float)(l1 * (1.0 + 1e-6)));
    r.bmax = std::max(r.bmax, std::fabs(sh[o]));
  }
  return r;
}

void finalize_emb(H* h) {
  const dzn_config& c = h->cfg;
  const std::string E = "embedding.resnet.";
  const int NB = c.num_mel_bins;
  if (NB != 80 || c.embed_out_dim <= 0) throw EngineError(DZN_E_INVALID, "embedding config");
  const int flen = 400, NFFT = 512, Kp = 416;
  {
    // torch.hamming_window(400, periodic=False): 0.54 - 0.46 cos(2 pi n / (N-1))
    std::vector<float> wdw(flen);
    for (int i = 0; i < flen; ++i)
      wdw[i] = (float)(0.54 - 0.46 * std::cos(2.0 * M_PI * (double)i / (double)(flen - 1)));
    h->hamming = upload(h, wdw);
    // real DFT as a contraction: rows 0..255 cos, 256..511 sin, bins 0..255 (bin 256 unused)
    std::vector<float> dft((size_t)512 * Kp, 0.f);
    for (int f = 0; f < 256; ++f)
      for (int t = 0; t < flen; ++t) {
        const double ang = 2.0 * M_PI * (double)((int64_t)f * t % NFFT) / (double)NFFT;
        dft[(size_t)f * Kp + t] = (float)std::cos(ang);
        dft[(size_t)(256 + f) * Kp + t] = (float)(-std::sin(ang));
      }
    h->dft = make_lin(h, dft, nullptr, 512, Kp, 512, Kp);
    // Kaldi mel banks (torchaudio.compliance.kaldi.get_mel_banks, float32 arithmetic):
    // low 20 Hz, high = Nyquist, 80 triangles in mel(f) = 1127 ln(1 + f/700), bins 0..255
    const float mel_low = 1127.0f * logf(1.0f + 20.0f / 700.0f);
    const float mel_high = 1127.0f * logf(1.0f + 8000.0f / 700.0f);
    const float delta = (mel_high - mel_low) / (float)(NB + 1);
    std::vector<float> mw((size_t)NB * 256, 0.f);
    for (int b = 0; b < NB; ++b) {
      const float left = mel_low + (float)b * delta;
      const float center = mel_low + ((float)b + 1.0f) * delta;
      const float right = mel_low + ((float)b + 2.0f) * delta;
      for (int f = 0; f < 256; ++f) {
        const float mel = 1127.0f * logf(1.0f + (31.25f * (float)f) / 700.0f);
        const float up = (mel - left) / (center - left);
        const float down = (right - mel) / (right - center);
        mw[(size_t)b * 256 + f] = std::max(0.0f, std::min(up, down));
      }
    }
    h->mel = make_lin(h, mw, nullptr, NB, 256, NB, 256);
  }
  const int m = 32;
  {
    const HostT& w = need(h, E + "conv1.weight");
    expect_numel(w, (int64_t)m * 9, E + "conv1.weight");
    std::vector<float> sc, sh;
    fold_bn(h, E + "bn1", m, sc, sh);
    std::vector<float> wf(w.v);
    for (int o = 0; o < m; ++o)
      for (int k = 0; k < 9; ++k) wf[(size_t)o * 9 + k] *= sc[o];
    h->stem_w = upload(h, wf);
    h->stem_b = upload(h, sh);
  }
  const int nblocks[4] = {3, 4, 6, 3};
  h->stages.resize(4);
  int cin = m;
  for (int s = 0; s < 4; ++s) {
    const int cout = m << s;
    h->sC[s] = cout;
    h->stages[s].resize(nblocks[s]);
    for (int j = 0; j < nblocks[s]; ++j) {
      const std::string bp = E + "layer" + std::to_string(s + 1) + "." + std::to_string(j);
      ResBlock& rb = h->stages[s][j];
      const int stride = (j == 0 && s > 0) ? 2 : 1;
      rb.c1 = make_resconv(h, bp + ".conv1", bp + ".bn1", cin, cout, 3, stride);
      rb.c2 = make_resconv(h, bp + ".conv2", bp + ".bn2", cout, cout, 3, 1);
      rb.has_sc = (stride != 1 || cin != cout);
      if (rb.has_sc) rb.sc = make_resconv(h, bp + ".shortcut.0", bp + ".shortcut.1", cin, cout, 1, stride);
      cin = cout;
    }
  }
  const int feat = (m << 3) * (NB / 8) * 2;  // 256 * 10 * 2 = 5120
  h->seg1 = linear_from_sd(h, E + "seg_1", c.embed_out_dim, feat, c.embed_out_dim, feat);

  // workspace
  const int64_t B = c.max_batch;
  const int Tmax = c.max_samples < flen ? 0 : 1 + (c.max_samples - flen) / 160;
  if (Tmax < 8) throw EngineError(DZN_E_INVALID, "max_samples too short for the embedding model");
  h->maxTf = Tmax;
  h->frames = dalloc<float>(h, B * Tmax * Kp);
  h->spec = dalloc<float>(h, B * Tmax * 512);
  h->pw = dalloc<float>(h, B * Tmax * 256);
  h->fb = dalloc<float>(h, B * Tmax * NB);
  int Hs = NB, Ws = Tmax;
  for (int s = 0; s < 4; ++s) {
    if (s > 0) {
      Hs = (Hs - 1) / 2 + 1;
      Ws = (Ws - 1) / 2 + 1;
    }
    h->sbuf_elems[s] = (int64_t)(Hs + 2) * (Ws + 2) * h->sC[s];
    for (int k = 0; k < 3; ++k) h->sbuf[s][k] = dalloc<float>(h, B * h->sbuf_elems[s]);
  }
  h->pool = dalloc<float>(h, B * 8 * feat);
  h->win_flag = dalloc<int>(h, B);
  h->win_idx = dalloc<int>(h, B);
  h->win_cnt = dalloc<int>(h, 1);
  h->emb_totals = dalloc<long long>(h, 2);
}

// bake the geometry of T fbank frames into the row-offset tables and re-zero the image borders.
// Tables are built once per T and stay on the device; a later switch back to that T is two async
// operations on the caller's stream (no host synchronisation, no upload).
void emb_set_geometry(H* h, int T, hipStream_t st) {
  if (T == h->emb_T) return;
  const int64_t B = h->cfg.max_batch;
  int Hs = h->cfg.num_mel_bins, Ws = T;
  int Hp = 0, Wp = 0;
  auto it = h->geoms.find(T);
  const bool fresh = it == h->geoms.end();
  if (fresh && h->geoms.size() >= 64) throw EngineError(DZN_E_INVALID, "too many distinct window lengths on one handle");
  H::EmbGeom& gm = h->geoms[T];
  for (int s = 0; s < 4; ++s) {
    if (s > 0) {
      Hp = Hs;
      Wp = Ws;
      Hs = (Hs - 1) / 2 + 1;
      Ws = (Ws - 1) / 2 + 1;
    }
    h->sH[s] = Hs;
    h->sW[s] = Ws;
    const int Cc = h->sC[s];
    h->simg[s] = (int64_t)(Hs + 2) * (Ws + 2) * Cc;
    if (fresh) {
      std::vector<int32_t> t1((size_t)Hs * Ws), t2((size_t)Hs * Ws, 0);
      for (int y = 0; y < Hs; ++y)
        for (int x = 0; x < Ws; ++x) {
          t1[(size_t)y * Ws + x] = (y * (Ws + 2) + x) * Cc;  // top-left of the 3x3 patch
          if (s > 0) t2[(size_t)y * Ws + x] = ((2 * y) * (Wp + 2) + 2 * x) * h->sC[s - 1];
        }
      gm.t1[s] = dalloc<int32_t>(h, (int64_t)Hs * Ws, false);
      gm.t2[s] = dalloc<int32_t>(h, (int64_t)Hs * Ws, false);
      HIPCHK(hipMemcpy(gm.t1[s], t1.data(), t1.size() * 4, hipMemcpyHostToDevice));  // synchronous, first use only
      HIPCHK(hipMemcpy(gm.t2[s], t2.data(), t2.size() * 4, hipMemcpyHostToDevice));
    }
    h->tab1[s] = gm.t1[s];
    h->tab2[s] = gm.t2[s];
    for (int k = 0; k < 3; ++k)
      HIPCHK(hipMemsetAsync(h->sbuf[s][k], 0, B * h->sbuf_elems[s] * sizeof(float), st));
  }
  h->emb_T = T;
}

// ------------------------------------------------------------------ forward helpers
void chk(int rc, const char* what) {
  if (rc != DZN_OK) throw EngineError(rc, std::string("kernel launch failed: ") + what);
}

// byte-level offset into an activation buffer whose element type depends on the engine mode
inline float* eoff(float* p, int64_t elems, bool bf16) {
  return reinterpret_cast<float*>(reinterpret_cast<char*>(p) + elems * (bf16 ? 2 : 4));
}
inline const float* eoff(const float* p, int64_t elems, bool bf16) {
  return reinterpret_cast<const float*>(reinterpret_cast<const char*>(p) + elems * (bf16 ? 2 : 4));
}

void tap(H* h, const char* name, const float* p, int64_t rows, int cols, int64_t ld, hipStream_t st,
         bool bf16 = false) {
  if (!h->debug) return;
  HIPCHK(hipStreamSynchronize(st));
  std::vector<float>& v = h->taps[name];
  v.resize((size_t)rows * cols);
  if (!bf16) {
    HIPCHK(hipMemcpy2D(v.data(), (size_t)cols * 4, p, (size_t)ld * 4, (size_t)cols * 4, (size_t)rows,
                       hipMemcpyDeviceToHost));
    return;
  }
  std::vector<u16> t((size_t)rows * cols);
  HIPCHK(hipMemcpy2D(t.data(), (size_t)cols * 2, p, (size_t)ld * 2, (size_t)cols * 2, (size_t)rows,
                     hipMemcpyDeviceToHost));
  for (size_t i = 0; i < t.size(); ++i) {
    const uint32_t u = (uint32_t)t[i] << 16;
    memcpy(&v[i], &u, 4);
  }
}

dzn_gemm_desc gd(H* h, const float* A, const Lin& l, float* C, int64_t M, int64_t lda, int64_t ldc) {
  dzn_gemm_desc d{};
  d.A = A;
  d.W = l.W;
  d.W16 = l.W16;
  d.W3 = l.W3;
  d.W2h = l.W2h;
  d.col_scale = l.wsc;
  d.Wmx = l.Wmx;
  d.col_scale_mx = l.wsc_mx;
  d.amax_count = h->cfg.max_batch;     // every tracker array has one entry per window of the largest batch (checked builds)
  d.C = C;
  d.bias = l.b;
  d.M = (int)M;
  d.N = l.N;
  d.K = l.K;
  d.lda = lda;
  d.kc = l.K;
  d.ldk = 0;
  d.ldw = l.K;
  d.ldc = ldc;
  d.alpha = 1.f;
  d.nz = 1;
  d.zdiv = 1;
  d.precision = h->cfg.precision;
  d.alg_flops = 2.0 * (double)M * l.Nt * l.Kt;  // per z
  return d;
}

// LayerNorm with typed input / output (fp32, or bf16 in the bf16 engine mode)
void ln_t(const float* x, bool x16, int64_t ldx, float* y, bool y16, int64_t ldy, const LNp& p, int64_t rows,
          int Cpad, int gelu, hipStream_t st, const float* post = nullptr, float* amax = nullptr,
          int64_t amax_unit = 0) {
  chk(launch_layernorm_t(x, x16, ldx, y, y16, ldy, p.g, p.b, post, rows, p.C, Cpad, 1e-5f, gelu, st, amax, amax_unit),
      "layernorm");
}

// positional conv rows: m = b*L + t reads xpad[b][t .. t+Kc) — one table for all batch sizes
void ensure_pos_rowoff(H* h, int L, int Lp, int rowD, hipStream_t st) {
  if (L == h->pos_L && rowD == h->pos_rowD) return;
  h->pos_rowD = rowD;
  HIPCHK(hipStreamSynchronize(st));
  const int Bm = h->cfg.max_batch;
  std::vector<int32_t> t((size_t)Bm * L);
  for (int b = 0; b < Bm; ++b)
    for (int i = 0; i < L; ++i) t[(size_t)b * L + i] = (int32_t)(((int64_t)b * Lp + i) * h->pos_rowD);
  if (!h->pos_rowoff) h->pos_rowoff = dalloc<int32_t>(h, (int64_t)Bm * h->maxL, false);
  HIPCHK(hipMemcpy(h->pos_rowoff, t.data(), t.size() * 4, hipMemcpyHostToDevice));
  h->pos_L = L;
}

void ensure_table(H* h, int L, hipStream_t st) {
  if (L == h->table_L) return;
  HIPCHK(hipStreamSynchronize(st));
  const int Hn = h->H, W = 2 * L - 1;
  std::vector<float> t((size_t)Hn * W);
  for (int r = -(L - 1); r <= L - 1; ++r) {
    const int bk = relpos_bucket(r, h->cfg.num_buckets, h->cfg.max_distance);
    for (int hh = 0; hh < Hn; ++hh) t[(size_t)hh * W + (r + L - 1)] = h->rel_embed[(size_t)bk * Hn + hh];
  }
  if (!h->table) h->table = dalloc<float>(h, (int64_t)Hn * (2 * h->maxL - 1), false);
  HIPCHK(hipMemcpy(h->table, t.data(), t.size() * 4, hipMemcpyHostToDevice));
  h->table_L = L;
}

// ------------------------------------------------------------------ segmentation forward
// `lp` (bf16 engine mode): every contraction input is a bf16 buffer written by its producer
// (conv0, LayerNorm, GELU / Swish epilogues, attention, depthwise conv); the residual stream x, the
// layer-weighted sum, q/k/v, softmax and all norm statistics stay fp32.
void seg_forward(H* h, const float* wave, int B, int N, float* d_logp, uint8_t* d_ml, hipStream_t st) {
  const dzn_config& c = h->cfg;
  const bool lp = c.precision == DZN_PREC_BF16;
  const bool lnx = c.extractor_layer_norm != 0;
  int T[DZN_MAX_CONV];
  {
    int n = N;
    for (int i = 0; i < c.n_conv; ++i) {
      n = conv_out(n, c.conv_k[i], c.conv_s[i]);
      T[i] = n;
    }
  }
  const int L = T[c.n_conv - 1];
  if (L < 1) throw EngineError(DZN_E_INVALID, "window too short");
  const int D = h->D, A = h->A, Fh = h->Fh;
  const int64_t ML = (int64_t)B * L;
  // DZN_PREC_F16: contraction classes that keep BOTH fp16 terms (three products) — f16_keep2 is a bit mask over
  // F16_CLASSES below (dzn_handle::f16_keep2, default chosen from the measured sensitivity: profiles/r4_f16_sensitivity*)
  auto gemm = [&](dzn_gemm_desc& d, bool a16, bool c16, const char* what) {
    d.a_bf16 = a16;
    d.c_bf16 = c16;
    if (c.precision == DZN_PREC_F16) {
      const int cls = f16_class(what);
      const bool mx = cls >= 0 && ((h->f16_mx >> cls) & 1) && d.Wmx;
      if (!mx) d.Wmx = nullptr;     // single-term fp16 (or two terms, next line)
      if (cls >= 0 && ((h->f16_keep2 >> cls) & 1)) d.precision = DZN_PREC_F32_H2;
      else if (!mx && d.ln_stats && h->f16_center) d.ln_centered = 1;
    }
    // |max| trackers are per window: rows of a [B*L, .] tensor belong to window m / L; z-batched launches
    // (conv stack: z = window) use the z index
    // (r4 fix: a conv-stack launch with B == 1 also has nz == 1, but its rows are T_i frames of window 0, not B * L rows —
    // with unit = L its tracker index m / L ran past the one-window tracker array and scaled by whatever lay behind it)
    if ((d.nz == 1 && d.M == (int)ML) || (d.a_rowoff && d.a_rowoff == h->pos_rowoff)) d.amax_unit = L;
    chk(launch_gemm(d, st), what);
  };
  // DZN_PREC_F32_H2: |max| trackers (see dzn_handle::amax).  am(slot) is NULL in the other modes, which makes
  // every contraction take its bf16 three-term / fp32 kernel.
  const bool h2 = prec_is_h2(c.precision);
  const int64_t MB = c.max_batch;
  if (h2) {
    chk(launch_fill_u32(h->amax, 0u, dzn_handle::AM_IMG0 * MB, st), "tracker reset");
    if (lnx && h->conv0_bound > 0.f) {   // conv0 (LN + GELU) writes bufA: static bound instead of a tracker
      uint32_t bits;
      memcpy(&bits, &h->conv0_bound, 4);
      chk(launch_fill_u32(h->amax + dzn_handle::AM_CONVA * MB, bits, B, st), "conv0 bound");
    }
  }
  auto am = [&](int slot) -> float* { return h2 ? h->amax + slot * MB : nullptr; };
  auto conv_slot = [&](const float* buf) { return buf == h->bufA ? dzn_handle::AM_CONVA : dzn_handle::AM_CONVB; };

  // ---- conv feature extractor ----
  const float* stats = nullptr;
  if (c.normalize_waveform) {
    chk(launch_wave_stats(wave, B, N, 1e-5f, h->stats, st), "wave_stats");
    stats = h->stats;
  }
  // DZN_PREC_F32_H2: conv0 + LN + GELU + conv1 in one kernel (frontend_fused.hip): conv0's 52 MB / window never
  // reach HBM.  (debug taps need the intermediate -> unfused)
  const bool fuse01 = prec_is_h2(c.precision) && lnx && h->conv1_W2n && !h->debug && c.n_conv > 1 && T[1] > 0;
  if (lnx && fuse01) {
  } else if (lnx) {
    chk(launch_conv0(wave, B, N, stats, h->conv0_w, h->conv_ln[0].g, h->conv_ln[0].b, h->C[0], h->Cp[0],
                     c.conv_k[0], c.conv_s[0], T[0], 1, 1e-5f, h->bufA, lp, st, h->conv0_lnq),
        "conv0");
  } else {
    float* raw = lp ? h->craw : h->bufA;
    chk(launch_conv0(wave, B, N, stats, h->conv0_w, nullptr, nullptr, h->C[0], h->Cp[0], c.conv_k[0],
                     c.conv_s[0], T[0], 0, 1e-5f, raw, 0, st),
        "conv0");
    chk(launch_groupnorm_gelu(raw, h->bufA, lp, B, T[0], h->C[0], h->Cp[0], h->Cp[0], h->conv_ln[0].g,
                              h->conv_ln[0].b, 1e-5f, h->gn_stats, st, am(conv_slot(h->bufA))),
        "groupnorm");
  }
  tap(h, "conv0", h->bufA, (int64_t)B * T[0], h->C[0], h->Cp[0], st, lp);
  float* cur = h->bufA;
  float* nxt = h->bufB;
  const int last = c.n_conv - 1;
  for (int i = 1; i < c.n_conv; ++i) {
    // conv1d(k, s) over channels-last rows: row t of the contraction starts at (t*s)*Cp_in
    float* dst = (lnx && lp) ? h->craw : nxt;
    dzn_gemm_desc d = gd(h, cur, h->conv[i], dst, T[i], (int64_t)c.conv_s[i] * h->Cp[i - 1], h->Cp[i]);
    d.nz = B;
    d.a_z0 = (int64_t)T[i - 1] * h->Cp[i - 1];
    d.c_z0 = (int64_t)T[i] * h->Cp[i];
    if (!lnx) d.act = DZN_ACT_GELU;
    d.a_amax = am(conv_slot(cur));   // (r3: the group-norm element pass tracks its |max| too, so conv1 takes the fp16 split)
    if (!lnx) d.c_amax = am(conv_slot(nxt));
    // (r3) the fused front end also finishes conv1's own LayerNorm + GELU in its epilogue (its tile holds whole rows)
    const bool ln_in_01 = i == 1 && fuse01 && lnx && !lp && i != last && !getenv("DZN_CONV01_NO_LN");
    if (i == 1 && fuse01)
      chk(launch_conv01_fused(wave, B, N, stats, h->conv0_w, h->conv_ln[0].g, h->conv_ln[0].b, h->conv0_lnq, h->C[0],
                              T[0], T[1], h->conv1_W2n, h->conv1_wscn, h->Cp[1], h->conv0_bound, 1e-5f, ln_in_01 ? nxt : dst,
                              st, ln_in_01 ? h->conv_ln[1].g : nullptr, ln_in_01 ? h->conv_ln[1].b : nullptr, h->C[1],
                              ln_in_01 ? am(conv_slot(nxt)) : nullptr),
          "conv01 fused");
    else
      gemm(d, lp, lp && !lnx, "conv gemm");
    if (lnx && !ln_in_01)  // channel LayerNorm + GELU (+ dummy_weight after the last conv, components.py:208)
      ln_t(dst, false, h->Cp[i], nxt, lp, h->Cp[i], h->conv_ln[i], (int64_t)B * T[i], h->Cp[i], 1, st,
           i == last ? h->dummy_w : nullptr, am(conv_slot(nxt)), T[i]);
    std::swap(cur, nxt);
  }
  if (!lnx || c.n_conv == 1)
    chk(launch_col_scale(cur, lp, ML, h->C[last], h->Cp[last], h->dummy_w, st), "dummy_weight");
  tap(h, "features", cur, ML, h->C[last], h->Cp[last], st, lp);

  // ---- feature projection (components.py:305-306) ----
  const bool fold = h->fold_ln;   // LayerNorms feeding only linears are folded into those contractions
  auto folded = [&](dzn_gemm_desc& d, const Lin& l) {
    d.ln_stats = h->rstat;
    d.ln_colsum = l.csum;
  };
  if (fold) {
    chk(launch_row_stats(cur, h->Cp[last], ML, h->C[last], 1e-5f, h->rstat, st), "row_stats");
    dzn_gemm_desc d = gd(h, cur, h->fp, h->x, ML, h->Cp[last], D);
    folded(d, h->fp);
    // (col_scale of the base model multiplies conv6's output AFTER its tracker: |dummy_weight| ~ 1, and the
    // tracker only needs to bound the magnitude within the 2^15 / 65504 headroom -> skip the fp16 path there)
    if (lnx && c.n_conv > 1) d.a_amax = am(conv_slot(cur));
    d.c_amax = am(dzn_handle::AM_X);
    gemm(d, false, false, "feature projection");
  } else {
    ln_t(cur, lp, h->Cp[last], nxt, lp, h->Cp[last], h->fp_ln, ML, h->Cp[last], 0, st);
    dzn_gemm_desc d = gd(h, nxt, h->fp, h->x, ML, h->Cp[last], D);
    d.c_amax = am(dzn_handle::AM_X);
    gemm(d, lp, false, "feature projection");
  }
  tap(h, "featproj", h->x, ML, D, D, st);

  // ---- positional conv: x = x + gelu(conv_pos(x))  (components.py:980-987, 366-380) ----
  {
    const int Kc = c.pos_conv_kernel, G = c.pos_conv_groups, cg = D / G, Lp = L + Kc;
    // the LDS-DMA bf16 kernel needs kc % 32 == 0; otherwise (base: 768/16 = 48 channels per group)
    // keep this one contraction on fp32 activations (register-staged kernel converts on the fly)
    const bool pc16 = lp && (cg % 32 == 0);
    // f32s: every element of the padded copy feeds 128 taps — split it ONCE here (three bf16 planes) and
    // let the contraction read the planes (gemm_split_pre.hip) instead of re-splitting it per K tile
    const int cgp = h->pos_cgp > 0 ? h->pos_cgp : cg, Dp = G * cgp;     // plane rows: G groups of cgp channels
    const bool pre3 = h->xpad3 != nullptr && (int64_t)c.max_batch * Lp * Dp < (int64_t)1 << 31;
    const bool pre2 = pre3 && h2 && h->posconv.W2h;   // two fp16 planes, scaled by the |max| of x (snapshotted)
    if (pre2)
      chk(launch_pad_rows_split2(h->x, h->xpad3, h->xpad3_plane, B, L, Lp, Kc / 2, Dp, am(dzn_handle::AM_X),
                                 am(dzn_handle::AM_XPAD), st, cg, cgp),
          "pad_rows_split2");
    else if (pre3)
      chk(launch_pad_rows_split3(h->x, h->xpad3, h->xpad3_plane, B, L, Lp, Kc / 2, Dp, st, cg, cgp), "pad_rows_split3");
    else
      chk(launch_pad_rows(h->x, h->xpad, pc16, B, L, Lp, Kc / 2, D, st), "pad_rows");
    // the un-split copy (fp32 / bf16 modes) keeps cg channels per group; the planes have cgp
    const int ca = pre3 ? cgp : cg, Da = pre3 ? Dp : D;
    if (!pre3 && cgp != cg) throw EngineError(DZN_E_INVALID, "padded positional-conv groups need the pre-split planes");
    // one contraction per channel group over ALL B*L rows (row-offset table into the padded copy), so
    // the 128-row tiles are not padded per window (L = 399 would waste 22 % of every window's last tile)
    const bool tabled = (int64_t)c.max_batch * Lp * Da < (int64_t)1 << 31;
    if (tabled) ensure_pos_rowoff(h, L, Lp, Da, st);
    dzn_gemm_desc d = gd(h, h->xpad, h->posconv, h->x, tabled ? ML : L, Da, D);
    d.N = cg;
    d.kc = ca;
    d.ldk = Da;
    d.act = DZN_ACT_GELU;
    d.R = h->x;
    d.a_z1 = ca;
    d.w_z1 = (int64_t)cg * h->posconv.K;
    d.c_z1 = cg;
    d.b_z1 = cg;
    if (tabled) {
      d.a_rowoff = h->pos_rowoff;
      d.nz = G;
      d.zdiv = G;
      d.alg_flops = 2.0 * (double)ML * cg * h->posconv.Kt;
    } else {
      d.nz = B * G;
      d.zdiv = G;
      d.a_z0 = (int64_t)Lp * Da;
      d.c_z0 = (int64_t)L * D;
      d.alg_flops = 2.0 * (double)L * cg * h->posconv.Kt;
    }
    if (pre3) {
      d.A = reinterpret_cast<const float*>(h->xpad3);
      d.a_split3 = pre2 ? 2 : 1;
      d.a_plane = h->xpad3_plane;
      if (pre2) d.a_amax = am(dzn_handle::AM_XPAD);
    }
    d.c_amax = am(dzn_handle::AM_X);
    gemm(d, pc16, false, "pos conv");
  }
  if (!c.layer_norm_first) ln_t(h->x, false, D, h->x, false, D, h->enc_ln, ML, D, 0, st, nullptr, am(dzn_handle::AM_X), L);
  // the residual stream: `xr` = rows of the representation the next operation reads.  With the deferred layer-weighted sum
  // every layer moves it into its own buffer (the first residual update of layer i reads xr and writes xl[i], the second
  // one works in place there) and (xr, weight) is noted per layer; else xr == h->x throughout.
  const bool defer = h->xl != nullptr;
  float* xr = h->x;
  WsSumArgs wsum{};
  if (defer) {
    wsum.x[wsum.n] = xr;
    wsum.w[wsum.n++] = h->wsum_w[0];
  } else {
    chk(launch_ws_accum(h->x, h->ws, h->wsum_w[0], 1, ML * D, st), "ws_accum");
  }
  tap(h, "rep0", h->x, ML, D, D, st);

  // ---- transformer layers (components.py:920-942) ----
  static const bool stats_from_epilogue = getenv("DZN_NO_EPILOGUE_STATS") == nullptr;
  ensure_table(h, L, st);
  for (int i = 0; i < c.n_layers; ++i) {
    EncLayer& Ly = h->layers[i];
    const float wl = h->wsum_w[i + 1];
    float* const own = defer ? h->xl + (int64_t)i * ML * D : xr;   // where this layer's output rows live
    bool have_stats = false;   // LN2's statistics already left in rstat by the out_proj epilogue
    if (Ly.attn) {
      const float* yin = xr;
      bool y16 = false;
      const bool fold1 = fold && c.layer_norm_first;
      if (fold1) {
        // one pass over x: LN statistics for the folded q/k/v contraction + the gate on LN(x) (never written)
        chk(launch_gate_stats(xr, D, Ly.ln1.g, Ly.ln1.b, Ly.Wg, Ly.bg, Ly.cst, h->gate, h->rstat, ML, h->H, 1e-5f,
                              st),
            "gate_stats");
      } else if (c.layer_norm_first) {
        ln_t(xr, false, D, h->y, lp, D, Ly.ln1, ML, D, 0, st, nullptr, am(dzn_handle::AM_Y), L);
        yin = h->y;
        y16 = lp;
      } else if (lp) {
        chk(launch_cast_bf16(xr, h->y, ML * D, st), "cast");
        yin = h->y;
        y16 = true;
      }
      if (!fold1) chk(launch_gate_t(yin, y16, D, Ly.Wg, Ly.bg, Ly.cst, h->gate, ML, h->H, st), "gate");
      const int hd = Ly.h * 64;
      dzn_gemm_desc d = gd(h, yin, Ly.qkv, h->qkv, ML, D, 3 * hd);
      if (fold1) folded(d, Ly.qkv);
      d.a_amax = am(yin == xr ? dzn_handle::AM_X : dzn_handle::AM_Y);
      d.c_amax = am(dzn_handle::AM_QKV);
      const bool planes = h->kvp != nullptr && !y16 && d.W2h && d.col_scale && d.a_amax;     // the f32h contraction writes them
      const int64_t pstride = (ML + 64) * (int64_t)(2 * hd);
      if (planes) {
        d.kv_planes = h->kvp;
        d.kv_plane_stride = pstride;
        d.kv_scale = h->kvs;
        d.kv_ld = 2 * hd;
        d.kv_col0 = hd;
      }
      gemm(d, y16, false, "qkv");
      if (planes)
        chk(launch_attention_planes(h->qkv, h->kvp, pstride, h->kvs, 2 * hd, h->ao, h->gate, h->table, Ly.head_idx, B, L, Ly.h, h->H,
                                    3 * hd, hd, 0.125f, st, am(dzn_handle::AM_QKV)),
            "attention");
      else if (prec_is_split(c.precision))
        chk(launch_attention_split(h->qkv, h->ao, h->gate, h->table, Ly.head_idx, B, L, Ly.h, h->H, 3 * hd, hd,
                                   0.125f, st, am(dzn_handle::AM_QKV)),
            "attention");
      else
        chk(launch_attention_t(h->qkv, h->ao, lp, h->gate, h->table, Ly.head_idx, B, L, Ly.h, h->H, 3 * hd,
                               hd, 0.125f, st),
            "attention");
      dzn_gemm_desc o = gd(h, h->ao, Ly.out, own, ML, hd, D);
      o.R = xr;
      o.a_amax = am(dzn_handle::AM_QKV);   // rows of ao are convex combinations of v rows: |ao| <= max |qkv|
      o.c_amax = am(dzn_handle::AM_X);
      if (fold && c.layer_norm_first && Ly.ffn && stats_from_epilogue) {
        // the rows it writes are exactly what the FFN's (folded) LayerNorm normalises: leave their statistics
        o.stat_partial = h->spart;
        o.stat_final = h->rstat;
        o.stat_C = D;
        o.stat_eps = 1e-5f;
        have_stats = true;
      }
      gemm(o, lp, false, "out_proj");
      xr = own;
    }
    if (c.layer_norm_first) {
      if (Ly.ffn) {
        const float* fin = h->y;
        if (fold) {
          if (!have_stats) chk(launch_row_stats(xr, D, ML, D, 1e-5f, h->rstat, st), "row_stats");
          fin = xr;
        } else {
          ln_t(xr, false, D, h->y, lp, D, Ly.ln2, ML, D, 0, st, nullptr, am(dzn_handle::AM_Y), L);
        }
        dzn_gemm_desc f1 = gd(h, fin, Ly.f1, h->mid, ML, D, Ly.Fp);
        if (fold) folded(f1, Ly.f1);
        f1.act = DZN_ACT_GELU;
        f1.a_amax = am(fin == xr ? dzn_handle::AM_X : dzn_handle::AM_Y);
        f1.c_amax = am(dzn_handle::AM_MID);
        gemm(f1, lp, lp, "ffn1");
        dzn_gemm_desc f2 = gd(h, h->mid, Ly.f2, own, ML, Ly.Fp, D);
        f2.R = xr;
        if (!defer) {
          f2.WS = h->ws;
          f2.ldws = D;
          f2.ws_w = wl;
        }
        f2.a_amax = am(dzn_handle::AM_MID);
        f2.c_amax = am(dzn_handle::AM_X);
        gemm(f2, lp, false, "ffn2");
        xr = own;
      } else if (!defer) {
        chk(launch_ws_accum(xr, h->ws, wl, 0, ML * D, st), "ws_accum");
      }
      if (defer) {   // a layer pruned to nothing leaves xr where it was: the same rows enter the sum again with this weight
        wsum.x[wsum.n] = xr;
        wsum.w[wsum.n++] = wl;
      }
    } else {
      ln_t(xr, false, D, xr, false, D, Ly.ln1, ML, D, 0, st, nullptr, am(dzn_handle::AM_X), L);
      if (Ly.ffn) {
        const float* fin = xr;
        if (lp) {
          chk(launch_cast_bf16(xr, h->y, ML * D, st), "cast");
          fin = h->y;
        }
        dzn_gemm_desc f1 = gd(h, fin, Ly.f1, h->mid, ML, D, Ly.Fp);
        f1.act = DZN_ACT_GELU;
        f1.a_amax = am(dzn_handle::AM_X);
        f1.c_amax = am(dzn_handle::AM_MID);
        gemm(f1, lp, lp, "ffn1");
        dzn_gemm_desc f2 = gd(h, h->mid, Ly.f2, xr, ML, Ly.Fp, D);
        f2.R = xr;
        f2.a_amax = am(dzn_handle::AM_MID);
        f2.c_amax = am(dzn_handle::AM_X);
        gemm(f2, lp, false, "ffn2");
      }
      ln_t(xr, false, D, xr, false, D, Ly.ln2, ML, D, 0, st, nullptr, am(dzn_handle::AM_X), L);
      chk(launch_ws_accum(xr, h->ws, wl, 0, ML * D, st), "ws_accum");
    }
    if (h->debug) {
      const std::string nm = "layer" + std::to_string(i);
      tap(h, nm.c_str(), xr, ML, D, D, st);
    }
  }
  if (defer) chk(launch_ws_sum(wsum, h->ws, ML * D, st), "ws_sum");
  tap(h, "wsum", h->ws, ML, D, D, st);

  // ---- head: proj + LN, Conformer x conf_layers, classifier (model_wavlm_conformer.py:256-262) ----
  {
    const float* pin = h->ws;
    if (lp) {
      chk(launch_cast_bf16(h->ws, h->y, ML * D, st), "cast");
      pin = h->y;
    }
    dzn_gemm_desc d = gd(h, pin, h->proj, h->hz, ML, D, A);
    gemm(d, lp, false, "proj");   // (ws has no tracker: bf16 split)
    ln_t(h->hz, false, A, h->hz, false, A, h->lnorm, ML, A, 0, st, nullptr, am(dzn_handle::AM_HZ), L);
  }
  tap(h, "head_in", h->hz, ML, A, A, st);
  for (int i = 0; i < c.conf_layers; ++i) {
    ConfLayer& Cl = h->conf[i];
    auto half_ffn = [&](const LNp& ln, const Lin& w1, const Lin& w2) {
      const float* in = h->ht;
      if (fold) {
        chk(launch_row_stats(h->hz, A, ML, A, 1e-5f, h->rstat, st), "row_stats");
        in = h->hz;
      } else {
        ln_t(h->hz, false, A, h->ht, lp, A, ln, ML, A, 0, st);
      }
      dzn_gemm_desc a = gd(h, in, w1, h->hmid, ML, A, Fh);
      if (fold) folded(a, w1);
      a.act = DZN_ACT_SWISH;
      if (in == h->hz) a.a_amax = am(dzn_handle::AM_HZ);
      a.c_amax = am(dzn_handle::AM_HMID);
      gemm(a, lp, lp, "conf ffn w1");
      dzn_gemm_desc b = gd(h, h->hmid, w2, h->hz, ML, Fh, A);
      b.alpha = 0.5f;
      b.R = h->hz;
      b.a_amax = am(dzn_handle::AM_HMID);
      b.c_amax = am(dzn_handle::AM_HZ);
      gemm(b, lp, false, "conf ffn w2");
    };
    half_ffn(Cl.ffn1_ln, Cl.ffn1_w1, Cl.ffn1_w2);
    // MHSA
    if (fold) chk(launch_row_stats(h->hz, A, ML, A, 1e-5f, h->rstat, st), "row_stats");
    else ln_t(h->hz, false, A, h->ht, lp, A, Cl.mha_ln, ML, A, 0, st);
    {
      dzn_gemm_desc q = gd(h, fold ? h->hz : h->ht, Cl.qkv, h->hmid, ML, A, 3 * A);
      if (fold) {
        folded(q, Cl.qkv);
        q.a_amax = am(dzn_handle::AM_HZ);
      }
      q.c_amax = am(dzn_handle::AM_HMID);
      gemm(q, lp, false, "conf qkv");
      if (prec_is_split(c.precision))
        chk(launch_attention_split(h->hmid, h->hv, nullptr, nullptr, nullptr, B, L, c.conf_heads, 0, 3 * A, A,
                                   0.125f, st, am(dzn_handle::AM_HMID)),
            "conf attention");
      else
        chk(launch_attention_t(h->hmid, h->hv, lp, nullptr, nullptr, nullptr, B, L, c.conf_heads, 0, 3 * A, A,
                               0.125f, st),
            "conf attention");
      dzn_gemm_desc o = gd(h, h->hv, Cl.o, h->hz, ML, A, A);
      o.R = h->hz;
      o.a_amax = am(dzn_handle::AM_HMID);   // attention output <= max |q/k/v| (convex combinations of v)
      o.c_amax = am(dzn_handle::AM_HZ);
      gemm(o, lp, false, "conf out");
    }
    // conv module
    if (fold) chk(launch_row_stats(h->hz, A, ML, A, 1e-5f, h->rstat, st), "row_stats");
    else ln_t(h->hz, false, A, h->ht, lp, A, Cl.conv_ln, ML, A, 0, st);
    {
      dzn_gemm_desc p1 = gd(h, fold ? h->hz : h->ht, Cl.pw1, h->hmid, ML, A, 2 * A);
      if (fold) {
        folded(p1, Cl.pw1);
        p1.a_amax = am(dzn_handle::AM_HZ);
      }
      p1.c_amax = am(dzn_handle::AM_HMID);
      gemm(p1, lp, false, "conf pw1");
      chk(launch_glu_dwconv(h->hmid, 2 * A, Cl.dw, Cl.dwb, h->hv, lp, A, B, L, A, c.conf_kernel, st),
          "glu_dwconv");
      dzn_gemm_desc p2 = gd(h, h->hv, Cl.pw2, h->hz, ML, A, A);
      p2.R = h->hz;
      p2.c_amax = am(dzn_handle::AM_HZ);    // (glu_dwconv output has no tracker: bf16 split)
      gemm(p2, lp, false, "conf pw2");
    }
    half_ffn(Cl.ffn2_ln, Cl.ffn2_w1, Cl.ffn2_w2);
    ln_t(h->hz, false, A, h->hz, false, A, Cl.out_ln, ML, A, 0, st, nullptr, am(dzn_handle::AM_HZ), L);
    if (h->debug) {
      const std::string nm = "conf" + std::to_string(i);
      tap(h, nm.c_str(), h->hz, ML, A, A, st);
    }
  }
  chk(launch_classify(h->hz, A, h->cls_w, h->cls_b, h->mapping, ML, A, c.n_classes,
                      c.max_speakers_per_chunk, d_logp, d_ml, st),
      "classify");
}

// ------------------------------------------------------------------ embedding forward
// bf16 engine mode: the ResNet images are bf16 (operands, residuals and outputs of every conv);
// fbank (DFT / mel), pooling statistics and seg_1 stay fp32.
// `subset`: windows in which no speaker is active skip the trunk.  Zero weights pool to zero
// (PA/models/blocks/pooling.py:44-131 with its 1e-8 guards), so every one of their embeddings is seg_1's bias (SURVEY a18;
// tests/test_emb_gpu.py holds the device to that bit for bit).  The masks decide it and they only exist on the device, so
// the subset is built there (window_active -> compact_active: ascending list + count) and every trunk kernel is launched
// over all B grid rows with (count, list): surplus rows exit at once.  No flag is read back — the call only enqueues
// (include/dzn.h), and it can be captured in a HIP graph.  The fbank (1 % of the stage) stays dense.
void emb_forward(H* h, const float* wave, const float* masks, int B, int S, int N, int L, float* d_emb, hipStream_t st) {
  const dzn_config& c = h->cfg;
  const bool lp = c.precision == DZN_PREC_BF16;
  const int flen = 400, fshift = 160, Kp = 416, NB = c.num_mel_bins;
  if (N < flen) throw EngineError(DZN_E_INVALID, "waveform shorter than one fbank frame (400 samples)");
  const int T = 1 + (N - flen) / fshift;
  if (T > h->maxTf) throw EngineError(DZN_E_INVALID, "N exceeds max_samples");
  if (S < 1 || S > 8) throw EngineError(DZN_E_INVALID, "S must be in [1, 8]");
  emb_set_geometry(h, T, st);
  const bool subset = h->emb_skip && !h->debug && !lp && h->seg1.b != nullptr;
  const int *zc = nullptr, *zl = nullptr, *zflag = nullptr;
  if (subset) {
    chk(launch_window_active(masks, B, S * L, h->win_flag, st), "window_active");
    chk(launch_compact_active(h->win_flag, B, h->win_cnt, h->win_idx, h->emb_totals, st), "compact_active");
    zc = h->win_cnt;
    zl = h->win_idx;
    zflag = h->win_flag;
  } else {
    h->emb_dense_windows += B;   // dzn_embed_skip_stats counts every window, whichever path it took
  }
  const int64_t MT = (int64_t)B * T;
  // ---- kaldi fbank ----
  chk(launch_frame_prep(wave, B, N, T, flen, fshift, Kp, h->hamming, 0.97f, h->frames, st), "frame_prep");
  {
    dzn_gemm_desc d = gd(h, h->frames, h->dft, h->spec, MT, Kp, 512);
    d.precision = DZN_PREC_F32;  // the spectrum spans ~9 decades: keep the transform in fp32
    chk(launch_gemm(d, st), "dft");
    chk(launch_power(h->spec, MT, 256, h->pw, st), "power");
    dzn_gemm_desc m = gd(h, h->pw, h->mel, h->fb, MT, 256, NB);
    m.precision = DZN_PREC_F32;
    chk(launch_gemm(m, st), "mel");
    chk(launch_log_cmn(h->fb, B, T, NB, 1.1920928955078125e-07f, st), "log_cmn");
  }
  tap(h, "fbank", h->fb, MT, NB, NB, st);
  // ---- ResNet34 trunk ----
  const float* prev = nullptr;  // output image of the previous stage
  int cur = 0;
  // DZN_PREC_F32_H2: one |max| tracker per image buffer (sbuf[s][k] -> slot AM_IMG0 + 3 s + k), running over the
  // forward (stem kernel, stage-1 kernel and the contraction epilogues all track what they write)
  const bool h2 = prec_is_h2(c.precision);
  const int64_t MB = c.max_batch;
  if (h2) chk(launch_fill_u32(h->amax + dzn_handle::AM_IMG0 * MB, 0u, 12 * MB, st), "image tracker reset");
  auto img_am = [&](const float* buf) -> float* {
    if (!h2 || !buf) return nullptr;
    for (int s2 = 0; s2 < 4; ++s2)
      for (int k = 0; k < 3; ++k)
        if (buf == h->sbuf[s2][k]) return h->amax + (dzn_handle::AM_IMG0 + 3 * s2 + k) * MB;
    return nullptr;
  };
  chk(launch_stem_conv(h->fb, B, T, NB, h->sC[0], h->stem_w, h->stem_b, h->sbuf[0][0], lp, st, img_am(h->sbuf[0][0]), zc, zl),
      "stem");
  for (int s = 0; s < 4; ++s) {
    const int Hs = h->sH[s], Ws = h->sW[s], Cc = h->sC[s];
    const int64_t img = h->simg[s];
    const int64_t interior = ((int64_t)(Ws + 2) + 1) * Cc;
    const int M = Hs * Ws;
    auto conv3 = [&](const float* in, const ResConv& rc, float* out, const float* R, int act,
                     int post_relu) {
      if (prec_is_split(c.precision) && rc.cin == 32 && rc.cout == 32 && rc.l.W3 &&
          (act == DZN_ACT_NONE || act == DZN_ACT_RELU)) {
        // first ResNet stage: dedicated kernel, every input pixel split once instead of once per tap
        chk(launch_conv3x3_c32_split(in, rc.l.W3, rc.l.b, R, out, B, Hs, Ws, act == DZN_ACT_RELU, post_relu, st,
                                     img_am(out), rc.l.W2h, rc.l.wsc, img_am(in), zc, zl),
            "resnet conv3x3 c32");
        return;
      }
      // stride-1 3x3 inside stage s: patch rows via tab1, two-level K = (dh | dw*C + ci)
      dzn_gemm_desc d = gd(h, in, rc.l, eoff(out, interior, lp), M, 0, 0);
      d.a_rowoff = h->tab1[s];
      d.c_rowoff = h->tab1[s];
      d.kc = 3 * rc.cin;
      d.ldk = (int64_t)(Ws + 2) * rc.cin;
      d.act = act;
      d.R = R ? eoff(R, interior, lp) : nullptr;
      d.post_relu = post_relu;
      d.nz = B;
      d.a_z0 = img;
      d.c_z0 = img;
      d.a_bf16 = d.c_bf16 = d.r_bf16 = lp;
      d.a_amax = img_am(in);
      d.c_amax = img_am(out);
      d.z_count = zc;
      d.z_list = zl;
      if (c.precision == DZN_PREC_F16 && ((h->f16_keep2 >> 14) & 1)) d.precision = DZN_PREC_F32_H2;
      chk(launch_gemm(d, st), "resnet conv3x3");
    };
    for (size_t j = 0; j < h->stages[s].size(); ++j) {
      ResBlock& rb = h->stages[s][j];
      if (rb.has_sc) {
        // down-sampling block: reads the previous stage's image with stride 2 (tab2)
        const int Wp = h->sW[s - 1], Cpv = h->sC[s - 1];
        const int64_t pimg = h->simg[s - 1];
        float* midb = h->sbuf[s][0];
        float* outb = h->sbuf[s][1];
        float* scb = h->sbuf[s][2];
        dzn_gemm_desc d = gd(h, prev, rb.c1.l, eoff(midb, interior, lp), M, 0, 0);
        d.a_rowoff = h->tab2[s];
        d.c_rowoff = h->tab1[s];
        d.kc = 3 * Cpv;
        d.ldk = (int64_t)(Wp + 2) * Cpv;
        d.act = DZN_ACT_RELU;
        d.nz = B;
        d.a_z0 = pimg;
        d.c_z0 = img;
        d.a_bf16 = d.c_bf16 = lp;
        d.a_amax = img_am(prev);
        d.c_amax = img_am(midb);
        d.z_count = zc;
        d.z_list = zl;
        if (c.precision == DZN_PREC_F16 && ((h->f16_keep2 >> 14) & 1)) d.precision = DZN_PREC_F32_H2;
        chk(launch_gemm(d, st), "resnet conv3x3 s2");
        dzn_gemm_desc e = gd(h, eoff(prev, ((int64_t)(Wp + 2) + 1) * Cpv, lp), rb.sc.l,
                             eoff(scb, interior, lp), M, 0, 0);
        e.a_rowoff = h->tab2[s];
        e.c_rowoff = h->tab1[s];
        e.nz = B;
        e.a_z0 = pimg;
        e.c_z0 = img;
        e.a_bf16 = e.c_bf16 = lp;
        e.a_amax = img_am(prev);
        e.c_amax = img_am(scb);
        e.z_count = zc;
        e.z_list = zl;
        if (c.precision == DZN_PREC_F16 && ((h->f16_keep2 >> 14) & 1)) e.precision = DZN_PREC_F32_H2;
        chk(launch_gemm(e, st), "resnet shortcut");
        conv3(midb, rb.c2, outb, scb, DZN_ACT_NONE, 1);
        cur = 1;
      } else {
        float* inb = h->sbuf[s][cur];
        float* midb = h->sbuf[s][(cur + 1) % 3];
        float* outb = h->sbuf[s][(cur + 2) % 3];
        // (r4) 32-channel blocks in the two-term fp16 modes: both convolutions in one kernel, the intermediate image
        // stays in LDS (resblock_fused.hip); DZN_NO_RESBLOCK_FUSION=1 (read at dzn_create) keeps the per-conv kernels
        const bool fusable = h->fuse_resblock && prec_is_h2(c.precision) && rb.c1.cin == rb.c1.cout && rb.c1.l.W2h &&
                             rb.c2.l.W2h && img_am(inb) != nullptr;
        // (r5) DZN_PREC_F16: the fused blocks keep ONE fp16 term unless bit 14 of DZN_F16_KEEP2 asks for two (the embedding
        // trunk is three orders of magnitude inside its cosine bar with one term: profiles/r4_reduced_mode_emulation.txt)
        const int rb_np = (c.precision == DZN_PREC_F16 && !((h->f16_keep2 >> 14) & 1)) ? 1 : 2;
        if (fusable && rb.c1.cin == 32 && !(h->resblock_ws & 1)) {
          chk(launch_resblock32_fused(inb, outb, rb.c1.l.W2h, rb.c1.l.wsc, rb.c1.l.b, rb.c2.l.W2h, rb.c2.l.wsc, rb.c2.l.b,
                                      img_am(inb), img_am(outb), rb.c1.l1max, rb.c1.bmax, B, Hs, Ws, rb_np, st, zc, zl),
              "resblock32 fused");
        } else if (fusable && ((rb.c1.cin == 32 && (h->resblock_ws & 1)) || (rb.c1.cin == 64 && (h->resblock_ws & 2)))) {
          // producer / consumer wavefronts (resblock_ws.hip): the only fused form for the 64-plane stage
          chk(launch_resblock_ws(inb, outb, rb.c1.l.W2h, rb.c1.l.wsc, rb.c1.l.b, rb.c2.l.W2h, rb.c2.l.wsc, rb.c2.l.b, img_am(inb),
                                 img_am(outb), rb.c1.l1max, rb.c1.bmax, B, Hs, Ws, rb.c1.cin, st, zc, zl, rb_np),
              "resblock ws");
        } else {
          conv3(inb, rb.c1, midb, nullptr, DZN_ACT_RELU, 0);
          conv3(midb, rb.c2, outb, inb, DZN_ACT_NONE, 1);
        }
        cur = (cur + 2) % 3;
      }
    }
    prev = h->sbuf[s][cur];
  }
  // ---- TSTP pooling for all S masks + seg_1 ----
  const int feat = h->sC[3] * h->sH[3];
  chk(launch_stats_pool(prev, lp, B, h->sH[3], h->sW[3], h->sC[3], masks, S, L, h->pool, st, zflag), "stats_pool");
  tap(h, "pool", h->pool, (int64_t)B * S, 2 * feat, 2 * feat, st);
  {
    dzn_gemm_desc d = gd(h, h->pool, h->seg1, d_emb, (int64_t)B * S, 2 * feat, c.embed_out_dim);
    d.precision = DZN_PREC_F32;
    chk(launch_gemm(d, st), "seg_1");
  }
}

template <typename F>
int guarded(H* h, F&& f) {
  try {
    f();
    return DZN_OK;
  } catch (const EngineError& e) {
    if (h) h->err = e.what();
    else last_create_error = e.what();
    return e.code;
  } catch (const std::bad_alloc&) {
    if (h) h->err = "host allocation failed";
    return DZN_E_NOMEM;
  } catch (const std::exception& e) {
    if (h) h->err = e.what();
    return DZN_E_INVALID;
  }
}

}  // namespace

// ====================================================================== C ABI
extern "C" {

int dzn_create(const dzn_config* cfg, dzn_handle** out) {
  if (!cfg || !out) return DZN_E_INVALID;
  if (cfg->struct_size != (int32_t)sizeof(dzn_config)) {
    last_create_error = "dzn_config.struct_size mismatch (ABI)";
    return DZN_E_INVALID;
  }
  if (cfg->max_batch < 1 || cfg->max_samples < 400 ||
      (cfg->precision != DZN_PREC_F32 && cfg->precision != DZN_PREC_BF16 &&
       cfg->precision != DZN_PREC_F32_SPLIT && cfg->precision != DZN_PREC_F32_H2 &&
       cfg->precision != DZN_PREC_F16)) {
    last_create_error = "bad max_batch / max_samples / precision";
    return DZN_E_INVALID;
  }
#ifndef DZN_TUNING
  if (cfg->precision == DZN_PREC_BF16) {
    last_create_error = "the bf16 engine mode is quarantined: it exists only in DZN_TUNING=1 builds (reduced precision = DZN_PREC_F16)";
    return DZN_E_INVALID;
  }
#endif
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    last_create_error = "no HIP device visible (libdzn_hip has no CPU fallback)";
    return DZN_E_HIP;
  }
  dzn_handle* h = new (std::nothrow) dzn_handle();
  if (!h) return DZN_E_NOMEM;
  h->cfg = *cfg;
  if (hipGetDevice(&h->device) != hipSuccess) h->device = 0;
  h->emb_skip = getenv("DZN_EMB_NO_SKIP") == nullptr;
  h->fuse_resblock = getenv("DZN_NO_RESBLOCK_FUSION") == nullptr;
  if (const char* e = getenv("DZN_RESBLOCK_WS")) h->resblock_ws = atoi(e);
  if (const char* e = getenv("DZN_F16_KEEP2")) h->f16_keep2 = (unsigned)strtoul(e, nullptr, 0);
  if (const char* e = getenv("DZN_F16_CENTER")) h->f16_center = e[0] != '0';
  if (const char* e = getenv("DZN_F16_MX")) h->f16_mx = (unsigned)strtoul(e, nullptr, 0);
  const char* dbg = getenv("DZN_DEBUG_TAPS");
  h->debug = dbg && dbg[0] == '1';
  *out = h;
  return DZN_OK;
}

int dzn_load_tensor(dzn_handle* h, const char* key, const void* host_ptr, const int64_t* shape,
                    int32_t ndim, int32_t dtype) {
  if (!h || !key || (!host_ptr && ndim > 0) || ndim < 0 || ndim > 8) return DZN_E_INVALID;
  if (h->finalized) {
    h->err = "dzn_load_tensor after dzn_finalize_weights";
    return DZN_E_STATE;
  }
  return guarded(h, [&] {
    HostT t;
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) {
      if (shape[i] < 0) throw EngineError(DZN_E_INVALID, "negative dimension");
      t.shape.push_back(shape[i]);
      n *= shape[i];
    }
    t.v.resize((size_t)n);
    if (dtype == DZN_F32) {
      memcpy(t.v.data(), host_ptr, (size_t)n * 4);
    } else if (dtype == DZN_F64) {
      const double* p = static_cast<const double*>(host_ptr);
      for (int64_t i = 0; i < n; ++i) t.v[(size_t)i] = (float)p[i];
    } else if (dtype == DZN_I64) {
      const int64_t* p = static_cast<const int64_t*>(host_ptr);
      for (int64_t i = 0; i < n; ++i) t.v[(size_t)i] = (float)p[i];
    } else {
      throw EngineError(DZN_E_INVALID, "unsupported dtype");
    }
    h->sd[key] = std::move(t);
  });
}

int dzn_finalize_weights(dzn_handle* h) {
  if (!h) return DZN_E_INVALID;
  if (h->finalized) {
    h->err = "already finalized";
    return DZN_E_STATE;
  }
  DeviceGuard dg(h->device);
  int rc = guarded(h, [&] {
    h->mx_pack = true;
    finalize_seg(h);
    h->mx_pack = false;
    h->has_emb = h->cfg.has_embedding != 0;
    if (h->has_emb) finalize_emb(h);
    alloc_optional_workspace(h);
    HIPCHK(hipDeviceSynchronize());
  });
  if (rc == DZN_OK) {
    h->finalized = true;
    h->ignored = (int)h->sd.size() - (int)h->used.size();
    h->sd.clear();
  }
  return rc;
}

int dzn_num_frames(const dzn_handle* h, int32_t num_samples) {
  if (!h) return DZN_E_INVALID;
  int n = num_samples;
  for (int i = 0; i < h->cfg.n_conv; ++i) n = conv_out(n, h->cfg.conv_k[i], h->cfg.conv_s[i]);
  return n;
}

int dzn_segment_forward(dzn_handle* h, const float* d_wave, int32_t B, int32_t N, float* d_logp,
                        uint8_t* d_multilabel, void* hip_stream) {
  if (!h) return DZN_E_INVALID;
  if (!h->finalized) {
    h->err = "dzn_segment_forward before dzn_finalize_weights";
    return DZN_E_STATE;
  }
  if (!d_wave || B < 1 || B > h->cfg.max_batch || N > h->cfg.max_samples || N < 1) {
    h->err = "bad B / N (exceeds max_batch / max_samples?)";
    return DZN_E_INVALID;
  }
  DeviceGuard dg(h->device);
  return guarded(h, [&] {
    seg_forward(h, d_wave, B, N, d_logp, d_multilabel, reinterpret_cast<hipStream_t>(hip_stream));
  });
}

int dzn_embed_forward(dzn_handle* h, const float* d_wave, const float* d_masks, int32_t B, int32_t S,
                      int32_t N, int32_t L, float* d_emb, void* hip_stream) {
  if (!h) return DZN_E_INVALID;
  if (!h->finalized || !h->has_emb) {
    h->err = "embedding model not loaded / not finalized";
    return DZN_E_STATE;
  }
  if (!d_wave || !d_masks || !d_emb || B < 1 || B > h->cfg.max_batch || N > h->cfg.max_samples ||
      L < 1) {
    h->err = "bad arguments to dzn_embed_forward";
    return DZN_E_INVALID;
  }
  DeviceGuard dg(h->device);
  return guarded(h, [&] {
    emb_forward(h, d_wave, d_masks, B, S, N, L, d_emb, reinterpret_cast<hipStream_t>(hip_stream));
  });
}

int dzn_prepare_masks(dzn_handle* h, const uint8_t* d_multilabel, int32_t B, int32_t L,
                      int32_t median_size, int32_t exclude_overlap, int32_t min_num_frames,
                      uint8_t* d_filtered, float* d_masks, void* hip_stream) {
  if (!h || !d_multilabel) return DZN_E_INVALID;
  if (!h->finalized) {
    h->err = "dzn_prepare_masks before dzn_finalize_weights";
    return DZN_E_STATE;
  }
  if (B < 1 || B > h->cfg.max_batch || L < 1 || (!d_filtered && !d_masks)) {
    h->err = "bad arguments to dzn_prepare_masks (B exceeds max_batch?)";
    return DZN_E_INVALID;
  }
  DeviceGuard dg(h->device);
  return guarded(h, [&] {
    chk(launch_prepare_masks(d_multilabel, B, L, h->cfg.max_speakers_per_chunk, median_size,
                             exclude_overlap, min_num_frames, d_filtered, d_masks,
                             reinterpret_cast<hipStream_t>(hip_stream)),
        "prepare_masks");
  });
}

int dzn_debug_fetch(dzn_handle* h, const char* name, float* host_out, int64_t cap, int64_t* n_elems) {
  if (!h || !name) return DZN_E_INVALID;
  auto it = h->taps.find(name);
  if (it == h->taps.end()) {
    h->err = std::string("no such tap (set DZN_DEBUG_TAPS=1 before dzn_create): ") + name;
    return DZN_E_INVALID;
  }
  if (n_elems) *n_elems = (int64_t)it->second.size();
  if (host_out) {
    if (cap < (int64_t)it->second.size()) return DZN_E_INVALID;
    memcpy(host_out, it->second.data(), it->second.size() * 4);
  }
  return DZN_OK;
}

int dzn_embed_skip_stats(const dzn_handle* h, int64_t* windows, int64_t* skipped) {
  if (!h) return DZN_E_INVALID;
  long long t[2] = {0, 0};
  if (h->emb_totals) {      // the totals live on the device (the forward never reads them back): this call synchronises
    DeviceGuard dg(h->device);
    if (hipDeviceSynchronize() != hipSuccess ||
        hipMemcpy(t, h->emb_totals, sizeof(t), hipMemcpyDeviceToHost) != hipSuccess)
      return DZN_E_HIP;
  }
  t[0] += h->emb_dense_windows;
  if (windows) *windows = (int64_t)t[0];
  if (skipped) *skipped = (int64_t)t[1];
  return DZN_OK;
}

int dzn_num_ignored(const dzn_handle* h) { return h ? h->ignored : 0; }
int64_t dzn_workspace_bytes(const dzn_handle* h) { return h ? h->bytes : 0; }

const char* dzn_last_error(const dzn_handle* h) {
  return h ? h->err.c_str() : last_create_error.c_str();
}

int dzn_destroy(dzn_handle* h) {
  if (!h) return DZN_OK;
  DeviceGuard dg(h->device);
  for (void* p : h->allocs) (void)hipFree(p);
  delete h;
  return DZN_OK;
}

const char* dzn_version(void) {
#ifdef DZN_TUNING
  return "dzn-hip 0.3.0 (gfx950, MFMA f32 / fp16x2 / bf16x3 / fp16 + MX fp8) [tuning build: probe tiles + the quarantined bf16 engine mode]";
#else
#ifdef DZN_CHECKED
  return "dzn-hip 0.3.0 (gfx950, MFMA f32 / fp16x2 / bf16x3 / fp16 + MX fp8) [checked build: device-side bounds assertions]";
#else
  return "dzn-hip 0.3.0 (gfx950, MFMA f32 / fp16x2 / bf16x3 / fp16 + MX fp8)";
#endif
#endif
}

// CHECKED builds (csrc/checked.h): failed device-side checks of every translation unit since the last reset.
// out4 = {failed checks, id of the first, its workgroup, its detail value}; returns the number of failed checks, 0 in a clean
// run, DZN_E_STATE when the library is not a checked build.
#ifdef DZN_CHECKED
extern "C" {
int dzn_checked_collect_gemm_split(unsigned int*, int);
int dzn_checked_collect_gemm_mx(unsigned int*, int);
int dzn_checked_collect_gemm_split_pre(unsigned int*, int);
int dzn_checked_collect_resblock_fused(unsigned int*, int);
int dzn_checked_collect_resblock_ws(unsigned int*, int);
int dzn_checked_collect_attention_split(unsigned int*, int);
int dzn_checked_collect_attention_planes(unsigned int*, int);
int dzn_checked_collect_frontend_fused(unsigned int*, int);
}
#endif
int dzn_checked_status(uint32_t* out4, int32_t reset) {
#ifdef DZN_CHECKED
  typedef int (*collect_fn)(unsigned int*, int);
  const collect_fn fns[] = {dzn_checked_collect_gemm_split, dzn_checked_collect_gemm_mx, dzn_checked_collect_gemm_split_pre,
                            dzn_checked_collect_resblock_fused, dzn_checked_collect_resblock_ws,
                            dzn_checked_collect_attention_split, dzn_checked_collect_attention_planes,
                            dzn_checked_collect_frontend_fused};
  unsigned int tot[4] = {0u, 0u, 0u, 0u};
  for (collect_fn f : fns) {
    unsigned int w[4] = {0u, 0u, 0u, 0u};
    if (f(w, reset) != 0) return DZN_E_HIP;
    if (w[0] && !tot[0]) { tot[1] = w[1]; tot[2] = w[2]; tot[3] = w[3]; }
    tot[0] += w[0];
  }
  if (out4) for (int i = 0; i < 4; ++i) out4[i] = tot[i];
  return (int)tot[0];
#else
  (void)out4;
  (void)reset;
  return DZN_E_STATE;
#endif
}

int dzn_op_relpos_bucket(int32_t rel, int32_t num_buckets, int32_t max_distance) {
  return relpos_bucket(rel, num_buckets, max_distance);
}

}  // extern "C"
