// common.h — shared device/host helpers for the gfx950 kernels of libdzn_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <string.h>
#include "../../include/dzn.h"
#include "../../include/dzn_ops.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

#define DZN_WAVE 64

// ---- wave-level reductions (64-wide wavefront) ----
// Sum over the 64 lanes with DPP cross-lane VALU ops (no LDS round trips, unlike __shfl_xor ->
// ds_bpermute): quad_perm swaps, row rotations inside each 16-lane row, then row_bcast15 /
// row_bcast31 fold the four rows into lane 63, which is read back as a wave-uniform scalar.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  const int t = __builtin_amdgcn_update_dpp(ROW_MASK == 0xF ? __float_as_int(v) : 0, __float_as_int(v), CTRL,
                                            ROW_MASK, 0xF, false);
  return v + __int_as_float(t);
}
__device__ __forceinline__ float wave_sum(float v) {
  v = dpp_add<0xB1, 0xF>(v);   // quad_perm:[1,0,3,2]
  v = dpp_add<0x4E, 0xF>(v);   // quad_perm:[2,3,0,1]
  v = dpp_add<0x124, 0xF>(v);  // row_ror:4
  v = dpp_add<0x128, 0xF>(v);  // row_ror:8  -> every lane holds its row's sum
  v = dpp_add<0x142, 0xA>(v);  // row_bcast:15 into rows 1, 3
  v = dpp_add<0x143, 0xC>(v);  // row_bcast:31 into rows 2, 3 -> lanes 48..63 hold the total
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// |max| tracker of an activation tensor (DZN_PREC_F32_H2: the consumer scales its fp16 split by it).  One call per
// wavefront with the wave's partial maximum.  |x| >= 0, so the IEEE bit patterns order like unsigned integers and
// the atomic max is order independent (deterministic).  The tracker is read first (device-scope load: the value
// only grows within a forward, so a stale read can only cost a redundant atomic) — without that check millions of
// same-address atomics serialise (measured: 7 ms for a LayerNorm over 3.3 M rows).
__device__ __forceinline__ void track_amax_lane(float* tracker, float m) {
  if (m > 0.f) {
    const unsigned cur = __hip_atomic_load(reinterpret_cast<unsigned int*>(tracker), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__float_as_uint(m) > cur) atomicMax(reinterpret_cast<unsigned int*>(tracker), __float_as_uint(m));
  }
}
__device__ __forceinline__ void track_amax(float* tracker, float wave_partial) {
  const float m = wave_max(wave_partial);
  if ((threadIdx.x & 63) == 0) track_amax_lane(tracker, m);
}

// ---- activations (match torch CPU fp32 semantics) ----
// erf to 1.5e-7 ABSOLUTE (Abramowitz & Stegun 7.1.26) in ~16 branch-free VALU ops: 1 / (1 + p|x|) on v_rcp_f32, a
// 5-term Horner polynomial, exp(-x^2) on v_exp_f32.  libm's erff costs ~40 issue slots per element once both of its
// branches are live in a wavefront, which made the GELU sites VALU-bound (conv0: 3.3 G elements per batch, measured
// 2.7 TB/s of a 13 GB write stream; the FFN / LayerNorm+GELU epilogues likewise).  The error is below one fp32 ulp of
// 1 + erf, i.e. |gelu_fast - gelu| <= 0.5 |x| 1.5e-7: inside the round-off of the contractions around it
// (tests/test_ops_gpu.py::test_gelu_accuracy, and every model-level parity test runs through it).
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
  const float y = fmaf(-p * t, e, 1.0f);
  return copysignf(y, x);
}
__device__ __forceinline__ float gelu_erf(float x) {
  // torch.nn.functional.gelu(approximate='none'): 0.5 * x * (1 + erf(x / sqrt(2)))
  return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float swishf_(float x) { return x * sigmoidf_(x); }

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case DZN_ACT_GELU: return gelu_erf(v);
    case DZN_ACT_SWISH: return swishf_(v);
    case DZN_ACT_RELU: return fmaxf(v, 0.0f);
    default: return v;
  }
}

// ---- activation element access: fp32 buffers, or bf16 buffers in the bf16 engine mode ----
__device__ __forceinline__ float ld_act(const float* p, int64_t i) { return p[i]; }
__device__ __forceinline__ float ld_act(const u16* p, int64_t i) {
  return __uint_as_float(((unsigned int)p[i]) << 16);
}
__device__ __forceinline__ void st_act(float* p, int64_t i, float v) { p[i] = v; }
__device__ __forceinline__ void st_act(u16* p, int64_t i, float v) {
  __bf16 h = (__bf16)v;  // round to nearest even
  p[i] = *reinterpret_cast<u16*>(&h);
}

// host-side float -> bf16 (round to nearest even), used when packing weights
static inline u16 f32_to_bf16_host(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u16)((u >> 16) | 0x40);  // NaN
  uint32_t r = 0x7fffu + ((u >> 16) & 1u);
  return (u16)((u + r) >> 16);
}

// DZN_PREC_F32_H2 is DZN_PREC_F32_SPLIT for every kernel that has no two-term fp16 variant
// (DZN_PREC_F16 is the DZN_PREC_F32_H2 engine whose plain contractions keep one fp16 term: gemm_split*.hip NP = 1)
static inline bool prec_is_h2(int p) { return p == DZN_PREC_F32_H2 || p == DZN_PREC_F16; }
static inline bool prec_is_split(int p) { return p == DZN_PREC_F32_SPLIT || prec_is_h2(p); }

// ---- kernel launchers (implemented in the .hip files) ----
int launch_gemm(const dzn_gemm_desc& d, hipStream_t s);
#ifdef DZN_TUNING
int launch_gemm_lowp(const dzn_gemm_desc& d, hipStream_t s);  // gemm_lowp.hip: A and W both bf16 (bf16 engine mode: DZN_TUNING builds only)
#endif
int launch_gemm_split(const dzn_gemm_desc& d, hipStream_t s); // gemm_split.hip: fp32 via 3-way bf16 split
int launch_gemm_split_pre(const dzn_gemm_desc& d, hipStream_t s);  // gemm_split_pre.hip: A pre-split planes
int launch_pad_rows_split3(const float* x, void* planes, int64_t plane_stride, int B, int L, int Lp, int pad, int D,
                           hipStream_t st, int cg = 0, int cgp = 0);   // D = plane row width (groups padded from cg to cgp channels)
int launch_pad_rows_split2(const float* x, void* planes, int64_t plane_stride, int B, int L, int Lp, int pad, int D,
                           const float* amax, float* snapshot, hipStream_t st, int cg = 0, int cgp = 0);
int launch_split_weights(const float* W, int64_t rows, int K, int64_t ldw, void* W3, hipStream_t s);
int launch_split_weights_h2(const float* W, int64_t rows, int K, int64_t ldw, void* W2, float* col_scale, hipStream_t s);
// linkage.hip (r5): the host stage's device context — one non-blocking highest-priority stream per device, and a second grow-only
// arena for an opaque STATE that lives across several calls (vbx.hip): leased to one state at a time.  host_state_lease returns
// DZN_OK with *base == nullptr when the arena is already leased (the caller then owns a plain allocation); device < 0 = current.
int host_stage_stream(int device, hipStream_t* stream);
int host_state_lease(int device, size_t bytes, hipStream_t* stream, char** base);
void host_state_release(int device);
int launch_gemm_mx(const dzn_gemm_desc& d, hipStream_t s);   // gemm_mx.hip: fp16 hi*hi + fp8 cross terms (DZN_PREC_F16)
int launch_split_weights_mx(const float* W, int64_t rows, int K, int64_t ldw, void* Wmx, float* col_scale, hipStream_t s);
int launch_amax(const float* x, int64_t n, float* amax, hipStream_t s);
int launch_fill_u32(void* p, unsigned value, int64_t n, hipStream_t st);   // norm.hip: p[0 .. n) = value, as a kernel (graph-safe ordering)
int launch_stats_finalize(const float* partial, int64_t rows, int P, int C, float eps, float* stats, hipStream_t s);
int launch_layernorm(const float* x, int64_t ldx, float* y, int64_t ldy, const float* g,
                     const float* b, int64_t rows, int C, int Cpad, float eps, int gelu,
                     hipStream_t s);
int launch_layernorm_t(const void* x, int x_bf16, int64_t ldx, void* y, int y_bf16, int64_t ldy,
                       const float* g, const float* b, const float* post, int64_t rows, int C, int Cpad,
                       float eps, int gelu, hipStream_t s, float* amax = nullptr, int64_t amax_unit = 0);
int launch_cast_bf16(const float* x, void* y, int64_t n, hipStream_t s);
int launch_row_stats(const float* x, int64_t ldx, int64_t rows, int C, float eps, float* stats, hipStream_t s);
int launch_gate_stats(const float* x, int64_t ldx, const float* gamma, const float* beta, const float* Wg,
                      const float* bg, const float* cst, float* gate, float* stats, int64_t rows, int Htot, float eps,
                      hipStream_t s);
int launch_wave_stats(const float* w, int B, int N, float eps, float* stats, hipStream_t s);
int launch_gate(const float* y, int64_t ldy, const float* Wg, const float* bg, const float* cst,
                float* gate, int64_t rows, int Htot, hipStream_t s);
int launch_gate_t(const void* y, int y_bf16, int64_t ldy, const float* Wg, const float* bg,
                  const float* cst, float* gate, int64_t rows, int Htot, hipStream_t s);
int launch_attention_t(const float* qkv, void* out, int out_bf16, const float* gate, const float* table,
                       const int32_t* head_idx, int B, int L, int h, int Htot, int ldqkv, int ldo,
                       float scale, hipStream_t s);
int launch_attention_split(const float* qkv, float* out, const float* gate, const float* table,
                           const int32_t* head_idx, int B, int L, int h, int Htot, int ldqkv, int ldo,
                           float scale, hipStream_t s, const float* amax = nullptr);  // attention_split.hip
                           // (amax = per-window |max| of qkv -> fp16 two-term variant, else bf16 three-term)
int launch_attention(const float* qkv, float* out, const float* gate, const float* table,
                     const int32_t* head_idx, int B, int L, int h, int Htot, int ldqkv, int ldo,
                     float scale, int precision, hipStream_t s);
// attention_planes.hip (r6): K / V arrive as the fp16 two-term planes the q/k/v contraction's epilogue wrote (dzn_gemm_desc.kv_planes)
int launch_attention_planes(const float* qkv, const void* planes, int64_t plane_stride, const float* kvs, int kv_ld, float* out,
                            const float* gate, const float* table, const int32_t* head_idx, int B, int L, int h, int Htot,
                            int ldqkv, int ldo, float scale, hipStream_t s, const float* amax);

// frontend.hip
int launch_conv0(const float* wave, int B, int N, const float* stats, const float* w,
                 const float* gamma, const float* beta, int C0, int Cp, int k, int s, int T0,
                 int layer_norm, float eps, void* out, int out_bf16, hipStream_t st, const float* lnq = nullptr);
int launch_groupnorm_gelu(const float* x, void* y, int y_bf16, int B, int T, int C, int Cp, int64_t ld,
                          const float* gamma, const float* beta, float eps, float* stats, hipStream_t st, float* amax = nullptr);
int64_t gn_stats_floats(int B, int T, int C, int Cp);   // size of launch_groupnorm_gelu's `stats` (statistics + chunk partials)
int launch_pad_rows(const float* x, void* xpad, int out_bf16, int B, int L, int Lp, int pad, int D,
                    hipStream_t st);
int launch_ws_accum(const float* x, float* ws, float w, int init, int64_t n, hipStream_t st);
// ws = ((0 + w[0] x[0]) + w[1] x[1]) + ... in that order, element-wise over n floats: the layer-weighted sum in ONE pass over
// the per-layer buffers (the same additions, in the same order, as n ws_accum / epilogue read-modify-writes)
constexpr int WS_SUM_MAX = 40;
struct WsSumArgs {
  const float* x[WS_SUM_MAX];
  float w[WS_SUM_MAX];
  int n;
};
int launch_ws_sum(const WsSumArgs& a, float* ws, int64_t n, hipStream_t st);
int launch_col_scale(void* x, int x_bf16, int64_t rows, int C, int64_t ld, const float* scale,
                     hipStream_t st);
// frontend_fused.hip (DZN_PREC_F32_H2): conv0 + LN + GELU + conv1 in one kernel
int launch_split_weights_h2_natural(const float* W, int64_t rows, int K, void* W2, float* col_scale, hipStream_t s);
int launch_fragment_major(const void* W2, int rows, int K, void* out, hipStream_t s);   // frontend_fused.hip: conv1 planes for the producer / consumer kernel
int launch_conv01_fused(const float* wave, int B, int N, const float* wstats, const float* w0, const float* gamma0,
                        const float* beta0, const float* lnq, int C0, int T0, int T1, const void* W2h,
                        const float* col_scale, int N1p, float act_bound, float eps, float* out, hipStream_t st,
                        const float* gamma1 = nullptr, const float* beta1 = nullptr, int C1 = 0, float* amax1 = nullptr);
// conformer.hip
int launch_glu_dwconv(const float* u, int64_t ldu, const float* w, const float* bias, void* out,
                      int out_bf16, int64_t ldo, int B, int L, int A, int ks, hipStream_t st);
int launch_classify(const float* z, int64_t ldz, const float* W, const float* bias,
                    const uint8_t* mapping, int64_t rows, int A, int NC, int S, float* logp,
                    uint8_t* multilabel, hipStream_t st);
// embed.hip
int launch_frame_prep(const float* wave, int B, int N, int T, int flen, int fshift, int Kp,
                      const float* window, float preemph, float* frames, hipStream_t st);
int launch_power(const float* spec, int64_t rows, int nb, float* pw, hipStream_t st);
int launch_window_active(const float* masks, int B, int per_window, int* flag, hipStream_t st);
int launch_compact_active(const int* flag, int B, int* count, int* list, long long* totals, hipStream_t st);
int launch_log_cmn(float* mel, int B, int T, int NB, float eps, hipStream_t st);
int launch_stem_conv(const float* fb, int B, int T, int NB, int C, const float* w, const float* bias,
                     void* img, int out_bf16, hipStream_t st, float* amax = nullptr, const int* z_count = nullptr,
                     const int* z_list = nullptr);   // (z_count, z_list): device-chosen subset of the B images
int launch_stats_pool(const void* img, int in_bf16, int B, int H, int W, int C, const float* masks, int S,
                      int L, float* stats, hipStream_t st, const int* active = nullptr);   // active[b] == 0 -> zeros

// conv_split.hip: 3x3 stride-1 conv 32 -> 32 over zero-bordered NHWC images (DZN_PREC_F32_SPLIT)
int launch_conv3x3_c32_split(const float* in, const void* W3, const float* bias, const float* R, float* out, int B,
                             int Hs, int Ws, int relu, int post_relu, hipStream_t s, float* amax = nullptr,
                             const void* W2h = nullptr, const float* col_scale = nullptr, const float* amax_in = nullptr,
                             const int* z_count = nullptr, const int* z_list = nullptr);

// resblock_fused.hip (r4): one BasicBlock of the 32-channel ResNet stage, intermediate kept in LDS (fp16 two-term modes)
int launch_resblock32_fused(const float* in, float* out, const void* W1, const float* cs1, const float* b1, const void* W2,
                            const float* cs2, const float* b2, const float* amax_in, float* amax_out, float l1max1,
                            float bmax1, int B, int Hs, int Ws, int np, hipStream_t s, const int* z_count = nullptr,
                            const int* z_list = nullptr);

// resblock_ws.hip (r4): the same block with producer / consumer wavefronts, C = 32 or 64 planes
int launch_resblock_ws(const float* in, float* out, const void* W1, const float* cs1, const float* b1, const void* W2,
                       const float* cs2, const float* b2, const float* amax_in, float* amax_out, float l1max1, float bmax1,
                       int B, int Hs, int Ws, int C, hipStream_t s, const int* z_count = nullptr, const int* z_list = nullptr,
                       int np = 2);   // np = 1 (r5): leading fp16 term only (DZN_PREC_F16)

int op_resblock_np();   // resblock_fused.hip: terms per operand of the dzn_op_resblock* test entry points (dzn_op_set_resblock_np)

// post.hip
int launch_prepare_masks(const uint8_t* ml, int B, int L, int S, int median, int exclude_overlap,
                         int min_num_frames, uint8_t* filtered, float* masks, hipStream_t st);

// prof.cpp
bool prof_enabled();
int prof_begin(hipStream_t st, const char* cls, double flops, double bytes);
void prof_end(int id, hipStream_t st);

// true exactly once per (call site, HIP device): hipFuncSetAttribute is per device
static inline bool first_use_on_device(unsigned long long& mask) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (mask & bit) return false;
  mask |= bit;
  return true;
}

// algorithmic HBM bytes of one contraction launch (profiling only): every operand once — A (rows that overlap, as in
// the conv-as-contraction layouts, count once: min(K, lda) per row), the weight planes, C, the residual, the
// layer-weighted-sum read-modify-write
static inline double gemm_alg_bytes(const dzn_gemm_desc& d, int w_bytes_per_elem) {
  const double nz = d.nz > 0 ? d.nz : 1;
  const double a_cols = d.a_rowoff ? (double)d.kc : (d.lda > 0 && d.lda < d.K ? (double)d.lda : (double)d.K);
  double b = (double)d.M * a_cols * (d.a_bf16 ? 2.0 : 4.0) * nz;
  b += (double)d.N * d.K * w_bytes_per_elem * ((d.w_z0 || d.w_z1) ? nz : 1.0);
  b += (double)d.M * d.N * (d.c_bf16 ? 2.0 : 4.0) * nz;
  if (d.R) b += (double)d.M * d.N * (d.r_bf16 ? 2.0 : 4.0) * nz;
  if (d.WS) b += 2.0 * (double)d.M * d.N * 4.0 * nz;
  return b;
}

// brackets everything a launcher enqueues with one profiler record (no-op unless dzn_profile_enable(1))
struct ProfScope {
  int id;
  hipStream_t st;
  ProfScope(hipStream_t s, const char* cls, double flops = 0.0, double bytes = 0.0)
      : id(prof_enabled() ? prof_begin(s, cls, flops, bytes) : -1), st(s) {}
  ~ProfScope() { prof_end(id, st); }
  ProfScope(const ProfScope&) = delete;
  ProfScope& operator=(const ProfScope&) = delete;
};

// Makes `dev` the calling thread's current HIP device for the lifetime of the object and RESTORES the previous one
// (every C-ABI entry point that takes a device ordinal or a handle owning one: engine.cpp, linkage.hip, vbx.hip).
// dev < 0 = leave the current device alone.
struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    if (dev < 0) return;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    else prev = -1;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }
