/*
 * dzn_ops.h — kernel-level entry points of libdzn_hip.so.
 *
 * These expose the individual gfx950 kernels the engine is built from, on raw device
 * pointers, so that tests/ can check each kernel against a plain fp32 reference of the
 * same op (torch on the host side) and bench.py can time a kernel in isolation.  They
 * are not part of the drop-in surface (that is dzn.h); they have no reference-side
 * counterpart other than the torch op named in each comment.
 *
 * All functions enqueue on `stream` and return 0 / negative DZN_E_* (see dzn.h).
 */
#ifndef DZN_OPS_H_
#define DZN_OPS_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* activations applied to (acc + bias) */
#define DZN_ACT_NONE 0
#define DZN_ACT_GELU 1  /* exact erf GELU, torch.nn.functional.gelu */
#define DZN_ACT_SWISH 2 /* x * sigmoid(x) */
#define DZN_ACT_RELU 3

/*
 * Generic MFMA contraction  C[m,n] = epilogue( sum_k A(m,k) * W[n,k] ).
 *
 *   A(m,k)   = A[ zA + rowbase(m) + (k / kc) * ldk + (k % kc) ]
 *   rowbase  = a_rowoff ? a_rowoff[m] : m * lda            (element offsets)
 *   W[n,k]   = W[ zW + n * ldw + k ]                        (torch Linear / packed conv layout)
 *   acc      = rstd[m] * (acc - mean[m] * ln_colsum[n])      if ln_stats  (LayerNorm folded into the weights)
 *   v        = act(acc + bias[zB + n]) * alpha
 *   v       += R[ zC + crow(m) + n ]        if R            crow = c_rowoff ? c_rowoff[m] : m*ldc
 *   v        = max(v, 0)                    if post_relu
 *   C[ zC + crow(m) + n ] = v
 *   WS[ m*ldws + n ] (+)= ws_w * v          if WS   ('=' when ws_init)
 *
 * The two-level K addressing (kc, ldk) turns 1-D / 2-D convolutions over channels-last
 * activations into this same contraction without an im2col copy.  grid.z batches:
 * z -> (z0 = z / zdiv, z1 = z % zdiv), zA = z0*a_z0 + z1*a_z1 etc.
 * Requirements: K % 32 == 0, kc % 32 == 0, all strides/offsets multiples of 4 elements.
 */
typedef struct dzn_gemm_desc {
  const float* A;
  const float* W;        /* fp32 weights (precision f32) */
  const void* W16;       /* bf16 weights, same layout (precision bf16); may be NULL */
  float* C;
  const float* bias;
  const float* R;
  float* WS;
  const int32_t* a_rowoff;
  const int32_t* c_rowoff;
  int32_t M, N, K;
  int64_t lda;
  int32_t kc;
  int64_t ldk;
  int32_t ldw;
  int64_t ldc;
  int64_t ldws;
  int32_t act;
  float alpha;
  int32_t post_relu;
  float ws_w;
  int32_t ws_init;
  int32_t nz, zdiv;
  int64_t a_z0, a_z1, w_z0, w_z1, c_z0, c_z1, b_z0, b_z1;
  int32_t precision;     /* DZN_PREC_* */
  double alg_flops;      /* algorithmic flops of this launch for profiling (0 -> 2*M*N*K*nz) */
  /* bf16 mode only: element type of the activation buffers (pointers above are then bf16 data) */
  int32_t a_bf16;        /* A holds bf16 (both operands arrive by LDS-DMA)               */
  int32_t c_bf16;        /* C is written as bf16 (requires a_bf16)                       */
  int32_t r_bf16;        /* R is read as bf16 (requires a_bf16)                          */
  /* DZN_PREC_F32_SPLIT only: the weights pre-split into three bf16 planes by dzn_op_split_weights
   * (same z / row offsets as W, times 3); NULL, K % 32 or kc % 32 != 0 -> fp32 MFMA kernel */
  const void* W3;
  /* DZN_PREC_F32_SPLIT only: A is ALREADY split — `A` points at plane 0 of three bf16 planes a_plane
   * elements apart, written by the producer in the weight planes' k order (csrc/gemm_split_pre.hip) */
  int32_t a_split3;
  int64_t a_plane;
  /* LayerNorm folded into this contraction (pre-norm sites: y = LN(x) feeds only linears): A is the RAW row x,
   * W already carries gamma (W' = W diag(gamma)), bias carries W beta, and the epilogue finishes the norm with the
   * per-row statistics: acc = rstd[m] * (acc - mean[m] * ln_colsum[n]), ln_colsum[n] = sum_k W'[n,k].
   * ln_stats = f32 [M][2] (mean, rstd) written by dzn_op_row_stats / the fused gate kernel; NULL = plain. */
  const float* ln_stats;
  const float* ln_colsum;
  /* DZN_PREC_F32_H2 ("f32h"): two-term fp16 split, three products (csrc/gemm_split.hip, NP = 2).
   * W2h = planes from dzn_op_split_weights_h2 ([rows][K/32][2][32] fp16 of w * 2^e_row), col_scale[n] = 2^-e_row,
   * a_amax = device array of per-unit bounds >= max |A| (tracked by the producer of A through c_amax, or dzn_op_amax).  Any of them
   * NULL -> the bf16 three-term kernel (W3).  c_amax (any mode): atomically max'ed with |C| as stored. */
  const void* W2h;
  const float* col_scale;
  const float* a_amax;
  float* c_amax;
  /* a_amax / c_amax are ARRAYS with one |max| per scale unit, so that a window's result does not depend on the rest
   * of the batch: unit of row m = m / amax_unit when amax_unit > 0 (rows of one window), else the z0 batch index. */
  int32_t amax_unit;
  /* LayerNorm statistics of the OUTPUT rows for a following folded LayerNorm: the epilogue leaves per-row partial
   * (sum, sum of squares) in stat_partial (f32 [M][P][2] scratch, P = column tiles x wavefront columns <= 32) and the
   * launcher reduces them to stat_final f32 [M][2] = (mean, rstd) over stat_C columns (plain z == 1 launches only). */
  float* stat_partial;
  float* stat_final;
  int32_t stat_C;
  float stat_eps;
  /* z-batched launches over a SUBSET of the z0 indices that is chosen ON THE DEVICE (the embedding trunk skips windows
   * without an active speaker without a host round trip): the launch still has nz grid rows; grid row y works on
   * z0 = z_list[y / zdiv] when y / zdiv < z_count[0] and exits otherwise.  NULL = every z0. */
  const int32_t* z_count;
  const int32_t* z_list;
  /* DZN_PREC_F16 with ln_stats: the single-term kernel subtracts the row mean BEFORE it rounds A to fp16 and the epilogue
   * only multiplies by rstd — C = rstd * ((x - mean) W'^T) + bias, LayerNorm as defined, instead of the folded form
   * rstd * (x W'^T - mean colsum), whose two terms cancel to the size of the rounding error when |mean| >> std(x). */
  int32_t ln_centered;
  /* DZN_PREC_F16 (r5): the reduced-precision contraction of csrc/gemm_mx.hip — fp16 hi*hi plus the two cross terms in fp8 on the
   * block-scaled matrix instruction.  Wmx = planes from dzn_op_split_weights_mx ([rows][K/32][128 B]: fp16 hi | fp8 hi, fp8 lo of
   * w * 2^e_row), col_scale_mx[n] = 2^-e_row; needs a_amax like the fp16 two-term form.  Either NULL -> the single-term fp16
   * kernel on the leading plane of W2h (r2-r4's DZN_PREC_F16 arithmetic, kept for the positional conv and the ResNet trunk). */
  const void* Wmx;
  const float* col_scale_mx;
  /* number of entries of the a_amax / c_amax arrays (0 = not stated): only read by CHECKED builds (csrc/checked.h), which
   * assert that every scale-unit index stays inside them */
  int32_t amax_count;
  /* (r6) attention operands written by the q/k/v contraction itself (csrc/attention_planes.hip; reference arithmetic
   * W2V/components.py:453-486): the columns n >= kv_col0 of C = [q | k | v] x heads x 64 — the K and V slots — are NOT stored to C
   * as fp32; every (row, 64-column head slot) is scaled by the exact power of two that puts its own |max| into [2^14, 2^15) and
   * stored as the two fp16 terms of the f32h split: kv_planes = fp16 [2][rows][kv_ld] (plane p at p * kv_plane_stride elements,
   * row-major, kv_ld = N - kv_col0), kv_scale = f32 [rows][kv_ld / 64] = the INVERSE scale of the slot.  The attention kernel then
   * stages ready tiles (no split, no 7-fold re-split per query tile) and un-scales per key.  NULL = plain fp32 store.  Needs the
   * 16x16-block f32h / f32s contraction kernels, kv_col0 % 64 == 0 and N % 64 == 0 (refused otherwise). */
  void* kv_planes;
  int64_t kv_plane_stride;
  float* kv_scale;
  int32_t kv_ld;
  int32_t kv_col0;
} dzn_gemm_desc;

/* (r4) One BasicBlock of the 32-channel ResNet stage in one kernel (csrc/resblock_fused.hip):
 * out = relu(conv2(relu(conv1(in) + b1)) + b2 + in) over zero-bordered NHWC fp32 images [B][Hs+2][Ws+2][32]; W1 / W2 = fp16
 * two-term planes of the folded [32][288] weights (dzn_op_split_weights_h2), cs = their inverse row scales, amax_in f32 [B]
 * = per-image max |in|, l1max1 = max_oc sum_k |W1[oc][k]|, bmax1 = max_oc |b1[oc]| (bound of the intermediate). */
int dzn_op_resblock32_fused(const float* in, float* out, const void* W1, const float* cs1, const float* b1, const void* W2,
                            const float* cs2, const float* b2, const float* amax_in, float l1max1, float bmax1, int32_t B,
                            int32_t Hs, int32_t Ws, void* stream);

/* (r4) The same block with producer / consumer wavefronts (csrc/resblock_ws.hip), C = 32 or 64 planes: images
 * [B][Hs+2][Ws+2][C], W = planes of the folded [C][9 C] weights with k = (dh*3 + dw) * C + ci. */
int dzn_op_resblock_ws(const float* in, float* out, const void* W1, const float* cs1, const float* b1, const void* W2,
                       const float* cs2, const float* b2, const float* amax_in, float l1max1, float bmax1, int32_t B,
                       int32_t Hs, int32_t Ws, int32_t C, void* stream);

/* test knob: terms per operand the two entry points above run with — 2 (default, DZN_PREC_F32_H2) or 1 (the DZN_PREC_F16 form:
 * leading fp16 term only, plane 0 of the same weight buffers) */
int dzn_op_set_resblock_np(int32_t np);
/* (r5) test switch of the attention kernels (csrc/attention_split.hip): 1 = the wavefronts of a partial last query tile that own no
 * query compute scores / softmax / P.V as r2-r4 did; the stored bits are the same. */
int dzn_op_set_attention_noskip(int32_t on);

int dzn_op_gemm(const dzn_gemm_desc* d, void* stream);

/* Exact 3-way bf16 split of fp32 weights for DZN_PREC_F32_SPLIT (csrc/gemm_split.hip):
 * W fp32 [rows][K] (row stride ldw, K % 32 == 0) -> W3 bf16 [rows][K/32][3][32] (3*rows*K u16). */
int dzn_op_split_weights(const float* W, int64_t rows, int32_t K, int64_t ldw, void* W3, void* stream);

/* fp16 two-term split of fp32 weights for DZN_PREC_F32_H2: W2h fp16 [rows][K/32][2][32] (2*rows*K u16) of
 * W * 2^e_row with max|row| in [2^14, 2^15), col_scale f32 [rows] = 2^-e_row. */
int dzn_op_split_weights_h2(const float* W, int64_t rows, int32_t K, int64_t ldw, void* W2h, float* col_scale,
                            void* stream);

/* planes of the reduced-precision contraction (DZN_PREC_F16, csrc/gemm_mx.hip): Wmx = 128 bytes per (row, 32 k) = 4 * rows * K
 * bytes — fp16 of w * 2^e_row (64 B, fragment order) then four 16-B slots [fp8 e4m3 of w * 2^e_row x 8 | fp8 of the fp16
 * remainder * 2^11 x 8]; max|row| * 2^e_row in [2^7, 2^8); col_scale f32 [rows] = 2^-e_row. */
int dzn_op_split_weights_mx(const float* W, int64_t rows, int32_t K, int64_t ldw, void* Wmx, float* col_scale,
                            void* stream);
/* CHECKED builds only (python -m diarizen_amd.build --checked, csrc/checked.h): device-side bounds assertions in the hand-
 * scheduled kernels (LDS stage / fragment offsets, tracker indices, tile ranges).  out4 (may be NULL) = {failed checks since the
 * last reset, id of the first, its workgroup, its detail value}; reset != 0 clears the counters.  Returns the number of failed
 * checks (0 = clean), DZN_E_STATE in a release build. */
int dzn_checked_status(uint32_t* out4, int32_t reset);

/* tests / tuning: force one tile shape of csrc/gemm_mx.hip ("128x128", "128x64"; "auto" / NULL = the shape rule) */
int dzn_op_set_gemm_mx_cfg(const char* cfg);

/* tuning knob (scripts/bench_gemm_h2.py): force one tile configuration of csrc/gemm_split.hip for the calls that
 * follow ("128x128", "256x128s3", ...; "auto" / NULL = the shape heuristic).  Same effect as the DZN_GEMM_CFG
 * environment variable, which is read once per process. */
int dzn_op_set_gemm_cfg(const char* cfg);

/* amax[0] = max(amax[0], max |x[0..n)|) — the |max| tracker of a tensor whose producer has no fused tracker */
int dzn_op_amax(const float* x, int64_t n, float* amax, void* stream);

/* fp32 rows [rows, D] (D % 32 == 0) -> three bf16 planes [rows, D], plane_stride elements apart, channels of
 * every 32-block in fragment order: the pre-split A operand (dzn_gemm_desc.a_split3) */
int dzn_op_split_rows(const float* x, void* planes, int64_t plane_stride, int64_t rows, int32_t D, void* stream);

/* 3x3 stride-1 convolution 32 -> 32 channels over zero-bordered fp32 NHWC images [B][Hs+2][Ws+2][32]
 * (first ResNet34 stage, wespeaker/resnet.py:139-144 with BatchNorm folded into W / bias) in the fp32-split
 * arithmetic: W3 = dzn_op_split_weights of W [32][(dh*3+dw)*32 + ci]; out = post_relu?max(0,·):(·) of
 * (relu?max(0,·):(·))(conv + bias) + R.  Only interior pixels are written (borders stay zero). */
int dzn_op_conv3x3_c32(const float* in, const void* W3, const float* bias, const float* R, float* out,
                       int32_t B, int32_t Hs, int32_t Ws, int32_t relu, int32_t post_relu, void* stream);

/* the same convolution in the fp16 two-term arithmetic (DZN_PREC_F32_H2): W2h / col_scale = dzn_op_split_weights_h2 of
 * W [32][288], amax_in = f32 [B] per-image |max| of `in` */
int dzn_op_conv3x3_c32_h2(const float* in, const void* W3, const void* W2h, const float* col_scale, const float* amax_in,
                          const float* bias, const float* R, float* out, int32_t B, int32_t Hs, int32_t Ws,
                          int32_t relu, int32_t post_relu, void* stream);

/* y[r,:] = LayerNorm(x[r,:C]) * gamma + beta (eps), optional fused erf-GELU; row strides
 * ldx / ldy; columns [C, Cpad) of y are written as zero.  torch F.layer_norm. */
int dzn_op_layernorm(const float* x, int64_t ldx, float* y, int64_t ldy, const float* gamma,
                     const float* beta, int64_t rows, int32_t C, int32_t Cpad, float eps,
                     int32_t gelu, void* stream);

/* stats[r] = (mean, 1/sqrt(var + eps)) of x[r, :C] (biased variance, two passes over registers): the row
 * statistics of torch F.layer_norm for a LayerNorm that is folded into the consuming contraction. */
int dzn_op_row_stats(const float* x, int64_t ldx, int64_t rows, int32_t C, float eps, float* stats, void* stream);

/* gate_a_1[row, H] of WavLM's gated relative position bias (W2V/components.py:702-710):
 * y f32 [rows, ldy] (attention input, Htot*64 wide), Wg [8,64], bg [8], cst [Htot]. */
int dzn_op_gate(const float* y, int64_t ldy, const float* Wg, const float* bg, const float* cst,
                float* gate, int64_t rows, int32_t Htot, void* stream);

/*
 * Fused multi-head attention with optional WavLM gated relative-position bias
 * (W2V/components.py:453-486, 690-725; conformer.py:47-71).
 *   qkv   f32 [B*L, ldqkv]: q at col 0, k at col h*64, v at col 2*h*64 (head-major, 64 each)
 *   out   f32 [B*L, ldo]  : head j at cols j*64..
 *   gate  f32 [B*L, Htot] or NULL: gate_a_1 per (row, ORIGINAL head)
 *   table f32 [Htot, 2L-1] or NULL: rel-pos bias by (key - query + L - 1)
 *   head_idx i32 [h] device: original head index of kept head j
 */
/* pre-norm fusion of the two above: one pass over the raw residual rows x [rows, Htot*64] writes the LayerNorm
 * statistics stats [rows][2] (for a contraction with folded gamma / beta) and gate[rows, Htot] evaluated on
 * LayerNorm(x) * gamma + beta. */
int dzn_op_gate_stats(const float* x, int64_t ldx, const float* gamma, const float* beta, const float* Wg,
                      const float* bg, const float* cst, float* gate, float* stats, int64_t rows, int32_t Htot,
                      float eps, void* stream);

/* the same attention in the fp16 two-term arithmetic (DZN_PREC_F32_H2): amax = f32 [B], |max| of each window's qkv */
int dzn_op_attention_h2(const float* qkv, float* out, const float* gate, const float* table, const int32_t* head_idx,
                        int32_t B, int32_t L, int32_t h, int32_t Htot, int32_t ldqkv, int32_t ldo, float scale,
                        const float* amax, void* stream);

/* (r6) the attention on PRE-SPLIT K / V (csrc/attention_planes.hip): packs the K and V slots of a plain fp32 qkv = [B L][3 h 64]
 * into fp16 two-term planes with one power-of-two scale per (row, head slot) — exactly what the q/k/v contraction's epilogue
 * writes (dzn_gemm_desc.kv_planes) — and runs the kernel the engine runs.  planes / kvs: caller's scratch, fp16 [2][B L + 64][2 h 64]
 * and f32 [B L + 64][2 h], zero-filled by the caller. */
int dzn_op_attention_planes(const float* qkv, float* out, const float* gate, const float* table, const int32_t* head_idx,
                            int32_t B, int32_t L, int32_t h, int32_t Htot, int32_t ldqkv, int32_t ldo, float scale,
                            const float* amax, void* planes, float* kvs, void* stream);

/* test / A-B switch of the planes kernel: 16-query blocks per wavefront — 1 (default: 64 queries per workgroup, three workgroups
 * per CU) or 2 (128 queries per workgroup: every staged tile and LDS fragment read serves twice the matrix work; measured 4 % SLOWER,
 * profiles/r6_attention_planes_ab.txt) */
int dzn_op_set_attention_qb(int32_t query_blocks_per_wavefront);
/* ... and whether the next K / V tile is requested before the current one is computed (default 1) */
int dzn_op_set_attention_prefetch(int32_t on);

int dzn_op_attention(const float* qkv, float* out, const float* gate, const float* table,
                     const int32_t* head_idx, int32_t B, int32_t L, int32_t h,
                     int32_t Htot, int32_t ldqkv, int32_t ldo, float scale,
                     int32_t precision, void* stream);

/*
 * In-situ kernel timing (HIP events on the launch stream).  Enable, run forwards, then collect:
 * one entry per kernel class with the number of launches, the summed event time and the summed
 * ALGORITHMIC flops / HBM bytes declared by the launch sites (un-padded reference shapes).
 */
typedef struct dzn_prof_entry {
  char name[64];
  int64_t launches;
  double ms;
  double flops;
  double bytes;
} dzn_prof_entry;
int dzn_profile_enable(int32_t on);
int dzn_profile_collect(dzn_prof_entry* out, int32_t cap, int32_t* n);
/* pre-create n_events HIP events (two per launch at most) so that a timed region never allocates one */
int dzn_profile_reserve(int32_t n_events);

/* host-side relative-position bucket of WavLM (W2V/components.py:629-666), exposed for tests */
int dzn_op_relpos_bucket(int32_t rel, int32_t num_buckets, int32_t max_distance);

#ifdef __cplusplus
}
#endif
#endif
