// frontend_fused.hip — conv0 -> conv1 of the WavLM feature extractor in ONE kernel (DZN_PREC_F32_H2):
//
//     waveform --[window LN]--> conv0 (k 10, s 5) --> LayerNorm over C0 --> erf-GELU  ==(LDS)==>  conv1 (k 3, s 2)
//     W2V/model.py:113, W2V/components.py:119-122, :63-70, :182-209
//
// Unfused, conv0's [T0 = 25 599, 512] fp32 activations are written to HBM (52.4 MB per 8 s window, 13.2 GB per
// batch of 256) and read straight back as the A operand of conv1's contraction.  Here a workgroup owns 128 conv1
// output frames of one window: it recomputes the 257 conv0 frames they touch, 64 channels at a time, as TWO fp16 planes
// (the two-term split of gemm_split.hip, scaled by the exact power of two of conv0's static bound) in LDS, and feeds
// conv1's MFMAs from there.  Only the waveform (0.5 MB / window) is read and conv1's raw output ([T1, Cp1] fp32,
// 8.2 MB / window) is written.
//
//   * LayerNorm statistics of a conv0 frame come from its 10 input samples: mean_c y = wbar . x, var_c y = x^T Q x
//     (wbar / Q = mean / covariance over channels of the taps, built in double at load) -> no pass over the 512
//     outputs, so a 64-channel slab can be normalised on its own.
//   * K order of conv1 is (tap j, channel c) -> k = j * C0 + c, natural order inside every 32-block (the planes are
//     written by lanes that own one channel each); the fp16 weight planes W2h [Cp1][K/32][2][32] are split in that
//     order by split_weights_h2_natural_kernel.  A fragments: frame 2 t + j, 16-byte slot (kb * 4 + lq) ^ ((frame >> 1)
//     & 7): the 16 lanes of a fragment read 8 distinct slots (rows two frames apart would all hit the same banks).
//   * phases per 64-channel slab: [VALU] 4 wavefronts x 65 frames x 64 lanes = conv0 + LN + GELU + split -> LDS;
//     [MFMA] 2 x 2 wavefronts, 64 x 80 outputs each, 6 k-steps (3 taps x 2 blocks), W fragments straight from L2 into
//     registers (the slab's W would not fit LDS next to the planes).  Two workgroups per CU (74 KB of LDS each), so one
//     multiplies while the other computes activations.
#include <cstdlib>

#include "checked.h"
#include "common.h"
#include "split.h"

DZN_CHECKED_TU(frontend_fused)

namespace {

constexpr int FF_BM = 128;                 // conv1 output frames per workgroup
constexpr int FF_FR = 2 * FF_BM + 1;       // conv0 frames they read (k 3, s 2)
constexpr int FF_ROW = 128;                // bytes per frame per plane: 64 channels fp16
constexpr int FF_PLANE = (FF_FR + 3) * FF_ROW;

struct FusedArgs {
  const float* wave;      // [B, N]
  const float* wstats;    // [B, 2] waveform (mean, rstd) or null
  const float* w0;        // [C0, k0]
  const float* gamma0;    // [C0]
  const float* beta0;     // [C0]
  const float* lnq;       // [10 + 100]
  const u16* W2h;         // conv1 fp16 planes, natural k order: [N1p][K/32][2][32], K = 3 * C0
  const float* col_scale; // [N1p]
  float* out;             // [B, T1, N1p] raw conv1 output
  int N, T0, T1, C0, N1p;
  float eps, a_scale, a_inv;   // power-of-two scale of the conv0 activations (from their static bound) and inverse
  const float* gamma1;         // conv1's channel LayerNorm (+ GELU) fused into the epilogue when non-null: out = GELU(LN(conv1))
  const float* beta1;
  float* amax1;                // [B] |max| tracker of that output (scale of conv2's fp16 split) or null
  int C1;                      // real conv1 channels (<= N1p)
  int abl;                     // timing probe only (DZN_CONV01_ABL): 1 = skip the VALU phase, 2 = skip the MFMA phase (wrong results)
};

// PP = true (round 3, "ping-pong"): ONE 512-thread workgroup per CU carries TWO tiles.  Wavefronts 0-3 (group 0) own
// tile 2 x, wavefronts 4-7 (group 1) tile 2 x + 1, each with its private planes / strip in LDS, and group 1 runs one
// barrier interval behind: while one group is in its VALU phase (conv0 + LayerNorm + GELU + split -> LDS) the other is in
// its MFMA phase (conv1 from LDS), on every SIMD, by construction.  Measured with the phases switched off one at a time
// (DZN_CONV01_ABL, scripts/probe_kernel_class.py, 374 windows): VALU alone 7.8 ms, MFMA alone 8.9 ms, both 13.9 ms with
// two independent 256-thread workgroups per CU (PP = false: their phases overlap only by chance).
// UF (r5): conv0 frames a wavefront carries through the VALU phase at once (independent 10-FMA -> LayerNorm -> erf chains that
// interleave); 4 = r2-r4.
template <bool PP, int UF = 4>
__global__ __launch_bounds__(PP ? 512 : 256, 2) void conv01_fused_kernel(const FusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  constexpr int GROUP_LDS = (2 * FF_PLANE + (int)sizeof(float) * (5 * (FF_FR - 1) + 10 + 6 + 112) + (int)sizeof(float2) * FF_FR + 15) / 16 * 16;
  const int grp = PP ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8) : 0;
  unsigned char* smem = smem_all + grp * GROUP_LDS;
  unsigned char* pl0 = smem;                                   // hi plane  [FF_FR][64 ch] fp16, slots swizzled
  unsigned char* pl1 = smem + FF_PLANE;                        // lo plane
  float* sx = reinterpret_cast<float*>(smem + 2 * FF_PLANE);   // normalised samples of the strip
  float2* sst = reinterpret_cast<float2*>(sx + (5 * (FF_FR - 1) + 10 + 6));   // (mean, rstd) per conv0 frame
  // (r5) the 110 LayerNorm-statistics coefficients live in LDS (behind sst): as kernel-argument scalars they needed more SGPRs
  // than a wavefront has (r2-r4: 153 spilled SGPRs, ~200 v_readlane reloads in the statistics loop; now 0 —
  // profiles/r5_conv01_resources.txt)
  float* slnq = reinterpret_cast<float*>(sst + FF_FR);

  const int tid = threadIdx.x & 255, lane = tid & 63;  // thread / wavefront index inside the group
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int b = blockIdx.y;
  const int ntile = (a.T1 + FF_BM - 1) / FF_BM;
  int tile = PP ? 2 * (int)blockIdx.x + grp : (int)blockIdx.x;
  const bool live = tile < ntile;                      // PP, odd tile count: the last group 1 re-does the last tile, stores nothing
  tile = live ? tile : ntile - 1;
  const int t1_0 = tile * FF_BM;                       // first conv1 frame of the tile
  const int f0 = 2 * t1_0;                             // first conv0 frame
  const int nfr = min(FF_FR, a.T0 - f0);               // conv0 frames that exist
  const int nsamp = 5 * (nfr - 1) + 10;
  const float wmean = a.wstats ? a.wstats[2 * b] : 0.f;
  const float wrstd = a.wstats ? a.wstats[2 * b + 1] : 1.f;
  const float* wp = a.wave + (int64_t)b * a.N + (int64_t)f0 * 5;
  for (int i = tid; i < nsamp; i += 256) sx[i] = (wp[i] - wmean) * wrstd;
  if (tid < 110) slnq[tid] = a.lnq[tid];
  __syncthreads();
  for (int f = tid; f < nfr; f += 256) {     // LayerNorm statistics from the 10 samples of the frame (see header)
    float xv[10];
#pragma unroll
    for (int t = 0; t < 10; ++t) xv[t] = sx[f * 5 + t];
    float mu = 0.f, var = 0.f;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      mu = fmaf(slnq[i], xv[i], mu);
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < 10; ++j) q = fmaf(slnq[10 + i * 10 + j], xv[j], q);
      var = fmaf(q, q, var);     // lnq rows are the factor F of Q = F^T F: a sum of squares
    }
    sst[f] = make_float2(mu, 1.0f / sqrtf(fmaxf(var, 0.f) + a.eps));
  }

  // MFMA roles: 2 x 2 wavefronts over the 128 x 160 tile
  const int wm = wave >> 1, wn = wave & 1;
  constexpr int MI = 4, NI = 5;
  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int KB = 3 * a.C0 / 32;            // 32-blocks of conv1's K
  const int nslab = a.C0 / 64;

  if (PP && grp) __syncthreads();     // group 1 runs one interval behind (both groups execute the same number of barriers)
  for (int slab = 0; slab < nslab; ++slab) {
    __syncthreads();   // statistics written (first slab) / previous slab's planes fully consumed
    // ---- VALU phase: conv0 + LayerNorm + GELU + two-term split of channel (slab * 64 + lane).  A wavefront takes
    // frames wave, wave + 4, ...; FOUR of them per iteration so that the long dependent chains (10 FMAs -> LayerNorm ->
    // erf) of different frames interleave — with one frame at a time the chain latency, not the VALU rate, set the pace.
    {
      const int ch = slab * 64 + lane;
      float w0[10];
#pragma unroll
      for (int t = 0; t < 10; ++t) w0[t] = a.w0[ch * 10 + t];
      const float g0 = a.gamma0[ch], b0 = a.beta0[ch];
      const int cslot = lane >> 3, cbyte = (lane & 7) * 2;
      for (int fb = wave; fb < (a.abl == 1 ? 0 : FF_FR); fb += 4 * UF) {
        float o[UF];
#pragma unroll
        for (int u = 0; u < UF; ++u) {
          const int f = fb + 4 * u;
          const int fc = f < nfr ? f : nfr - 1;            // clamped: results of frames past the strip are zeroed below
          float acc0 = 0.f;
#pragma unroll
          for (int t = 0; t < 10; ++t) acc0 = fmaf(sx[fc * 5 + t], w0[t], acc0);
          const float2 st = sst[fc];
          o[u] = (acc0 - st.x) * st.y * g0 + b0;
        }
#pragma unroll
        for (int u = 0; u < UF; ++u) o[u] = gelu_erf(o[u]);
#pragma unroll
        for (int u = 0; u < UF; ++u) {
          const int f = fb + 4 * u;
          if (f < FF_FR) {                                  // wave-uniform
            const float xs = (f < nfr ? o[u] : 0.f) * a.a_scale;
            const _Float16 hi = (_Float16)xs;
            const _Float16 lo = (_Float16)(xs - (float)hi);
            const int off = f * FF_ROW + ((cslot ^ ((f >> 1) & 7)) << 4) + cbyte;
            DZN_CHECK(off + 2 <= FF_PLANE, 0x701, off);                                          // activation written inside its plane
            *reinterpret_cast<_Float16*>(pl0 + off) = hi;
            *reinterpret_cast<_Float16*>(pl1 + off) = lo;
          }
        }
      }
    }
    __syncthreads();
    // ---- MFMA phase: 6 k-steps = 3 taps x 2 blocks of 32 channels; the W fragments of step ks + 1 are fetched
    // (L2 -> registers) while step ks multiplies ----
    auto load_w = [&](int ks, u32x4 (&wf)[NI][2]) {
      const int j = ks >> 1, kb = ks & 1;
      const int kblk = j * (a.C0 / 32) + slab * 2 + kb;
#pragma unroll
      for (int jn = 0; jn < NI; ++jn) {
        const int n = wn * 80 + jn * 16 + lr;
        const u16* wpn = a.W2h + ((int64_t)n * KB + kblk) * 64 + lq * 8;
        DZN_CHECK(n < a.N1p && kblk < KB, 0x703, kblk);                                              // weight fragment inside the planes
        wf[jn][0] = *reinterpret_cast<const u32x4*>(wpn);
        wf[jn][1] = *reinterpret_cast<const u32x4*>(wpn + 32);
      }
    };
    auto mma_step = [&](int ks, const u32x4 (&wf)[NI][2]) {
      const int j = ks >> 1, kb = ks & 1;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int f = 2 * (wm * 64 + i * 16 + lr) + j;
        const int off = f * FF_ROW + (((kb * 4 + lq) ^ ((f >> 1) & 7)) << 4);
        DZN_CHECK(f < FF_FR + 3 && off + 16 <= FF_PLANE, 0x702, f);                                  // conv1 fragment (frame 2 t + tap) inside the plane
        u32x4 af[2];
        af[0] = *reinterpret_cast<const u32x4*>(pl0 + off);
        af[1] = *reinterpret_cast<const u32x4*>(pl1 + off);
#pragma unroll
        for (int tt = 0; tt < 3; ++tt)
#pragma unroll
          for (int jn = 0; jn < NI; ++jn)
            acc[i][jn] = mfma_np<2>(wf[jn][SplitTerms<2>::A[tt]], af[SplitTerms<2>::B[tt]], acc[i][jn]);
      }
    };
    u32x4 wfa[NI][2], wfb[NI][2];
    if (a.abl == 2) continue;
    load_w(0, wfa);
#pragma unroll
    for (int ks = 0; ks < 6; ks += 2) {
      load_w(ks + 1, wfb);
      mma_step(ks, wfa);
      if (ks + 2 < 6) load_w(ks + 2, wfa);
      mma_step(ks + 1, wfb);
    }
  }

  if (PP && !grp) __syncthreads();    // matches group 1's offset barrier
  // ---- epilogue: lane (lr, lq) of block (i, jn) holds frame t1_0 + wm*64 + i*16 + lr, channels n0 .. n0 + 3 ----
  float* ob = a.out + (int64_t)b * a.T1 * a.N1p;
  // undo the exact power-of-two operand scales first
#pragma unroll
  for (int jn = 0; jn < NI; ++jn) {
    const float4 c4 = *reinterpret_cast<const float4*>(a.col_scale + wn * 80 + jn * 16 + lq * 4);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      acc[i][jn][0] *= a.a_inv * c4.x; acc[i][jn][1] *= a.a_inv * c4.y;
      acc[i][jn][2] *= a.a_inv * c4.z; acc[i][jn][3] *= a.a_inv * c4.w;
    }
  }
  if (a.gamma1) {
    // r3: conv1's channel LayerNorm + GELU (W2V/components.py:63-70, 119-122) finished HERE: the 160 channels of a frame
    // live in the two wavefronts wn = 0 / 1 of its row half, so the two row reductions (mean, then centred squares: the
    // same two-pass arithmetic as norm.hip) go lane group -> wavefront (xor shuffles) -> the sibling wavefront through
    // LDS.  The stand-alone pass it replaces re-read and re-wrote conv1's 3.1 GB per 374 windows.
    __syncthreads();                                   // the planes are dead: their LDS carries the row partials now
    float* red = reinterpret_cast<float*>(smem);       // [2 wn][128 rows]
    float mean[MI], rstd[MI];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        float s1 = 0.f;
#pragma unroll
        for (int jn = 0; jn < NI; ++jn)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool in = wn * 80 + jn * 16 + lq * 4 + e < a.C1;
            const float d = pass == 0 ? acc[i][jn][e] : acc[i][jn][e] - mean[i];
            s1 += in ? (pass == 0 ? d : d * d) : 0.f;
          }
        s1 += __shfl_xor(s1, 16, 64);
        s1 += __shfl_xor(s1, 32, 64);
        if (lq == 0) red[wn * 128 + wm * 64 + i * 16 + lr] = s1;
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int r = wm * 64 + i * 16 + lr;
        const float tot = red[r] + red[128 + r];        // fixed order: wn 0 then wn 1 on both wavefronts
        if (pass == 0) mean[i] = tot / (float)a.C1;
        else rstd[i] = 1.0f / sqrtf(tot / (float)a.C1 + a.eps);
      }
      __syncthreads();
    }
    float mx = 0.f;   // |max| over the rows that exist (rows past the strip hold finite junk from clamped frames)
#pragma unroll
    for (int jn = 0; jn < NI; ++jn) {
      const int n0 = wn * 80 + jn * 16 + lq * 4;
      float g[4], be[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        g[e] = n0 + e < a.C1 ? a.gamma1[n0 + e] : 0.f;
        be[e] = n0 + e < a.C1 ? a.beta1[n0 + e] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int t1 = t1_0 + wm * 64 + i * 16 + lr;
        if (!live || t1 >= a.T1) continue;
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = n0 + e < a.C1 ? gelu_erf((acc[i][jn][e] - mean[i]) * rstd[i] * g[e] + be[e]) : 0.f;
          mx = fmaxf(mx, fabsf(o[e]));
        }
        *reinterpret_cast<float4*>(ob + (int64_t)t1 * a.N1p + n0) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
    if (a.amax1) track_amax(a.amax1 + b, mx);
    return;
  }
  if (!live) return;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int t1 = t1_0 + wm * 64 + i * 16 + lr;
    if (t1 >= a.T1) continue;
#pragma unroll
    for (int jn = 0; jn < NI; ++jn) {
      const int n0 = wn * 80 + jn * 16 + lq * 4;
      const f32x4 v = acc[i][jn];
      *reinterpret_cast<float4*>(ob + (int64_t)t1 * a.N1p + n0) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

// ---- r6: producer / consumer wavefronts, persistent workgroups (conv01_ws_kernel) ---------------------------------------
// conv01_fused_kernel above alternates its two phases per workgroup and relies on a second workgroup of the CU being in the
// other phase: measured alone the VALU phase takes 11.7 ms and the MFMA phase 12.5 ms per 561-window launch, together 21.3 —
// they overlap by chance, and BOTH are inefficient on their own (the VALU phase issues ~47 slots per activation, 14 of them LDS
// traffic with one lane per CHANNEL: 10 broadcast sample reads and two 2-byte plane writes per frame; the MFMA phase keeps the
// matrix pipe 36 % busy).  Here one 512-thread workgroup per CU walks over (window, tile) items, and in every step of a
// slab-granular pipeline
//   * wavefronts 0-3 (producers) compute conv0 + LayerNorm + GELU + split of slab s + 1 into one of two plane buffers with one
//     lane per FRAME: the frame's 10 samples, mean and 1/std stay in registers for the whole tile, the weights of two channels
//     at a time arrive as scalars (s_load: w0 is [C0][10], two channels are 20 consecutive floats), conv0's 10 products are
//     summed as (even taps, odd taps) on v_pk_fma_f32, the two channels share the LayerNorm / erf polynomial as one float2
//     (packed fp32: two results per lane and issue slot), and 8 channels leave as ONE ds_write_b128 per plane — no LDS reads,
//     ~30 issue slots per activation;
//   * wavefronts 4-7 (consumers) multiply slab s from the other buffer (same fragments and MFMA order as above) — on every
//     SIMD one VALU wavefront and one MFMA wavefront, by construction, with ONE s_barrier per step;
//   * a tile is 127 conv1 frames (255 conv0 frames + one spare = 256 lanes; row 127 of the 128-row MFMA tile is dead), and
//     conv1's LayerNorm + GELU epilogue needs one exchange between the two wavefronts that share a row: each leaves the mean and
//     the centred sum of squares of ITS 80 (73) channels before the step's barrier, both combine them after it (Chan et al.'s
//     pairwise update: as stable as the two-pass form, one exchange instead of two) and finish the tile at the start of the next
//     step, while the producers are already a slab into the next tile.
constexpr int WS_OUT = 127;                    // conv1 frames stored per tile
constexpr int WS_ROWS = 258;                   // plane rows: frames 0 .. 256 are read (row 256 only by the dead MFMA row), one slack
constexpr int WS_PLANE = WS_ROWS * FF_ROW;
constexpr int WS_LDS = 4 * WS_PLANE + (int)sizeof(float2) * 2 * 128 + (int)sizeof(float) * 112;

typedef float v2f __attribute__((ext_vector_type(2)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) float* cfloat_ptr;

// gelu_erf (common.h) on two values: the same operations per component, the multiply-adds through packed fp32 instructions
__device__ __forceinline__ v2f gelu_erf2(v2f v) {
  const v2f x = v * (v2f){0.70710678118654752440f, 0.70710678118654752440f};
  const v2f ax = {fabsf(x[0]), fabsf(x[1])};
  const v2f den = __builtin_elementwise_fma((v2f){0.3275911f, 0.3275911f}, ax, (v2f){1.0f, 1.0f});
  const v2f t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
  v2f p = __builtin_elementwise_fma((v2f){1.061405429f, 1.061405429f}, t, (v2f){-1.453152027f, -1.453152027f});
  p = __builtin_elementwise_fma(p, t, (v2f){1.421413741f, 1.421413741f});
  p = __builtin_elementwise_fma(p, t, (v2f){-0.284496736f, -0.284496736f});
  p = __builtin_elementwise_fma(p, t, (v2f){0.254829592f, 0.254829592f});
  const v2f q = (v2f){-1.4426950408889634f, -1.4426950408889634f} * ax * ax;
  const v2f e = {__builtin_amdgcn_exp2f(q[0]), __builtin_amdgcn_exp2f(q[1])};
  const v2f y = __builtin_elementwise_fma(-p * t, e, (v2f){1.0f, 1.0f});
  const v2f er = {copysignf(y[0], x[0]), copysignf(y[1], x[1])};
  return (v2f){0.5f, 0.5f} * v * ((v2f){1.0f, 1.0f} + er);
}

__global__ __launch_bounds__(512, 1) void conv01_ws_kernel(const FusedArgs a, const int B, const int ntile) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // [buffer][plane][WS_ROWS][128 B], then the row statistics of the epilogue, then the LayerNorm-statistics coefficients
  float2* red = reinterpret_cast<float2*>(smem + 4 * WS_PLANE);     // [2 wn][128 rows] (local mean, centred sum of squares)
  float* slnq = reinterpret_cast<float*>(red + 2 * 128);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave < 4;
  const int nitem = B * ntile;
  const int nslab = a.C0 / 64;
  // rows 256 / 257 of every plane are read by the dead MFMA row only, and never written: give them finite contents once
  if (tid < 4 * (FF_ROW / 4)) {
    const int pl = tid / (FF_ROW / 4), w = tid % (FF_ROW / 4);
    *reinterpret_cast<unsigned*>(smem + pl * WS_PLANE + 256 * FF_ROW + 4 * w) = 0u;
    *reinterpret_cast<unsigned*>(smem + pl * WS_PLANE + 257 * FF_ROW + 4 * w) = 0u;
  }
  if (tid < 110) slnq[tid] = a.lnq[tid];
  __syncthreads();
  int my_items = 0;
  for (int it = blockIdx.x; it < nitem; it += gridDim.x) ++my_items;
  const int nstep = my_items * nslab;

  if (producer) {
    // ================================ producers: lane -> conv0 frame f0 + 64 wave + lane ================================
    const int fl = wave * 64 + lane;                 // frame inside the tile = plane row
    const int swz = (fl >> 1) & 7;
    float x[10], xn[10];
    float mu = 0.f, rstd = 0.f, sc = 0.f;
    auto fetch = [&](int item, float (&dst)[10]) {   // raw samples of this lane's frame of `item` (clamped inside the window)
      const int b = item / ntile, tile = item - b * ntile;
      const int f = 2 * tile * WS_OUT + fl;
      const int fc = f < a.T0 ? f : a.T0 - 1;
      const float* wp = a.wave + (int64_t)b * a.N + (int64_t)fc * 5;
#pragma unroll
      for (int t = 0; t < 10; ++t) dst[t] = wp[t];
    };
    if (my_items > 0) fetch(blockIdx.x, xn);
    for (int g = 0; g <= nstep; ++g) {
      if (g < nstep) {
        const int k = g / nslab, slab = g - k * nslab;
        const int item = blockIdx.x + k * gridDim.x;
        if (slab == 0) {
          const int b = item / ntile, tile = item - b * ntile;
          const float wmean = a.wstats ? a.wstats[2 * b] : 0.f;
          const float wrstd = a.wstats ? a.wstats[2 * b + 1] : 1.f;
#pragma unroll
          for (int t = 0; t < 10; ++t) x[t] = (xn[t] - wmean) * wrstd;
          // LayerNorm statistics of the frame from its 10 samples (header of this file)
          float m = 0.f, var = 0.f;
#pragma unroll
          for (int i = 0; i < 10; ++i) {
            m = fmaf(slnq[i], x[i], m);
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < 10; ++j) q = fmaf(slnq[10 + i * 10 + j], x[j], q);
            var = fmaf(q, q, var);
          }
          mu = m;
          rstd = 1.0f / sqrtf(fmaxf(var, 0.f) + a.eps);
          sc = 2 * tile * WS_OUT + fl < a.T0 ? a.a_scale : 0.f;      // frames past the window are zero rows
        }
        if (slab == nslab - 1 && k + 1 < my_items) fetch(item + gridDim.x, xn);     // lands during this slab's arithmetic
        unsigned char* pl0 = smem + (g & 1) * 2 * WS_PLANE;
        unsigned char* pl1 = pl0 + WS_PLANE;
        const v2f xp[5] = {{x[0], x[1]}, {x[2], x[3]}, {x[4], x[5]}, {x[6], x[7]}, {x[8], x[9]}};
        const v2f mup = {mu, mu}, rsp = {rstd, rstd}, scp = {sc, sc};
        for (int cg = ((a.abl & 3) == 1 ? 8 : 0); cg < 8; ++cg) {        // 8 channels -> one 16-byte slot of the row, per plane
          u32x4 hw, lw;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int c = __builtin_amdgcn_readfirstlane(slab * 64 + cg * 8 + q * 2);
            // uniform addresses in the CONSTANT address space: the weights are read-only for the launch, and only loads the
            // compiler may treat as invariant become s_load (as plain global loads they were 20 broadcast vector loads per
            // 8 channels into 80 registers, waited for at the top of every iteration)
            const cfloat_ptr w = (cfloat_ptr)(uintptr_t)(a.w0 + c * 10);      // 20 scalars for the two channels
            const cfloat_ptr gp = (cfloat_ptr)(uintptr_t)(a.gamma0 + c), bp = (cfloat_ptr)(uintptr_t)(a.beta0 + c);
            v2f a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 5; ++t) {
              a0 = __builtin_elementwise_fma(xp[t], (v2f){w[2 * t], w[2 * t + 1]}, a0);
              a1 = __builtin_elementwise_fma(xp[t], (v2f){w[10 + 2 * t], w[10 + 2 * t + 1]}, a1);
            }
            v2f y;
            y[0] = a0[0] + a0[1];
            y[1] = a1[0] + a1[1];
            y = __builtin_elementwise_fma((y - mup) * rsp, (v2f){gp[0], gp[1]}, (v2f){bp[0], bp[1]});
            const v2f xs = gelu_erf2(y) * scp;
            const v2h h = {(_Float16)xs[0], (_Float16)xs[1]};
            const v2f r = xs - (v2f){(float)h[0], (float)h[1]};
            const v2h l = {(_Float16)r[0], (_Float16)r[1]};
            hw[q] = __builtin_bit_cast(unsigned, h);
            lw[q] = __builtin_bit_cast(unsigned, l);
          }
          const int off = fl * FF_ROW + ((cg ^ swz) << 4);
          DZN_CHECK(off + 16 <= 256 * FF_ROW && slab * 64 + cg * 8 + 7 < a.C0, 0x711, off);                            // 8 channels of a frame inside its plane rows
          *reinterpret_cast<u32x4*>(pl0 + off) = hw;
          *reinterpret_cast<u32x4*>(pl1 + off) = lw;
        }
      }
      __syncthreads();
    }
    return;
  }

  // ==================================== consumers: 2 x 2 wavefronts over the 128 x 160 tile ====================================
  const int cw = wave - 4;
  const int wm = cw >> 1, wn = cw & 1;
  const int lr = lane & 15, lq = lane >> 4;
  constexpr int MI = 4, NI = 5;
  f32x4 acc[MI][NI];
  const int nw = wn == 0 ? 80 : a.C1 - 80;           // channels of a row this wavefront holds (LayerNorm epilogue)
  bool pending = false;                              // the previous tile's epilogue waits for the sibling's row statistics
  int p_b = 0, p_t1 = 0;
  u32x4 wfa[NI][2], wfb[NI][2];                      // W fragments of two consecutive k-steps (wfa lives across the step barrier)
  // fragment-major copy of the planes (fragment_major_kernel, behind the natural-order copy): [kblk][n / 16][plane][16][32]
  const u16* wbase = a.W2h + (int64_t)2 * a.N1p * 3 * a.C0 + (wn * 5) * 1024 + lr * 32 + lq * 8;
  // finish the LayerNorm + GELU epilogue of the pending tile: the sibling's (mean, M2) were written before the last barrier
  auto finish = [&]() {
    float* ob = a.out + (int64_t)p_b * a.T1 * a.N1p;
    float mean[MI], rs[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int r = wm * 64 + i * 16 + lr;
      const float2 s0 = red[r], s1 = red[128 + r];
      const float n0 = 80.f, n1 = (float)(a.C1 - 80), n = (float)a.C1;
      const float dl = s1.x - s0.x;
      mean[i] = s0.x + dl * (n1 / n);
      const float m2 = s0.y + s1.y + dl * dl * (n0 * n1 / n);
      rs[i] = 1.0f / sqrtf(m2 / n + a.eps);
    }
    float mx = 0.f;
#pragma unroll
    for (int jn = 0; jn < NI; ++jn) {
      const int n0 = wn * 80 + jn * 16 + lq * 4;
      float gm[4], be[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        gm[e] = n0 + e < a.C1 ? a.gamma1[n0 + e] : 0.f;
        be[e] = n0 + e < a.C1 ? a.beta1[n0 + e] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int rr = wm * 64 + i * 16 + lr;
        const int t1 = p_t1 + rr;
        if (rr >= WS_OUT || t1 >= a.T1) continue;
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = n0 + e < a.C1 ? gelu_erf((acc[i][jn][e] - mean[i]) * rs[i] * gm[e] + be[e]) : 0.f;
          mx = fmaxf(mx, fabsf(o[e]));
        }
        *reinterpret_cast<float4*>(ob + (int64_t)t1 * a.N1p + n0) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
    if (a.amax1) track_amax(a.amax1 + p_b, mx);
    pending = false;
  };

  for (int g = 0; g <= nstep; ++g) {
    if (g >= 1) {
      const int gc = g - 1;
      const int k = gc / nslab, slab = gc - k * nslab;
      const int item = blockIdx.x + k * gridDim.x;
      const int b = item / ntile, tile = item - b * ntile;
      if (slab == 0) {
        if (pending) finish();
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      const unsigned char* pl0 = smem + (gc & 1) * 2 * WS_PLANE;
      const unsigned char* pl1 = pl0 + WS_PLANE;
      // One MFMA wavefront per SIMD: nothing else fills the matrix pipe while this one waits, so the order is pinned
      // (sched_barrier: hipcc otherwise sinks the W loads of step ks + 1 into the middle of step ks and chains dependent
      // accumulators back to back).  W of step ks + 1 (L2 -> registers) is requested BEFORE step ks multiplies, W of the next
      // slab's first step before this step's barrier (wfa travels across it); the A fragments of row block i + 1 are read
      // from LDS while block i multiplies; within a block the 15 MFMAs go term-major (5 independent accumulators apart).
      auto load_w = [&](int sl, int ks, u32x4 (&wf)[NI][2]) {
        const int j = ks >> 1, kb = ks & 1;
        const int kblk = j * (a.C0 / 32) + sl * 2 + kb;
#pragma unroll
        for (int jn = 0; jn < NI; ++jn) {
          const u16* wpn = wbase + (int64_t)kblk * (a.N1p / 16) * 1024 + jn * 1024;
          DZN_CHECK(kblk < 3 * a.C0 / 32 && wn * 5 + jn < a.N1p / 16, 0x713, kblk);                      // weight fragment inside the fragment-major planes
          wf[jn][0] = *reinterpret_cast<const u32x4*>(wpn);
          wf[jn][1] = *reinterpret_cast<const u32x4*>(wpn + 512);
        }
      };
      auto load_a = [&](int ks, int i, u32x4 (&af)[2]) {
        const int j = ks >> 1, kb = ks & 1;
        const int f = 2 * (wm * 64 + i * 16 + lr) + j;
        const int off = f * FF_ROW + (((kb * 4 + lq) ^ ((f >> 1) & 7)) << 4);
        DZN_CHECK(f <= 256 && off + 16 <= WS_PLANE, 0x712, f);                                          // conv1 fragment (frame 2 t + tap) inside the plane
        af[0] = *reinterpret_cast<const u32x4*>(pl0 + off);
        af[1] = *reinterpret_cast<const u32x4*>(pl1 + off);
      };
      auto mma_block = [&](int i, const u32x4 (&wf)[NI][2], const u32x4 (&af)[2]) {
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {
#pragma unroll
          for (int jn = 0; jn < NI; ++jn)
            acc[i][jn] = mfma_np<2>(wf[jn][SplitTerms<2>::A[tt]], af[SplitTerms<2>::B[tt]], acc[i][jn]);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      if ((a.abl & 3) != 2) {
        u32x4 afa[2], afb[2];
        if (gc == 0) load_w(slab, 0, wfa);          // (later steps: requested before the previous barrier)
        load_a(0, 0, afa);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
          u32x4 (&wc)[NI][2] = (ks & 1) ? wfb : wfa;
          u32x4 (&wx)[NI][2] = (ks & 1) ? wfa : wfb;
          if (ks + 1 < 6 && !(a.abl & 4)) load_w(slab, ks + 1, wx);      // (abl 4 / 8: timing probes without the W / A traffic)
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            u32x4 (&ac)[2] = (i & 1) ? afb : afa;
            u32x4 (&ax)[2] = (i & 1) ? afa : afb;
            if (!(a.abl & 8)) {
              if (i + 1 < MI) load_a(ks, i + 1, ax);
              else if (ks + 1 < 6) load_a(ks + 1, 0, ax);
            }
            __builtin_amdgcn_sched_barrier(0);
            mma_block(i, wc, ac);
          }
        }
        // 6 steps, MI = 4 blocks each: the buffers alternate evenly, so the next slab starts on wfa / afa again
        if (g < nstep && !(a.abl & 4)) load_w(slab + 1 < nslab ? slab + 1 : 0, 0, wfa);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (slab == nslab - 1) {
        // ---- the tile is complete: undo the exact power-of-two operand scales ----
#pragma unroll
        for (int jn = 0; jn < NI; ++jn) {
          const float4 c4 = *reinterpret_cast<const float4*>(a.col_scale + wn * 80 + jn * 16 + lq * 4);
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            acc[i][jn][0] *= a.a_inv * c4.x; acc[i][jn][1] *= a.a_inv * c4.y;
            acc[i][jn][2] *= a.a_inv * c4.z; acc[i][jn][3] *= a.a_inv * c4.w;
          }
        }
        const int t1_0 = tile * WS_OUT;
        if (a.gamma1) {
          // local statistics of this wavefront's channels of every row: mean, then centred sum of squares (two passes over
          // registers), left for the sibling; the tile is finished after the step's barrier (finish())
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            float s1 = 0.f;
#pragma unroll
            for (int jn = 0; jn < NI; ++jn)
#pragma unroll
              for (int e = 0; e < 4; ++e) s1 += wn * 80 + jn * 16 + lq * 4 + e < a.C1 ? acc[i][jn][e] : 0.f;
            s1 += __shfl_xor(s1, 16, 64);
            s1 += __shfl_xor(s1, 32, 64);
            const float ml = s1 / (float)nw;
            float s2 = 0.f;
#pragma unroll
            for (int jn = 0; jn < NI; ++jn)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float d = acc[i][jn][e] - ml;
                s2 += wn * 80 + jn * 16 + lq * 4 + e < a.C1 ? d * d : 0.f;
              }
            s2 += __shfl_xor(s2, 16, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (lq == 0) red[wn * 128 + wm * 64 + i * 16 + lr] = make_float2(ml, s2);
          }
          pending = true;
          p_b = b;
          p_t1 = t1_0;
        } else {
          float* ob = a.out + (int64_t)b * a.T1 * a.N1p;
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            const int rr = wm * 64 + i * 16 + lr;
            const int t1 = t1_0 + rr;
            if (rr >= WS_OUT || t1 >= a.T1) continue;
#pragma unroll
            for (int jn = 0; jn < NI; ++jn) {
              const int n0 = wn * 80 + jn * 16 + lq * 4;
              const f32x4 v = acc[i][jn];
              *reinterpret_cast<float4*>(ob + (int64_t)t1 * a.N1p + n0) = make_float4(v[0], v[1], v[2], v[3]);
            }
          }
        }
      }
    }
    __syncthreads();
  }
  if (pending) finish();
}

// W [rows][K] fp32 -> fp16 planes [rows][K/32][2][32] in NATURAL k order (the fused kernel writes its A planes channel
// by channel), w * 2^e_row with max |row| in [2^14, 2^15); col_scale[row] = 2^-e_row
__global__ __launch_bounds__(256) void split_weights_h2_natural_kernel(const float* __restrict__ W, int64_t rows, int K,
                                                                       u16* __restrict__ W2, float* __restrict__ col_scale) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float m = 0.f;
  for (int k = lane; k < K; k += 64) m = fmaxf(m, fabsf(W[r * K + k]));
  m = wave_max(m);
  float sc, inv;
  h2_scale(m, sc, inv);
  if (lane == 0) col_scale[r] = inv;
  for (int k = lane; k < K; k += 64) {
    const float x = W[r * K + k] * sc;
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)(x - (float)h);
    u16* o = W2 + r * 2 * K + (int64_t)(k >> 5) * 64 + (k & 31);
    o[0] = __builtin_bit_cast(u16, h);
    o[32] = __builtin_bit_cast(u16, l);
  }
}

// conv1's planes for conv01_ws_kernel: [K/32][N/16][2 planes][16 rows][32] — the 64 lanes of one fragment load
// (row lr, 16-byte piece lq) read ONE contiguous KB, a wavefront's whole k-step 10 contiguous KB.  In the [row][K/32][2][32]
// order above a fragment load touches sixteen 64-byte pieces 6 KB apart, and the consumers' W stream (40 KB per k-step and CU
// through the vector L1) then set the pace: 15.0 ms per 561-window launch for the consumers alone against 11.3 ms with this
// order and 8.3 ms without any W traffic (profiles/r6_conv01_ws_probe.txt).
__global__ __launch_bounds__(256) void fragment_major_kernel(const u16* __restrict__ W2, int rows, int KB, u16* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;      // one 16-byte piece each
  const int64_t total = (int64_t)rows * KB * 2 * 4;
  if (idx >= total) return;
  const int lq = (int)(idx & 3);
  const int lr = (int)((idx >> 2) & 15);
  const int pl = (int)((idx >> 6) & 1);
  const int64_t rest = idx >> 7;
  const int nb = (int)(rest % (rows / 16)), kblk = (int)(rest / (rows / 16));
  const int n = nb * 16 + lr;
  const u32x4 v = *reinterpret_cast<const u32x4*>(W2 + ((int64_t)n * KB + kblk) * 64 + pl * 32 + lq * 8);
  *reinterpret_cast<u32x4*>(out + idx * 8) = v;
}

}  // namespace

// second copy of conv1's planes in fragment-major order (rows % 16 == 0), written behind the first: out = W2 + 2 rows K
int launch_fragment_major(const void* W2, int rows, int K, void* out, hipStream_t s) {
  if (rows <= 0 || (rows & 15) || K <= 0 || (K & 31)) return DZN_E_INVALID;
  const int64_t total = (int64_t)rows * (K / 32) * 8;
  hipLaunchKernelGGL(fragment_major_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, s, static_cast<const u16*>(W2), rows, K / 32,
                     static_cast<u16*>(out));
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}

int launch_split_weights_h2_natural(const float* W, int64_t rows, int K, void* W2, float* col_scale, hipStream_t s) {
  if (rows <= 0) return DZN_OK;
  if (K <= 0 || (K & 31)) return DZN_E_INVALID;
  hipLaunchKernelGGL(split_weights_h2_natural_kernel, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, s, W, rows, K,
                     static_cast<u16*>(W2), col_scale);
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}

// conv0 (k 10, s 5, C0 % 64 == 0) + LN + GELU + conv1 (k 3, s 2, 160 padded outputs): out = raw conv1 [B, T1, 160]
int launch_conv01_fused(const float* wave, int B, int N, const float* wstats, const float* w0, const float* gamma0,
                        const float* beta0, const float* lnq, int C0, int T0, int T1, const void* W2h,
                        const float* col_scale, int N1p, float act_bound, float eps, float* out, hipStream_t st,
                        const float* gamma1, const float* beta1, int C1, float* amax1) {
  if (B <= 0 || T1 <= 0) return DZN_OK;
  if (gamma1 && (!beta1 || C1 <= 0 || C1 > N1p)) return DZN_E_INVALID;
  if ((C0 & 63) || N1p != 160 || !lnq || !W2h || !col_scale || !(act_bound > 0.f)) return DZN_E_INVALID;
  FusedArgs a{};
  a.wave = wave; a.wstats = wstats; a.w0 = w0; a.gamma0 = gamma0; a.beta0 = beta0; a.lnq = lnq;
  a.W2h = static_cast<const u16*>(W2h); a.col_scale = col_scale; a.out = out;
  a.N = N; a.T0 = T0; a.T1 = T1; a.C0 = C0; a.N1p = N1p; a.eps = eps;
  a.gamma1 = gamma1; a.beta1 = beta1; a.C1 = C1; a.amax1 = amax1;
  static const int abl = getenv("DZN_CONV01_ABL") ? atoi(getenv("DZN_CONV01_ABL")) : 0;
  a.abl = abl;
  {   // exact power-of-two scale that puts the bound into [2^14, 2^15)
    int e;
    (void)frexpf(act_bound, &e);          // act_bound = m * 2^e, m in [0.5, 1)
    a.a_scale = ldexpf(1.0f, 15 - e);
    a.a_inv = ldexpf(1.0f, e - 15);
  }
  const size_t group_lds = (2 * FF_PLANE + sizeof(float) * (5 * (FF_FR - 1) + 10 + 6 + 112) + sizeof(float2) * FF_FR + 15) / 16 * 16;
  // two tiles per 512-thread workgroup with the phases in anti-phase: measured SLOWER (15.7 vs 13.7 ms per 374 windows,
  // profiles/r3_conv01_phase_probe.txt) — a phase that runs on ONE wavefront per SIMD is bound by its dependent chains
  // (VALU alone: 13.4 ms in this form, 7.8 ms when two workgroups share the SIMDs), so forcing the overlap costs more
  // than it hides.  Kept behind DZN_CONV01_PP=1 for the record; the default is two independent workgroups per CU.
  static const bool pp = getenv("DZN_CONV01_PP") != nullptr;
  // frames per wavefront iteration of the VALU phase: 8 since r5 (22.03 -> 21.65 ms per 561-window launch; 4 = r2-r4's kernel;
  // profiles/r5_conv01_probe.txt, which also re-measures the anti-phase form: 23.8 / 22.7 ms at 4 / 8 — still slower)
  static const int uf = getenv("DZN_CONV01_UF") ? atoi(getenv("DZN_CONV01_UF")) : 8;
  static unsigned long long attr_mask = 0;
  if (first_use_on_device(attr_mask)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv01_fused_kernel<false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv01_fused_kernel<true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv01_fused_kernel<false, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv01_fused_kernel<true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const int ntile = (T1 + FF_BM - 1) / FF_BM;
  // algorithmic work: conv0 + conv1 flops; algorithmic HBM bytes: waveform in, conv1's raw output out
  const int pid = prof_begin(st, "conv01_fused", 2.0 * B * ((double)T0 * C0 * 10 + (double)T1 * 153.0 * 3 * C0),
                             B * (4.0 * N + 4.0 * (double)T1 * N1p));
  // (r6) producer / consumer wavefronts in persistent workgroups (conv01_ws_kernel); DZN_CONV01_WS=0: the phase-alternating kernel
  const char* ws_env = getenv("DZN_CONV01_WS");      // read per call (tests compare the two kernels in one process)
  const bool ws = !(ws_env && atoi(ws_env) == 0);
  if (ws && !pp && (!gamma1 || (C1 > 80 && C1 <= 160))) {
    static unsigned long long ws_mask = 0;
    static int cus[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (first_use_on_device(ws_mask)) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv01_ws_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      int n = 0;
      if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
      cus[dev & 63] = n;
    }
    const int wtile = (T1 + WS_OUT - 1) / WS_OUT;
    const int64_t items = (int64_t)B * wtile;
    const int grid = (int)(items < cus[dev & 63] ? items : cus[dev & 63]);
    hipLaunchKernelGGL(conv01_ws_kernel, dim3(grid), dim3(512), WS_LDS, st, a, B, wtile);
  } else if (pp && uf == 8) hipLaunchKernelGGL((conv01_fused_kernel<true, 8>), dim3((ntile + 1) / 2, B), dim3(512), 2 * group_lds, st, a);
  else if (pp) hipLaunchKernelGGL((conv01_fused_kernel<true, 4>), dim3((ntile + 1) / 2, B), dim3(512), 2 * group_lds, st, a);
  else if (uf == 8) hipLaunchKernelGGL((conv01_fused_kernel<false, 8>), dim3(ntile, B), dim3(256), group_lds, st, a);
  else hipLaunchKernelGGL((conv01_fused_kernel<false, 4>), dim3(ntile, B), dim3(256), group_lds, st, a);
  prof_end(pid, st);
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}
