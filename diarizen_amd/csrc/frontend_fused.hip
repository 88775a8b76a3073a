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

}  // namespace

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
  if (pp && uf == 8) hipLaunchKernelGGL((conv01_fused_kernel<true, 8>), dim3((ntile + 1) / 2, B), dim3(512), 2 * group_lds, st, a);
  else if (pp) hipLaunchKernelGGL((conv01_fused_kernel<true, 4>), dim3((ntile + 1) / 2, B), dim3(512), 2 * group_lds, st, a);
  else if (uf == 8) hipLaunchKernelGGL((conv01_fused_kernel<false, 8>), dim3(ntile, B), dim3(256), group_lds, st, a);
  else hipLaunchKernelGGL((conv01_fused_kernel<false, 4>), dim3(ntile, B), dim3(256), group_lds, st, a);
  prof_end(pid, st);
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}
