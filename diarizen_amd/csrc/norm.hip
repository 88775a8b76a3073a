// norm.hip — row LayerNorm (+ optional erf-GELU, + optional per-column post-scale), element-type
// cast, and per-window waveform statistics.
//
// LayerNorm rows: torch.nn.functional.layer_norm over the last dim (biased variance, eps inside
// the sqrt).  Sites: channel LN after conv1..6 of the WavLM extractor (W2V/components.py:63-70,
// 119-122, fused with GELU; the last one also applies FeatureExtractor.dummy_weight, :208),
// FeatureProjection.layer_norm (:305), EncoderLayer.layer_norm / final_layer_norm (:923-941),
// Model.lnorm (model_wavlm_conformer.py:257) and every Conformer ln_norm (conformer.py).
// HBM-bound: one wavefront per row, the row is held in registers between the two passes, so each
// element is read once and written once.  Input / output may be fp32 or (bf16 engine mode) bf16.
#include <cstdint>
#include <initializer_list>
#include <type_traits>

#include "common.h"

namespace {

template <int MAXI, typename TI, typename TO>
__global__ __launch_bounds__(256) void layernorm_kernel(const TI* __restrict__ x, int64_t ldx,
                                                        TO* __restrict__ y, int64_t ldy,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        const float* __restrict__ post, int64_t rows,
                                                        int C, int Cpad, float eps, int gelu,
                                                        float* __restrict__ amax_out, int64_t amax_unit, int rows_per_wave) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // every wavefront owns a contiguous run of rows (rows_per_wave = 1 without a tracker): the |max| tracker of a
  // unit (window) is then touched once per run instead of once per row
  const int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * rows_per_wave;
  float amax = 0.f;
  int64_t unit = amax_out && amax_unit > 0 && r0 < rows ? r0 / amax_unit : 0;
  for (int64_t row = r0; row < r0 + rows_per_wave && row < rows; ++row) {
    const TI* xp = x + row * ldx;
    float v[MAXI];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
      const int idx = lane + 64 * i;
      v[i] = idx < C ? ld_act(xp, idx) : 0.f;
      sum += v[i];
    }
    const float mean = wave_sum(sum) / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
      const int idx = lane + 64 * i;
      const float dv = idx < C ? v[i] - mean : 0.f;
      sq += dv * dv;
    }
    const float var = wave_sum(sq) / (float)C;
    const float rstd = 1.0f / sqrtf(var + eps);
    if (amax_out && amax_unit > 0 && row / amax_unit != unit) {   // wave-uniform: flush the finished unit
      track_amax(amax_out + unit, amax);
      amax = 0.f;
      unit = row / amax_unit;
    }
    TO* yp = y + row * ldy;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
      const int idx = lane + 64 * i;
      if (idx < C) {
        float o = (v[i] - mean) * rstd;
        if (gamma) o = o * gamma[idx] + beta[idx];
        if (gelu) o = gelu_erf(o);
        if (post) o *= post[idx];
        st_act(yp, idx, o);
        amax = fmaxf(amax, fabsf(o));
      } else if (idx < Cpad) {
        st_act(yp, idx, 0.f);
      }
    }
  }
  // |max| of the output rows of this unit (window): scale of the consumer's fp16 split
  if (amax_out && r0 < rows) track_amax(amax_out + unit, amax);
}

__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ amax_out) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
  track_amax(amax_out, m);
}

// (mean, rstd) of each row — the LayerNorm whose affine part is folded into the consuming contraction
// (dzn_gemm_desc.ln_stats): one read of x, no write of a normalised copy.
template <int MAXI>
__global__ __launch_bounds__(256) void row_stats_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows, int C,
                                                        float eps, float* __restrict__ stats) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const float* xp = x + row * ldx;
  float v[MAXI];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    const int idx = lane + 64 * i;
    v[i] = idx < C ? xp[idx] : 0.f;
    sum += v[i];
  }
  const float mean = wave_sum(sum) / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    const float dv = lane + 64 * i < C ? v[i] - mean : 0.f;
    sq += dv * dv;
  }
  const float var = wave_sum(sq) / (float)C;
  if (lane == 0) *reinterpret_cast<float2*>(stats + 2 * row) = make_float2(mean, 1.0f / sqrtf(var + eps));
}

// per-row partial (sum, sum of squares) left by a contraction epilogue (dzn_gemm_desc.stat_partial) -> (mean, rstd);
// the P partials of a row are added in index order in double: deterministic, and E[x^2] - mean^2 loses nothing to the
// fp32 rounding already in the partials
__global__ __launch_bounds__(256) void stats_finalize_kernel(const float2* __restrict__ part, int64_t rows, int P, int C,
                                                             float eps, float2* __restrict__ stats) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  double s = 0.0, q = 0.0;
  for (int p = 0; p < P; ++p) {
    const float2 v = part[r * P + p];
    s += (double)v.x;
    q += (double)v.y;
  }
  const double mean = s / C;
  double var = q / C - mean * mean;
  var = var > 0.0 ? var : 0.0;
  stats[r] = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)eps)));
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ x, u16* __restrict__ y,
                                                        int64_t n4) {
  const float4* xs = reinterpret_cast<const float4*>(x);
  ushort4* yd = reinterpret_cast<ushort4*>(y);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 v = xs[i];
    u16 t[4];
    st_act(t, 0, v.x); st_act(t, 1, v.y); st_act(t, 2, v.z); st_act(t, 3, v.w);
    yd[i] = make_ushort4(t[0], t[1], t[2], t[3]);
  }
}

// mean / rstd of each window's N samples (F.layer_norm(waveforms, waveforms.shape), W2V/model.py:113)
__global__ __launch_bounds__(1024) void wave_stats_kernel(const float* __restrict__ w, int N,
                                                          float eps, float* __restrict__ stats) {
  __shared__ float red[32];
  const int b = blockIdx.x;
  const float* p = w + (int64_t)b * N;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float s = 0.f;
  for (int i = tid; i < N; i += 1024) s += p[i];
  s = wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  float tot = 0.f;
  for (int i = 0; i < 16; ++i) tot += red[i];
  const float mean = tot / (float)N;
  __syncthreads();
  float q = 0.f;
  for (int i = tid; i < N; i += 1024) {
    const float d = p[i] - mean;
    q += d * d;
  }
  q = wave_sum(q);
  if (lane == 0) red[wave] = q;
  __syncthreads();
  if (tid == 0) {
    float t2 = 0.f;
    for (int i = 0; i < 16; ++i) t2 += red[i];
    stats[2 * b] = mean;
    stats[2 * b + 1] = 1.0f / sqrtf(t2 / (float)N + eps);
  }
}

// ---- float4 form (round 3), fp32 in / out ----
// One wavefront-per-row with 4-byte loads issues ~100 instructions for a 160-channel row (the conv LayerNorm sites): the
// kernel ran at 2.8 TB/s, instruction-bound.  Here G lanes (16 / 32 / 64) own a row, NV float4s each, so a wavefront
// processes 64 / G rows per step with NV 16-byte loads and stores per lane; the two reductions run over the G lanes of a
// row with xor shuffles.  Same two-pass arithmetic (mean, then centred squares) as the scalar form.
template <int G, int NV>
__global__ __launch_bounds__(256) void layernorm_v4_kernel(const float* x, int64_t ldx, float* y,   /* y may alias x (post-norm layers) */
                                                           int64_t ldy, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ post,
                                                           int64_t rows, int C, int Cpad, float eps, int gelu,
                                                           float* __restrict__ amax_out, int64_t amax_unit, int iters) {
  constexpr int RPW = 64 / G;                  // rows per wavefront step
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / G, gl = lane % G;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * iters * RPW + sub;
  // this lane's columns: 4 (gl + G k) .. + 3
  float4 g4[NV], b4[NV], p4[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = 4 * (gl + G * k);
    float gg[4], bb[4], pp[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool in = c + e < C;
      gg[e] = (gamma && in) ? gamma[c + e] : 1.f;
      bb[e] = (beta && gamma && in) ? beta[c + e] : 0.f;
      pp[e] = (post && in) ? post[c + e] : 1.f;
    }
    g4[k] = make_float4(gg[0], gg[1], gg[2], gg[3]);
    b4[k] = make_float4(bb[0], bb[1], bb[2], bb[3]);
    p4[k] = make_float4(pp[0], pp[1], pp[2], pp[3]);
  }
  const float invC = 1.0f / (float)C;
  float seen = -1.f;                           // last tracker value this lane group saw for `seen_unit`
  int64_t seen_unit = -1;
  for (int it = 0; it < iters; ++it) {
    const int64_t row = row0 + (int64_t)it * RPW;
    const bool valid = row < rows;
    const int64_t rc = valid ? row : rows - 1;
    const float* xp = x + rc * ldx;
    float4 v[NV];
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = 4 * (gl + G * k);
      v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c + 3 < C) {
        v[k] = *reinterpret_cast<const float4*>(xp + c);
      } else if (c < C) {                       // ragged last quad (C % 4 != 0)
        v[k].x = xp[c];
        if (c + 1 < C) v[k].y = xp[c + 1];
        if (c + 2 < C) v[k].z = xp[c + 2];
      }
      sum += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float mean = sum * invC;
    float sq = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = 4 * (gl + G * k);
      const float d0 = c < C ? v[k].x - mean : 0.f, d1 = c + 1 < C ? v[k].y - mean : 0.f;
      const float d2 = c + 2 < C ? v[k].z - mean : 0.f, d3 = c + 3 < C ? v[k].w - mean : 0.f;
      sq += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
    const float rstd = 1.0f / sqrtf(sq * invC + eps);
    float* yp = y + rc * ldy;
    float m = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = 4 * (gl + G * k);
      float in[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
      const float gk[4] = {g4[k].x, g4[k].y, g4[k].z, g4[k].w}, bk[4] = {b4[k].x, b4[k].y, b4[k].z, b4[k].w};
      const float pk[4] = {p4[k].x, p4[k].y, p4[k].z, p4[k].w};
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = (in[e] - mean) * rstd;
        if (gamma) t = t * gk[e] + bk[e];
        if (gelu) t = gelu_erf(t);
        if (post) t *= pk[e];
        o[e] = c + e < C ? t : 0.f;
        m = fmaxf(m, fabsf(o[e]));
      }
      if (valid && c < (Cpad > C ? Cpad : C)) *reinterpret_cast<float4*>(yp + c) = make_float4(o[0], o[1], o[2], o[3]);
    }
    if (amax_out) {     // |max| of the row -> tracker of its unit (window); probed only when this lane group's maximum grows
#pragma unroll
      for (int o = G / 2; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
      const int64_t unit = amax_unit > 0 ? rc / amax_unit : 0;
      if (unit != seen_unit) {
        seen_unit = unit;
        seen = -1.f;
      }
      if (valid && gl == 0 && m > seen) {
        track_amax_lane(amax_out + unit, m);
        seen = m;
      }
    }
  }
}

template <typename TI, typename TO>
int launch_ln_typed(const TI* x, int64_t ldx, TO* y, int64_t ldy, const float* g, const float* b,
                    const float* post, int64_t rows, int C, int Cpad, float eps, int gelu, hipStream_t s,
                    float* amax, int64_t amax_unit) {
  if constexpr (std::is_same<TI, float>::value && std::is_same<TO, float>::value) {
    const int needv = Cpad > C ? Cpad : C;
    if (!(needv & 3) && !(ldx & 3) && !(ldy & 3) && needv <= 1024 && !(((uintptr_t)x | (uintptr_t)y) & 15)) {
      // lanes per row / float4s per lane: least padding first, narrowest row group on ties
      int G = 0, NV = 0, best = 1 << 30;
      for (int gcand : {16, 32, 64}) {
        const int nv = (needv + 4 * gcand - 1) / (4 * gcand);
        if (nv <= 4 && nv * 4 * gcand < best) { best = nv * 4 * gcand; G = gcand; NV = nv; }
      }
      if (G) {
        const int rpwv = 64 / G;
        // ~8 wavefront steps per wavefront, at least 2048 workgroups when there are rows enough
        int64_t steps = cdiv64(rows, rpwv);
        int iters = (int)(steps / (2048 * 4));
        iters = iters < 1 ? 1 : (iters > 8 ? 8 : iters);
        const unsigned gridv = (unsigned)cdiv64(cdiv64(steps, iters), 4);
#define DZN_LN4(GV, NVV)                                                                                         \
  hipLaunchKernelGGL((layernorm_v4_kernel<GV, NVV>), dim3(gridv), dim3(256), 0, s, x, ldx, y, ldy, g, b, post, rows, \
                     C, Cpad, eps, gelu, amax, amax_unit, iters)
        if (G == 16 && NV == 1) DZN_LN4(16, 1);
        else if (G == 16 && NV == 2) DZN_LN4(16, 2);
        else if (G == 16 && NV == 3) DZN_LN4(16, 3);
        else if (G == 16 && NV == 4) DZN_LN4(16, 4);
        else if (G == 32 && NV == 3) DZN_LN4(32, 3);
        else if (G == 32 && NV == 4) DZN_LN4(32, 4);
        else if (G == 64 && NV == 3) DZN_LN4(64, 3);
        else if (G == 64 && NV == 4) DZN_LN4(64, 4);
        else G = 0;
#undef DZN_LN4
        if (G) return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
      }
    }
  }
  // with a tracker: contiguous runs of rows per wavefront, >= 8 waves per SIMD worth of wavefronts in flight
  int rpw = 1;
  if (amax) {
    const int64_t waves = 256 * 4 * 8;
    rpw = (int)(rows / waves);
    rpw = rpw < 1 ? 1 : (rpw > 64 ? 64 : rpw);
  }
  const unsigned grid = (unsigned)cdiv64(cdiv64(rows, rpw), 4);
  const int need = (Cpad > C ? Cpad : C);
#define DZN_LN(MAXI)                                                                               \
  hipLaunchKernelGGL((layernorm_kernel<MAXI, TI, TO>), dim3(grid), dim3(256), 0, s, x, ldx, y, ldy, g, \
                     b, post, rows, C, Cpad, eps, gelu, amax, amax_unit, rpw)
  if (need <= 256) DZN_LN(4);
  else if (need <= 512) DZN_LN(8);
  else if (need <= 1024) DZN_LN(16);
  else DZN_LN(32);
#undef DZN_LN
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}

}  // namespace

int launch_layernorm_t(const void* x, int x_bf16, int64_t ldx, void* y, int y_bf16, int64_t ldy,
                       const float* g, const float* b, const float* post, int64_t rows, int C, int Cpad,
                       float eps, int gelu, hipStream_t s, float* amax, int64_t amax_unit) {
  ProfScope prof_scope_(s, "layernorm", 0.0, (double)rows * C * 8.0);
  if (rows <= 0) return DZN_OK;
  if (C <= 0 || C > 2048 || Cpad > 2048) return DZN_E_INVALID;
  const float* xf = static_cast<const float*>(x);
  const u16* xh = static_cast<const u16*>(x);
  float* yf = static_cast<float*>(y);
  u16* yh = static_cast<u16*>(y);
  if (!x_bf16 && !y_bf16) return launch_ln_typed(xf, ldx, yf, ldy, g, b, post, rows, C, Cpad, eps, gelu, s, amax, amax_unit);
  if (!x_bf16 && y_bf16) return launch_ln_typed(xf, ldx, yh, ldy, g, b, post, rows, C, Cpad, eps, gelu, s, amax, amax_unit);
  if (x_bf16 && !y_bf16) return launch_ln_typed(xh, ldx, yf, ldy, g, b, post, rows, C, Cpad, eps, gelu, s, amax, amax_unit);
  return launch_ln_typed(xh, ldx, yh, ldy, g, b, post, rows, C, Cpad, eps, gelu, s, amax, amax_unit);
}

int launch_layernorm(const float* x, int64_t ldx, float* y, int64_t ldy, const float* g,
                     const float* b, int64_t rows, int C, int Cpad, float eps, int gelu,
                     hipStream_t s) {
  return launch_layernorm_t(x, 0, ldx, y, 0, ldy, g, b, nullptr, rows, C, Cpad, eps, gelu, s);
}

int launch_stats_finalize(const float* partial, int64_t rows, int P, int C, float eps, float* stats, hipStream_t s) {
  if (rows <= 0) return DZN_OK;
  if (P < 1 || C < 1) return DZN_E_INVALID;
  ProfScope prof_scope_(s, "stats_finalize", 0.0, (double)rows * (P + 1) * 8.0);
  hipLaunchKernelGGL(stats_finalize_kernel, dim3((unsigned)cdiv64(rows, 256)), dim3(256), 0, s,
                     reinterpret_cast<const float2*>(partial), rows, P, C, eps, reinterpret_cast<float2*>(stats));
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}

int launch_amax(const float* x, int64_t n, float* amax, hipStream_t s) {
  if (n <= 0) return DZN_OK;
  int64_t g = cdiv64(n, 256 * 8);
  g = g > 4096 ? 4096 : g;
  hipLaunchKernelGGL(amax_kernel, dim3((unsigned)g), dim3(256), 0, s, x, n, amax);
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}

extern "C" int dzn_op_amax(const float* x, int64_t n, float* amax, void* stream) {
  if (!x || !amax) return DZN_E_INVALID;
  return launch_amax(x, n, amax, reinterpret_cast<hipStream_t>(stream));
}

int launch_row_stats(const float* x, int64_t ldx, int64_t rows, int C, float eps, float* stats, hipStream_t s) {
  if (rows <= 0) return DZN_OK;
  if (C <= 0 || C > 2048) return DZN_E_INVALID;
  const unsigned grid = (unsigned)cdiv64(rows, 4);
  int pid = prof_enabled() ? prof_begin(s, "row_stats", 0.0, (double)rows * C * 4.0) : -1;
#define DZN_RS(MAXI) hipLaunchKernelGGL((row_stats_kernel<MAXI>), dim3(grid), dim3(256), 0, s, x, ldx, rows, C, eps, stats)
  if (C <= 256) DZN_RS(4);
  else if (C <= 512) DZN_RS(8);
  else if (C <= 1024) DZN_RS(16);
  else DZN_RS(32);
#undef DZN_RS
  prof_end(pid, s);
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}

extern "C" int dzn_op_row_stats(const float* x, int64_t ldx, int64_t rows, int32_t C, float eps, float* stats,
                                void* stream) {
  if (!x || !stats) return DZN_E_INVALID;
  return launch_row_stats(x, ldx, rows, C, eps, stats, reinterpret_cast<hipStream_t>(stream));
}

int launch_cast_bf16(const float* x, void* y, int64_t n, hipStream_t s) {
  ProfScope prof_scope_(s, "cast_bf16");
  if (n <= 0) return DZN_OK;
  if (n & 3) return DZN_E_INVALID;
  int64_t g = cdiv64(n / 4, 256);
  g = g > 8192 ? 8192 : g;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)g), dim3(256), 0, s, x, static_cast<u16*>(y), n / 4);
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}

int launch_wave_stats(const float* w, int B, int N, float eps, float* stats, hipStream_t s) {
  ProfScope prof_scope_(s, "wave_stats", 0.0, (double)B * N * 4.0);
  if (B <= 0) return DZN_OK;
  hipLaunchKernelGGL(wave_stats_kernel, dim3(B), dim3(1024), 0, s, w, N, eps, stats);
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}

extern "C" int dzn_op_layernorm(const float* x, int64_t ldx, float* y, int64_t ldy,
                                const float* gamma, const float* beta, int64_t rows, int32_t C,
                                int32_t Cpad, float eps, int32_t gelu, void* stream) {
  if (!x || !y) return DZN_E_INVALID;
  return launch_layernorm(x, ldx, y, ldy, gamma, beta, rows, C, Cpad, eps, gelu,
                          reinterpret_cast<hipStream_t>(stream));
}


// ---- (r4) tracker resets as KERNELS -------------------------------------------------------------------------------------
// The per-forward resets of the |max| trackers were hipMemsetAsync / hipMemsetD32Async calls.  Stream-ordered when launched
// eagerly; but replayed from a captured HIP graph (tests/test_emb_gpu.py::test_forwards_only_enqueue_and_replay_from_a_hip_graph)
// one window's embeddings came out different in the last bits in some replays — the signature of a tracker that was reset
// AFTER its first producer had run (a smaller |max| -> another power-of-two scale of the fp16 split -> other roundings).
// A kernel node is ordered against its neighbours like every other launch of the forward, so the resets are kernels now.
namespace {
__global__ __launch_bounds__(256) void fill_u32_kernel(unsigned* __restrict__ p, unsigned v, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = v;
}
}  // namespace

int launch_fill_u32(void* p, unsigned value, int64_t n, hipStream_t st) {
  if (n <= 0) return DZN_OK;
  int64_t g = (n + 255) / 256;
  g = g > 1024 ? 1024 : g;
  hipLaunchKernelGGL(fill_u32_kernel, dim3((unsigned)g), dim3(256), 0, st, static_cast<unsigned*>(p), value, n);
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}
