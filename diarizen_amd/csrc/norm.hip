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

template <typename TI, typename TO>
int launch_ln_typed(const TI* x, int64_t ldx, TO* y, int64_t ldy, const float* g, const float* b,
                    const float* post, int64_t rows, int C, int Cpad, float eps, int gelu, hipStream_t s,
                    float* amax, int64_t amax_unit) {
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
