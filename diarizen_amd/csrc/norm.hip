// norm.hip — row LayerNorm (+ optional erf-GELU) and per-window waveform statistics.
//
// LayerNorm rows: torch.nn.functional.layer_norm over the last dim (biased variance,
// eps inside the sqrt).  Sites: channel LN after conv1..6 of the WavLM extractor
// (W2V/components.py:63-70, 119-122, fused with GELU), FeatureProjection.layer_norm
// (:305), EncoderLayer.layer_norm / final_layer_norm (:923-941), Model.lnorm
// (model_wavlm_conformer.py:257) and every Conformer ln_norm (conformer.py).
// HBM-bound: one wavefront per row, the row is held in registers between the two
// passes, so each element is read once and written once.
#include "common.h"

namespace {

template <int MAXI>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int64_t ldx,
                                                        float* __restrict__ y, int64_t ldy,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        int64_t rows, int C, int Cpad, float eps,
                                                        int gelu) {
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
    const int idx = lane + 64 * i;
    const float dv = idx < C ? v[i] - mean : 0.f;
    sq += dv * dv;
  }
  const float var = wave_sum(sq) / (float)C;
  const float rstd = 1.0f / sqrtf(var + eps);
  float* yp = y + row * ldy;
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    const int idx = lane + 64 * i;
    if (idx < C) {
      float o = (v[i] - mean) * rstd;
      if (gamma) o = o * gamma[idx] + beta[idx];
      if (gelu) o = gelu_erf(o);
      yp[idx] = o;
    } else if (idx < Cpad) {
      yp[idx] = 0.f;
    }
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

}  // namespace

int launch_layernorm(const float* x, int64_t ldx, float* y, int64_t ldy, const float* g,
                     const float* b, int64_t rows, int C, int Cpad, float eps, int gelu,
                     hipStream_t s) {
  if (rows <= 0) return DZN_OK;
  if (C <= 0 || C > 2048 || Cpad > 2048) return DZN_E_INVALID;
  const unsigned grid = (unsigned)cdiv64(rows, 4);
  const int need = (Cpad > C ? Cpad : C);
  if (need <= 256)
    hipLaunchKernelGGL(layernorm_kernel<4>, dim3(grid), dim3(256), 0, s, x, ldx, y, ldy, g, b, rows,
                       C, Cpad, eps, gelu);
  else if (need <= 512)
    hipLaunchKernelGGL(layernorm_kernel<8>, dim3(grid), dim3(256), 0, s, x, ldx, y, ldy, g, b, rows,
                       C, Cpad, eps, gelu);
  else if (need <= 1024)
    hipLaunchKernelGGL(layernorm_kernel<16>, dim3(grid), dim3(256), 0, s, x, ldx, y, ldy, g, b,
                       rows, C, Cpad, eps, gelu);
  else
    hipLaunchKernelGGL(layernorm_kernel<32>, dim3(grid), dim3(256), 0, s, x, ldx, y, ldy, g, b,
                       rows, C, Cpad, eps, gelu);
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}

int launch_wave_stats(const float* w, int B, int N, float eps, float* stats, hipStream_t s) {
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
