// frontend.hip — first layer of the WavLM convolutional feature extractor, plus the small
// glue kernels around the encoder.
//
// conv0 (W2V/components.py:119-122 with Conv1d(1 -> C0, k=10, s=5, bias=False)):
//   large : [waveform LayerNorm over the window, W2V/model.py:113] -> conv -> LayerNorm over the
//           C0 channels of every frame (components.py:63-70) -> erf-GELU   (all fused here)
//   base  : conv (raw) ; GroupNorm(C0, C0) = per-channel norm over TIME (components.py:1248-1253)
//           needs whole-window statistics -> col_stats_kernel + gn_gelu_kernel
// The stage is HBM-write bound (0.5 MB in, C0*T0*4 B out per window; half that in the bf16 engine
// mode, which stores the activations as bf16).  Layout: channels-last [B, T0, Cp] so that the next
// conv is a plain contraction over overlapping rows.  A workgroup stages a run of normalised
// samples in LDS (coalesced), every lane keeps its channels' 10 taps in registers, a wavefront
// produces one frame per iteration and stores 256 contiguous bytes per channel group.
#include <type_traits>

#include "common.h"

namespace {

constexpr int FR_PER_BLOCK = 64;   // frames per workgroup (16 per wavefront)

template <int CPL, bool LN, typename TO>
__global__ __launch_bounds__(256) void conv0_kernel(const float* __restrict__ wave, int N,
                                                    const float* __restrict__ stats,  // [B,2] or null
                                                    const float* __restrict__ w,      // [C0, k]
                                                    const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, int C0, int Cp,
                                                    int k, int s, int T0, float eps,
                                                    TO* __restrict__ out,
                                                    const float* __restrict__ lnq /* [10 + 100] or null */) {
  __shared__ float sx[FR_PER_BLOCK * 8 + 32];
  __shared__ float2 sst[FR_PER_BLOCK];   // per-frame (mean, rstd) of the channel LayerNorm
  const int b = blockIdx.y;
  const int f0 = blockIdx.x * FR_PER_BLOCK;
  const int tid = threadIdx.x, lane = tid & 63, wave_id = tid >> 6;
  const float mean = stats ? stats[2 * b] : 0.f;
  const float rstd = stats ? stats[2 * b + 1] : 1.f;
  const int nfr = min(FR_PER_BLOCK, T0 - f0);
  const int nsamp = (nfr - 1) * s + k;
  const float* wp = wave + (int64_t)b * N + (int64_t)f0 * s;
  for (int i = tid; i < nsamp; i += 256) sx[i] = (wp[i] - mean) * rstd;

  // lane -> channels 4*lane + e + 256*g (e = j % 4, g = j / 4): four consecutive channels per lane, so a
  // frame leaves as 16-byte stores (1 KiB contiguous per wavefront instruction; dword stores ran the
  // HBM write stream at 2.8 TB/s, float4 stores reach > 4)
  static_assert(CPL % 4 == 0, "channels per lane in groups of 4");
  float wr[CPL][10];
  float gr[CPL], br[CPL];
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    const int c = 256 * (j >> 2) + 4 * lane + (j & 3);
#pragma unroll
    for (int t = 0; t < 10; ++t) wr[j][t] = (c < C0 && t < k) ? w[c * k + t] : 0.f;
    gr[j] = (LN && c < C0) ? gamma[c] : 1.f;
    br[j] = (LN && c < C0) ? beta[c] : 0.f;
  }
  __syncthreads();
  // LayerNorm statistics of a frame WITHOUT touching its C0 outputs: y_c = w_c . x (x = the frame's k samples), so
  //   mean_c y = wbar . x   and   var_c y = x^T Q x = |F x|^2   with wbar = mean_c w_c, Q = cov_c(w_c) = F^T F (k x k, PSD, factored in
  // double at weight-load time).  One thread per frame here; the frame loop below has no cross-lane reductions left.
  const bool qstats = LN && lnq != nullptr;
  if (qstats && tid < nfr) {
    float xv[10];
#pragma unroll
    for (int t = 0; t < 10; ++t) xv[t] = t < k ? sx[tid * s + t] : 0.f;
    float mu = 0.f, var = 0.f;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      mu = fmaf(lnq[i], xv[i], mu);
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < 10; ++j) q = fmaf(lnq[10 + i * 10 + j], xv[j], q);
      var = fmaf(q, q, var);     // lnq rows are the factor F of Q = F^T F: a sum of squares
    }
    sst[tid] = make_float2(mu, 1.0f / sqrtf(fmaxf(var, 0.f) + eps));
  }
  if (qstats) __syncthreads();

  for (int f = wave_id; f < nfr; f += 4) {
    float xv[10];
#pragma unroll
    for (int t = 0; t < 10; ++t) xv[t] = t < k ? sx[f * s + t] : 0.f;
    float acc[CPL];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      float a = 0.f;
#pragma unroll
      for (int t = 0; t < 10; ++t) a = fmaf(xv[t], wr[j][t], a);
      acc[j] = a;
      sum += a;  // channels >= C0 have zero taps -> contribute 0
    }
    TO* op = out + ((int64_t)b * T0 + f0 + f) * Cp;
    float mu = 0.f, rs = 1.f;
    if (qstats) {
      const float2 st2 = sst[f];
      mu = st2.x;
      rs = st2.y;
    } else if constexpr (LN) {
      mu = wave_sum(sum) / (float)C0;
      float sq = 0.f;
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        const float d = (256 * (j >> 2) + 4 * lane + (j & 3) < C0) ? acc[j] - mu : 0.f;
        sq += d * d;
      }
      rs = 1.0f / sqrtf(wave_sum(sq) / (float)C0 + eps);
    }
#pragma unroll
    for (int g = 0; g < CPL / 4; ++g) {
      const int c0 = 256 * g + 4 * lane;
      if (c0 >= Cp) continue;          // Cp % 4 == 0: a group is entirely inside or outside the padded row
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int j = 4 * g + e;
        float v = LN ? gelu_erf((acc[j] - mu) * rs * gr[j] + br[j]) : acc[j];
        o[e] = (c0 + e < C0) ? v : 0.f;
      }
      if constexpr (sizeof(TO) == 4) {
        *reinterpret_cast<float4*>(op + c0) = make_float4(o[0], o[1], o[2], o[3]);
      } else {
        u16 h[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) st_act(h, e, o[e]);
        *reinterpret_cast<ushort4*>(op + c0) = make_ushort4(h[0], h[1], h[2], h[3]);
      }
    }
  }
}

// ---- GroupNorm(C, C) = per (window, channel) normalisation over TIME (components.py:1248-1253) ----
// r2's statistics kernel ran (C / 64) x B workgroups, each walking all T rows twice with one 4-byte load per thread and
// row: 64 workgroups on 256 CUs, 2.6 ms for 390 MB at BASELINE configs[1] (0.15 TB/s, 29 % of that step).  Now:
//   gn_partial_kernel : grid (T / GN_ROWS, B); a workgroup owns GN_ROWS consecutive rows of one window, threads own a
//                       channel QUAD (float4 loads, coalesced along the channel axis) and a row phase; pass 1 = chunk mean,
//                       pass 2 = centred sum of squares over the same rows (second read comes from L2) -> (mean, M2) per
//                       (window, chunk, channel): the two-pass arithmetic of the reference inside a chunk;
//   gn_finalize_kernel: one thread per (window, channel) merges the chunks in order with Chan's update (deterministic)
//                       -> (mean, rstd);
//   gn_gelu_kernel    : the element pass, float4 wide.
constexpr int GN_ROWS = 128;

__global__ __launch_bounds__(256) void gn_partial_kernel(const float* __restrict__ x, int T, int C, int Cp, int64_t ld,
                                                         float* __restrict__ part /* [B, nchunk, Cp, 2] */) {
  __shared__ float4 red[256];
  const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
  const int Q = Cp >> 2;                        // channel quads (Cp % 4 == 0, Q <= 256)
  const int nph = 256 / Q;                      // row phases
  const int q = threadIdx.x % Q, ph = threadIdx.x / Q;
  const bool on = ph < nph;
  const int r0 = chunk * GN_ROWS, r1 = min(T, r0 + GN_ROWS), n = r1 - r0;
  const float* xp = x + ((int64_t)b * T + r0) * ld + 4 * q;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (on)
    for (int t = ph; t < n; t += nph) {
      const float4 v = *reinterpret_cast<const float4*>(xp + (int64_t)t * ld);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  red[threadIdx.x] = s;
  __syncthreads();
  float4 mean = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int p = 0; p < nph; ++p) {               // fixed order: deterministic
    const float4 v = red[p * Q + q];
    mean.x += v.x; mean.y += v.y; mean.z += v.z; mean.w += v.w;
  }
  const float inv = 1.0f / (float)n;
  mean.x *= inv; mean.y *= inv; mean.z *= inv; mean.w *= inv;
  __syncthreads();
  float4 m2 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (on)
    for (int t = ph; t < n; t += nph) {
      const float4 v = *reinterpret_cast<const float4*>(xp + (int64_t)t * ld);
      const float dx = v.x - mean.x, dy = v.y - mean.y, dz = v.z - mean.z, dw = v.w - mean.w;
      m2.x += dx * dx; m2.y += dy * dy; m2.z += dz * dz; m2.w += dw * dw;
    }
  red[threadIdx.x] = m2;
  __syncthreads();
  if (ph == 0) {
    float4 t2 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = 0; p < nph; ++p) {
      const float4 v = red[p * Q + q];
      t2.x += v.x; t2.y += v.y; t2.z += v.z; t2.w += v.w;
    }
    float* o = part + (((int64_t)b * nchunk + chunk) * Cp + 4 * q) * 2;
    o[0] = mean.x; o[1] = t2.x; o[2] = mean.y; o[3] = t2.y; o[4] = mean.z; o[5] = t2.z; o[6] = mean.w; o[7] = t2.w;
  }
}

__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ part, int B, int T, int C, int Cp, int nchunk,
                                                          float eps, float* __restrict__ stats /* [B, C, 2] */) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i - b * C;
  float mean = 0.f, m2 = 0.f, cnt = 0.f;
  for (int k = 0; k < nchunk; ++k) {            // Chan et al.: merge (cnt, mean, M2) with the chunk's (n, mu, m)
    const float* p = part + (((int64_t)b * nchunk + k) * Cp + c) * 2;
    const float n = (float)min(GN_ROWS, T - k * GN_ROWS);
    const float delta = p[0] - mean, tot = cnt + n;
    mean += delta * (n / tot);
    m2 += p[1] + delta * delta * (cnt * n / tot);
    cnt = tot;
  }
  stats[2 * (int64_t)i] = mean;
  stats[2 * (int64_t)i + 1] = 1.0f / sqrtf(m2 / (float)T + eps);
}

// y[b,t,c] = gelu((x - mean[b,c]) * rstd[b,c] * gamma[c] + beta[c]); y may alias x when TO = float;
// columns [C, Cp) of y are zeroed.  One float4 of channels per thread and step (Cp % 4 == 0, ld % 4 == 0).
template <typename TO>
__global__ __launch_bounds__(256) void gn_gelu_kernel(const float* x, TO* y, int T, int C, int Cp,
                                                      int64_t ld, const float* __restrict__ stats,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float* __restrict__ amax /* [B] or null */) {
  const int b = blockIdx.y;
  const int Q = Cp >> 2;
  const int64_t n = (int64_t)T * Q;
  float mx = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int t = (int)(i / Q), c0 = 4 * (int)(i - (int64_t)t * Q);
    const int64_t off = ((int64_t)b * T + t) * ld + c0;
    const float4 v = *reinterpret_cast<const float4*>(x + off);
    const float in[4] = {v.x, v.y, v.z, v.w};
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = c0 + e;
      if (c < C) {
        const float mu = stats[((int64_t)b * C + c) * 2], rs = stats[((int64_t)b * C + c) * 2 + 1];
        o[e] = gelu_erf((in[e] - mu) * rs * gamma[c] + beta[c]);
        mx = fmaxf(mx, fabsf(o[e]));
      } else {
        o[e] = 0.f;
      }
    }
    if constexpr (std::is_same<TO, float>::value) {
      *reinterpret_cast<float4*>(y + off) = make_float4(o[0], o[1], o[2], o[3]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) st_act(y, off + e, o[e]);
    }
  }
  // per-window |max| of the output: the scale of conv1's fp16 two-term split (DZN_PREC_F32_H2).  One atomic per
  // WORKGROUP (wave maxima through LDS): per-wavefront atomics on 32 addresses cost 0.3 ms at BASELINE configs[1].
  if (amax) {
    __shared__ float wmax[4];
    const float m = wave_max(mx);
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) track_amax_lane(amax + b, fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3])));
  }
}

// xpad[b, t + pad, :] = x[b, t, :], zero borders (input of the positional conv)
template <typename TO>
__global__ __launch_bounds__(256) void pad_rows_kernel(const float* __restrict__ x, TO* __restrict__ xpad,
                                                       int L, int Lp, int pad, int D4) {
  const int b = blockIdx.y;
  const int64_t n = (int64_t)Lp * D4;
  const float4* xs = reinterpret_cast<const float4*>(x) + (int64_t)b * L * D4;
  TO* xd = xpad + (int64_t)b * Lp * D4 * 4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int t = (int)(i / D4) - pad;
    const int c = (int)(i % D4);
    const float4 v = (t >= 0 && t < L) ? xs[(int64_t)t * D4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
    st_act(xd, 4 * i + 0, v.x);
    st_act(xd, 4 * i + 1, v.y);
    st_act(xd, 4 * i + 2, v.z);
    st_act(xd, 4 * i + 3, v.w);
  }
}

// ws = (init ? 0 : ws) + w * x   (layer-weighted sum, model_wavlm_conformer.py:236,253-254)
__global__ __launch_bounds__(256) void ws_accum_kernel(const float* __restrict__ x,
                                                       float* __restrict__ ws, float w, int init,
                                                       int64_t n4) {
  const float4* xs = reinterpret_cast<const float4*>(x);
  float4* wd = reinterpret_cast<float4*>(ws);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 v = xs[i];
    float4 a = init ? make_float4(0.f, 0.f, 0.f, 0.f) : wd[i];
    a.x += w * v.x; a.y += w * v.y; a.z += w * v.z; a.w += w * v.w;
    wd[i] = a;
  }
}

// ws = sum_k w[k] x[k] with the additions in list order, starting from 0 (bit-identical to ws_accum(init) followed by
// ws_accum / epilogue read-modify-writes in the same order; -ffp-contract=off: multiply, then add)
__global__ __launch_bounds__(256) void ws_sum_kernel(const WsSumArgs a, float* __restrict__ ws, int64_t n4) {
  float4* wd = reinterpret_cast<float4*>(ws);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int k = 0;
    for (; k + 4 <= a.n; k += 4) {      // four independent loads in flight per thread
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = reinterpret_cast<const float4*>(a.x[k + u])[i];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float w = a.w[k + u];
        acc.x += w * v[u].x; acc.y += w * v[u].y; acc.z += w * v[u].z; acc.w += w * v[u].w;
      }
    }
    for (; k < a.n; ++k) {
      const float4 v = reinterpret_cast<const float4*>(a.x[k])[i];
      const float w = a.w[k];
      acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
    }
    wd[i] = acc;
  }
}

// x[r, c] *= scale[c]  (dummy_weight, components.py:208), columns >= C untouched
template <typename T>
__global__ __launch_bounds__(256) void col_scale_kernel(T* __restrict__ x, int64_t rows, int C,
                                                        int64_t ld, const float* __restrict__ scale) {
  const int64_t n = rows * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / C;
    const int c = (int)(i - r * C);
    st_act(x, r * ld + c, ld_act(x, r * ld + c) * scale[c]);
  }
}

inline unsigned grid_for(int64_t n, int per = 256, int cap = 4096) {
  int64_t g = cdiv64(n, per);
  return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

template <typename TO>
int conv0_dispatch(const float* wave, int B, int N, const float* stats, const float* w,
                   const float* gamma, const float* beta, int C0, int Cp, int k, int s, int T0,
                   int layer_norm, float eps, TO* out, hipStream_t st, const float* lnq) {
  dim3 grid((T0 + FR_PER_BLOCK - 1) / FR_PER_BLOCK, B);
  const int width = max(C0, Cp);   // a lane owns 4 consecutive channels per 256-channel group
#define DZN_C0(CPLV)                                                                                  \
  do {                                                                                                \
    if (layer_norm)                                                                                   \
      hipLaunchKernelGGL((conv0_kernel<CPLV, true, TO>), grid, dim3(256), 0, st, wave, N, stats, w,   \
                         gamma, beta, C0, Cp, k, s, T0, eps, out, lnq);                               \
    else                                                                                              \
      hipLaunchKernelGGL((conv0_kernel<CPLV, false, TO>), grid, dim3(256), 0, st, wave, N, stats, w,  \
                         gamma, beta, C0, Cp, k, s, T0, eps, out, lnq);                               \
  } while (0)
  if (width <= 256) DZN_C0(4);
  else DZN_C0(8);
#undef DZN_C0
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}

}  // namespace

int launch_conv0(const float* wave, int B, int N, const float* stats, const float* w,
                 const float* gamma, const float* beta, int C0, int Cp, int k, int s, int T0,
                 int layer_norm, float eps, void* out, int out_bf16, hipStream_t st, const float* lnq) {
  if (k > 10 || C0 > 512 || s > 8) return DZN_E_INVALID;
  // algorithmic HBM bytes: read the waveform once, write the [T0, C0] activations once
  const double esz = out_bf16 ? 2.0 : 4.0;
  const int pid = prof_begin(st, layer_norm ? "conv0_ln_gelu" : "conv0_raw",
                             2.0 * B * (double)T0 * C0 * k, B * (4.0 * N + esz * (double)T0 * C0));
  int rc;
  if (out_bf16)
    rc = conv0_dispatch(wave, B, N, stats, w, gamma, beta, C0, Cp, k, s, T0, layer_norm, eps,
                        static_cast<u16*>(out), st, lnq);
  else
    rc = conv0_dispatch(wave, B, N, stats, w, gamma, beta, C0, Cp, k, s, T0, layer_norm, eps,
                        static_cast<float*>(out), st, lnq);
  prof_end(pid, st);
  return rc;
}

int launch_groupnorm_gelu(const float* x, void* y, int y_bf16, int B, int T, int C, int Cp, int64_t ld,
                          const float* gamma, const float* beta, float eps, float* stats, hipStream_t st, float* amax) {
  // algorithmic bytes: x read once for the statistics, once for the element pass, y written once
  ProfScope prof_scope_(st, "groupnorm_gelu", 0.0, (double)B * T * C * (y_bf16 ? 10.0 : 12.0));
  if ((Cp & 3) || (ld & 3) || Cp > 1024) return DZN_E_INVALID;
  const int nchunk = (T + GN_ROWS - 1) / GN_ROWS;
  // scratch for the chunk partials: behind the [B, C, 2] statistics (the engine sizes `stats` for it: gn_stats_floats())
  float* part = stats + (int64_t)B * C * 2;
  hipLaunchKernelGGL(gn_partial_kernel, dim3(nchunk, B), dim3(256), 0, st, x, T, C, Cp, ld, part);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((B * C + 255) / 256), dim3(256), 0, st, part, B, T, C, Cp, nchunk, eps, stats);
  if (y_bf16)
    hipLaunchKernelGGL(gn_gelu_kernel<u16>, dim3(grid_for((int64_t)T * (Cp / 4), 256, 512), B), dim3(256), 0, st, x,
                       static_cast<u16*>(y), T, C, Cp, ld, stats, gamma, beta, amax);
  else
    hipLaunchKernelGGL(gn_gelu_kernel<float>, dim3(grid_for((int64_t)T * (Cp / 4), 256, 512), B), dim3(256), 0, st, x,
                       static_cast<float*>(y), T, C, Cp, ld, stats, gamma, beta, amax);
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}

// floats the `stats` argument of launch_groupnorm_gelu must hold: [B, C, 2] statistics + [B, nchunk, Cp, 2] partials
int64_t gn_stats_floats(int B, int T, int C, int Cp) {
  return (int64_t)B * C * 2 + (int64_t)B * ((T + GN_ROWS - 1) / GN_ROWS) * Cp * 2;
}

int launch_pad_rows(const float* x, void* xpad, int out_bf16, int B, int L, int Lp, int pad, int D,
                    hipStream_t st) {
  ProfScope prof_scope_(st, "pad_rows", 0.0, (double)B * (L + Lp) * D * 4.0);
  if (out_bf16)
    hipLaunchKernelGGL(pad_rows_kernel<u16>, dim3(grid_for((int64_t)Lp * (D / 4)), B), dim3(256), 0, st,
                       x, static_cast<u16*>(xpad), L, Lp, pad, D / 4);
  else
    hipLaunchKernelGGL(pad_rows_kernel<float>, dim3(grid_for((int64_t)Lp * (D / 4)), B), dim3(256), 0, st,
                       x, static_cast<float*>(xpad), L, Lp, pad, D / 4);
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}

int launch_ws_accum(const float* x, float* ws, float w, int init, int64_t n, hipStream_t st) {
  ProfScope prof_scope_(st, "ws_accum", 0.0, (double)n * (init ? 8.0 : 12.0));
  hipLaunchKernelGGL(ws_accum_kernel, dim3(grid_for(n / 4)), dim3(256), 0, st, x, ws, w, init, n / 4);
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}

int launch_ws_sum(const WsSumArgs& a, float* ws, int64_t n, hipStream_t st) {
  if (a.n <= 0 || a.n > WS_SUM_MAX || (n & 3)) return DZN_E_INVALID;
  ProfScope prof_scope_(st, "ws_sum", 0.0, (double)n * 4.0 * (a.n + 1));
  hipLaunchKernelGGL(ws_sum_kernel, dim3(grid_for(n / 4)), dim3(256), 0, st, a, ws, n / 4);
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}

int launch_col_scale(void* x, int x_bf16, int64_t rows, int C, int64_t ld, const float* scale,
                     hipStream_t st) {
  ProfScope prof_scope_(st, "col_scale", 0.0, (double)rows * C * 8.0);
  if (x_bf16)
    hipLaunchKernelGGL(col_scale_kernel<u16>, dim3(grid_for(rows * C)), dim3(256), 0, st,
                       static_cast<u16*>(x), rows, C, ld, scale);
  else
    hipLaunchKernelGGL(col_scale_kernel<float>, dim3(grid_for(rows * C)), dim3(256), 0, st,
                       static_cast<float*>(x), rows, C, ld, scale);
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}
