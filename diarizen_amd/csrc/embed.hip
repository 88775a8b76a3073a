// embed.hip — non-GEMM kernels of the WeSpeaker ResNet34 embedding path.
//
//  frame_prep_kernel : Kaldi framing of torchaudio.compliance.kaldi.fbank as called at
//      PA/models/embedding/wespeaker/__init__.py:69-78,94-103 — x * 2^15, snip_edges frames of
//      400 samples / shift 160, per-frame DC removal, pre-emphasis 0.97 (replicate pad),
//      Hamming(400, periodic=False), zero pad (the 512-point transform is a contraction with a
//      [512, Kp] cos/sin matrix on the MFMA kernel: 400 non-zero taps only).
//  power_kernel      : |rfft|^2 for bins 0..255 (bin 256 has zero mel weight).
//  log_cmn_kernel    : log(max(mel, eps)) and per-window mean subtraction over frames (:103).
//  stem_conv_kernel  : ResNet conv1 3x3 (1 -> 32) + folded BN + ReLU (resnet.py:358), writes the
//      zero-bordered NHWC image the 3x3 contractions read.
//  stats_pool_kernel : TSTP / StatsPool weighted mean + std for ALL speaker masks of a window
//      from one trunk pass (resnet.py:49-66, PA/models/blocks/pooling.py:44-75,107-131).
#include <type_traits>

#include "common.h"

namespace {

__global__ __launch_bounds__(256) void frame_prep_kernel(const float* __restrict__ wave, int N,
                                                         int T, int flen, int fshift, int Kp,
                                                         const float* __restrict__ window,
                                                         float preemph, float* __restrict__ frames) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int b = blockIdx.y;
  const int t = blockIdx.x * 4 + wv;
  if (t >= T) return;
  const float* xp = wave + (int64_t)b * N + (int64_t)t * fshift;
  float v[8];  // flen <= 512
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = lane + 64 * i;
    v[i] = idx < flen ? xp[idx] * 32768.0f : 0.f;
    s += v[i];
  }
  const float mean = wave_sum(s) / (float)flen;
  float* op = frames + ((int64_t)b * T + t) * Kp;
  // previous sample (idx-1) lives in lane-1, or in lane 63 of the previous register; x[-1] := x[0]
  // (all cross-lane traffic happens here, in uniform control flow)
  float pv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float prev = __shfl_up(v[i], 1, 64);
    const float prev_reg = __shfl(v[i > 0 ? i - 1 : 0], 63, 64);
    if (lane == 0) prev = i > 0 ? prev_reg : v[0];
    pv[i] = prev;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = lane + 64 * i;
    if (idx < flen) {
      const float cur = v[i] - mean;
      const float prev = pv[i] - mean;
      op[idx] = (cur - preemph * prev) * window[idx];
    } else if (idx < Kp) {
      op[idx] = 0.f;
    }
  }
}

__global__ __launch_bounds__(256) void power_kernel(const float* __restrict__ spec, int64_t rows,
                                                    int nb, float* __restrict__ pw) {
  const int64_t n = rows * nb;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / nb;
    const int f = (int)(i - r * nb);
    const float re = spec[r * 2 * nb + f], im = spec[r * 2 * nb + nb + f];
    const float a = sqrtf(re * re + im * im);  // torch: rfft(...).abs().pow(2.0)
    pw[i] = a * a;
  }
}

__global__ __launch_bounds__(256) void log_cmn_kernel(float* __restrict__ mel, int T, int NB,
                                                      float eps) {
  __shared__ float red[16][17];
  const int b = blockIdx.y;
  const int bin = blockIdx.x * 16 + (threadIdx.x & 15);
  const int ph = threadIdx.x >> 4;
  float* mp = mel + (int64_t)b * T * NB;
  float s = 0.f;
  if (bin < NB)
    for (int t = ph; t < T; t += 16) {
      const float v = logf(fmaxf(mp[(int64_t)t * NB + bin], eps));
      mp[(int64_t)t * NB + bin] = v;
      s += v;
    }
  red[ph][threadIdx.x & 15] = s;
  __syncthreads();
  float tot = 0.f;
  for (int i = 0; i < 16; ++i) tot += red[i][threadIdx.x & 15];
  const float mean = tot / (float)T;
  if (bin < NB)
    for (int t = ph; t < T; t += 16) mp[(int64_t)t * NB + bin] -= mean;
}

// fb [B, T, NB]  ->  img [B, NB+2, T+2, C] (interior only; borders stay zero)
// r3: one image per blockIdx.y, a workgroup walks a contiguous run of its pixels; a thread owns ONE channel quad for
// the whole loop (its 36 taps + 4 biases live in registers), stores one float4 per pixel (a pixel's 32 channels = one
// 128-byte line written by 8 neighbouring lanes) and the |max| tracker is updated once per workgroup.  r2's form
// (flat index over everything: three 64-bit div/mods, 36 tap loads, four 4-byte stores and a tracker probe PER ELEMENT)
// wrote its 3 GB per launch at 1.27 TB/s.
template <typename TO>
__global__ __launch_bounds__(256) void stem_conv_kernel(const float* __restrict__ fb, int B, int T,
                                                        int NB, int C,
                                                        const float* __restrict__ w,  // [C, 9] folded
                                                        const float* __restrict__ bias,
                                                        TO* __restrict__ img, float* __restrict__ amax,
                                                        const int* __restrict__ z_count, const int* __restrict__ z_list) {
  const int cq = C >> 2;                       // channel quads per pixel (8 for the 32-channel stem)
  int b = blockIdx.y;
  if (z_list) {                                // device-chosen subset of the batch (windows with an active speaker)
    if (b >= z_count[0]) return;
    b = z_list[b];
  }
  const int q = threadIdx.x % cq, pl = threadIdx.x / cq, ppb = 256 / cq;     // pixel lane, pixels per sweep
  float wr[4][9], br[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    br[c] = bias[q * 4 + c];
#pragma unroll
    for (int k = 0; k < 9; ++k) wr[c][k] = w[(q * 4 + c) * 9 + k];
  }
  const int npix = NB * T;
  const int per = (npix + gridDim.x - 1) / gridDim.x;
  const int p0 = blockIdx.x * per, p1 = min(npix, p0 + per);
  const float* fbb = fb + (int64_t)b * T * NB;
  TO* ib = img + (int64_t)b * (NB + 2) * (T + 2) * C;
  float mx = 0.f;
  for (int p = p0 + pl; p < p1; p += ppb) {
    const int h = p / T, wv = p - h * T;       // pixel (mel bin h, frame wv): consecutive pixels walk along time
    float in[9];
#pragma unroll
    for (int dh = 0; dh < 3; ++dh)
#pragma unroll
      for (int dw = 0; dw < 3; ++dw) {
        const int hh = h + dh - 1, ww = wv + dw - 1;
        in[dh * 3 + dw] = (hh >= 0 && hh < NB && ww >= 0 && ww < T) ? fbb[(int64_t)ww * NB + hh] : 0.f;
      }
    float o[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 9; ++k) acc = fmaf(in[k], wr[c][k], acc);
      o[c] = fmaxf(acc + br[c], 0.f);
    }
    TO* op = ib + ((int64_t)(h + 1) * (T + 2) + wv + 1) * C + q * 4;
    if constexpr (std::is_same<TO, float>::value) {
      *reinterpret_cast<float4*>(op) = make_float4(o[0], o[1], o[2], o[3]);
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) st_act(op, c, o[c]);
    }
    mx = fmaxf(mx, fmaxf(fmaxf(o[0], o[1]), fmaxf(o[2], o[3])));     // post-ReLU: non-negative
  }
  if (amax) {   // per-image |max| tracker (DZN_PREC_F32_H2: scale of the stage-1 convolutions' fp16 split)
    __shared__ float wmax[4];
    const float m = wave_max(mx);
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) track_amax_lane(amax + b, fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3])));
  }
}

// img [B, H+2, W+2, C] (padded NHWC), masks [B, S, L]  ->  stats [B, S, 2*C*H]
// feature index f = c*H + h (rearrange "b d c f -> b (d c) f" of resnet.py:49-66)
template <typename TI>
__global__ __launch_bounds__(256) void stats_pool_kernel(const TI* __restrict__ img, int H, int W,
                                                         int C, const float* __restrict__ masks,
                                                         int S, int L, float* __restrict__ stats,
                                                         const int* __restrict__ active) {
  extern __shared__ float sw[];  // [S][W] interpolated weights
  const int b = blockIdx.y, h = blockIdx.x;
  if (active && !active[b]) {
    // a window without any active speaker: its trunk pass was skipped, its image is stale.  All-zero weights pool to
    // mean = 0 / 1e-8 = 0 and std = sqrt(0 / 2e-8) = 0 for every feature (PA/models/blocks/pooling.py:44-75) — the very
    // values the loop below produces for zero masks, written directly
    for (int c = threadIdx.x; c < C; c += blockDim.x)
      for (int s = 0; s < S; ++s) {
        float* op = stats + ((int64_t)b * S + s) * (2 * C * H);
        op[c * H + h] = 0.f;
        op[C * H + c * H + h] = 0.f;
      }
    return;
  }
  // F.interpolate(mode="nearest"): src = min(floor(dst * (L / W)), L - 1)  (float32 scale)
  const float scale = (float)L / (float)W;
  for (int i = threadIdx.x; i < S * W; i += blockDim.x) {
    const int s = i / W, t = i - s * W;
    int src = (int)floorf((float)t * scale);
    src = src < L - 1 ? src : L - 1;
    sw[i] = (W == L) ? masks[((int64_t)b * S + s) * L + t] : masks[((int64_t)b * S + s) * L + src];
  }
  __syncthreads();
  const TI* base = img + (((int64_t)b * (H + 2) + h + 1) * (W + 2) + 1) * C;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    for (int s = 0; s < S; ++s) {
      const float* wt = sw + s * W;
      float v1 = 0.f, v2 = 0.f, sx = 0.f;
      for (int t = 0; t < W; ++t) {
        const float wv = wt[t];
        v1 += wv;
        v2 += wv * wv;
        sx += ld_act(base, (int64_t)t * C + c) * wv;
      }
      v1 += 1e-8f;
      const float mean = sx / v1;
      float sq = 0.f;
      for (int t = 0; t < W; ++t) {
        const float d = ld_act(base, (int64_t)t * C + c) - mean;
        sq += d * d * wt[t];
      }
      const float var = sq / (v1 - v2 / v1 + 1e-8f);
      float* op = stats + ((int64_t)b * S + s) * (2 * C * H);
      op[c * H + h] = mean;
      op[C * H + c * H + h] = sqrtf(var);
    }
  }
}

// ---- windows without any active speaker need no trunk (their embeddings are seg_1's bias) ----
// flag[b] = 1 when any of the window's S * L mask values is non-zero
__global__ __launch_bounds__(256) void window_active_kernel(const float* __restrict__ masks, int per_window,
                                                            int* __restrict__ flag) {
  const float* m = masks + (int64_t)blockIdx.x * per_window;
  int any = 0;
  for (int i = threadIdx.x; i < per_window; i += 256) any |= m[i] != 0.f;
  any = __syncthreads_or(any);
  if (threadIdx.x == 0) flag[blockIdx.x] = any != 0;
}

// (r4) the subset is built ON THE DEVICE: list[0 .. count) = the active windows in ascending order, count[0] = how many
// (one workgroup; B is a batch size).  Every trunk kernel takes (count, list) and its surplus grid rows exit, so the
// host never reads the flags back — dzn_embed_forward stays enqueue-only.  totals[0] += B, totals[1] += B - count
// (dzn_embed_skip_stats).
__global__ __launch_bounds__(256) void compact_active_kernel(const int* __restrict__ flag, int B, int* __restrict__ count,
                                                             int* __restrict__ list, long long* __restrict__ totals) {
  __shared__ int wsum[4];
  __shared__ int base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) base = 0;
  __syncthreads();
  for (int b0 = 0; b0 < B; b0 += 256) {
    const int b = b0 + threadIdx.x;
    const int f = b < B ? (flag[b] != 0) : 0;
    const unsigned long long bal = __ballot(f);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    if (f) list[off + before] = b;
    __syncthreads();
    if (threadIdx.x == 0) base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    count[0] = base;
    if (totals) {
      totals[0] += B;
      totals[1] += B - base;
    }
  }
}

inline unsigned grid_for(int64_t n, int per = 256, int cap = 8192) {
  int64_t g = cdiv64(n, per);
  return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

int launch_frame_prep(const float* wave, int B, int N, int T, int flen, int fshift, int Kp,
                      const float* window, float preemph, float* frames, hipStream_t st) {
  ProfScope prof_scope_(st, "frame_prep", 0.0, (double)B * N * 4.0 + (double)B * T * Kp * 4.0);
  if (flen > 512 || T <= 0) return DZN_E_INVALID;
  hipLaunchKernelGGL(frame_prep_kernel, dim3((T + 3) / 4, B), dim3(256), 0, st, wave, N, T, flen,
                     fshift, Kp, window, preemph, frames);
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}

int launch_window_active(const float* masks, int B, int per_window, int* flag, hipStream_t st) {
  ProfScope prof_scope_(st, "window_active", 0.0, (double)B * per_window * 4.0);
  hipLaunchKernelGGL(window_active_kernel, dim3(B), dim3(256), 0, st, masks, per_window, flag);
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}

int launch_compact_active(const int* flag, int B, int* count, int* list, long long* totals, hipStream_t st) {
  hipLaunchKernelGGL(compact_active_kernel, dim3(1), dim3(256), 0, st, flag, B, count, list, totals);
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}

int launch_power(const float* spec, int64_t rows, int nb, float* pw, hipStream_t st) {
  ProfScope prof_scope_(st, "power", 0.0, (double)rows * nb * 12.0);
  hipLaunchKernelGGL(power_kernel, dim3(grid_for(rows * nb)), dim3(256), 0, st, spec, rows, nb, pw);
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}

int launch_log_cmn(float* mel, int B, int T, int NB, float eps, hipStream_t st) {
  ProfScope prof_scope_(st, "log_cmn", 0.0, (double)B * T * NB * 12.0);   // read for the mean, read + write
  hipLaunchKernelGGL(log_cmn_kernel, dim3((NB + 15) / 16, B), dim3(256), 0, st, mel, T, NB, eps);
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}

int launch_stem_conv(const float* fb, int B, int T, int NB, int C, const float* w, const float* bias,
                     void* img, int out_bf16, hipStream_t st, float* amax, const int* z_count, const int* z_list) {
  ProfScope prof_scope_(st, "stem_conv", 2.0 * B * NB * (double)T * C * 9.0,
                        (double)B * NB * T * 4.0 + (double)B * NB * T * C * (out_bf16 ? 2.0 : 4.0));   // fbank in, C-channel image out
  if (C % 4 || 256 % (C / 4)) return DZN_E_INVALID;
  // ~8 sweeps of 256 / (C / 4) pixels per workgroup
  const int npix = NB * T, ppb = 256 / (C / 4);
  int gx = (npix + 8 * ppb - 1) / (8 * ppb);
  gx = gx < 1 ? 1 : gx;
  const dim3 grid(gx, B);
  if (out_bf16)
    hipLaunchKernelGGL(stem_conv_kernel<u16>, grid, dim3(256), 0, st, fb, B, T, NB, C, w, bias,
                       static_cast<u16*>(img), amax, z_count, z_list);
  else
    hipLaunchKernelGGL(stem_conv_kernel<float>, grid, dim3(256), 0, st, fb, B, T, NB, C, w, bias,
                       static_cast<float*>(img), amax, z_count, z_list);
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}

int launch_stats_pool(const void* img, int in_bf16, int B, int H, int W, int C, const float* masks, int S,
                      int L, float* stats, hipStream_t st, const int* active) {
  ProfScope prof_scope_(st, "stats_pool", 0.0, (double)B * H * W * C * (in_bf16 ? 2.0 : 4.0) + (double)B * S * L * 4.0);
  const size_t lds = (size_t)S * W * sizeof(float);
  if (in_bf16)
    hipLaunchKernelGGL(stats_pool_kernel<u16>, dim3(H, B), dim3(256), lds, st, static_cast<const u16*>(img),
                       H, W, C, masks, S, L, stats, active);
  else
    hipLaunchKernelGGL(stats_pool_kernel<float>, dim3(H, B), dim3(256), lds, st,
                       static_cast<const float*>(img), H, W, C, masks, S, L, stats, active);
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}
