// post.hip — on-device glue between the two device stages, so a window never leaves HBM between
// segmentation and embedding:
//   * median filter along frames of the hard multilabel decisions
//     (diarizen/pipelines/inference.py:131-132: scipy.ndimage.median_filter(size=(1,11,1),
//      mode="reflect"); on {0,1} data the median of an odd window is a majority vote);
//   * embedding masks (PA/pipelines/speaker_diarization.py:268-322): frames with >= 2 active
//     speakers are zeroed ("clean" mask); a speaker falls back to its full mask when its clean
//     mask has <= min_num_frames frames.
// Integer/byte work, one workgroup per window, everything staged in LDS.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void prepare_masks_kernel(const uint8_t* __restrict__ ml, int L, int S,
                                                            int median, int exclude_overlap,
                                                            int min_num_frames,
                                                            uint8_t* __restrict__ filtered,
                                                            float* __restrict__ masks) {
  extern __shared__ unsigned char sm[];
  unsigned char* raw = sm;                 // [L*S]
  unsigned char* fil = sm + L * S;         // [L*S]
  int* cnt = reinterpret_cast<int*>(sm + 2 * ((L * S + 3) & ~3));  // [S] clean-frame counts
  const int b = blockIdx.x;
  const uint8_t* in = ml + (int64_t)b * L * S;
  for (int i = threadIdx.x; i < L * S; i += 256) raw[i] = in[i];
  if (threadIdx.x < S) cnt[threadIdx.x] = 0;
  __syncthreads();
  const int half = median / 2;
  for (int i = threadIdx.x; i < L * S; i += 256) {
    const int t = i / S, s = i - t * S;
    unsigned char v = raw[i];
    if (median > 1) {
      int ones = 0;
      for (int d = -half; d <= half; ++d) {
        int tt = t + d;
        // scipy 'reflect' (d c b a | a b c d | d c b a); windows longer than the signal clamp
        if (tt < 0) tt = -tt - 1;
        if (tt >= L) tt = 2 * L - tt - 1;
        tt = tt < 0 ? 0 : (tt >= L ? L - 1 : tt);
        ones += raw[tt * S + s];
      }
      v = ones > half ? 1 : 0;
    }
    fil[i] = v;
  }
  __syncthreads();
  if (filtered) {
    uint8_t* out = filtered + (int64_t)b * L * S;
    for (int i = threadIdx.x; i < L * S; i += 256) out[i] = fil[i];
  }
  if (!masks) return;
  // clean-frame counts per speaker
  for (int t = threadIdx.x; t < L; t += 256) {
    int act = 0;
    for (int s = 0; s < S; ++s) act += fil[t * S + s];
    if (act < 2)
      for (int s = 0; s < S; ++s)
        if (fil[t * S + s]) atomicAdd(&cnt[s], 1);
  }
  __syncthreads();
  float* mo = masks + (int64_t)b * S * L;
  for (int i = threadIdx.x; i < L * S; i += 256) {
    const int s = i / L, t = i - s * L;
    int act = 0;
    for (int k = 0; k < S; ++k) act += fil[t * S + k];
    const unsigned char full = fil[t * S + s];
    const bool use_clean = exclude_overlap && cnt[s] > min_num_frames;
    mo[i] = (use_clean ? (act < 2 ? full : 0) : full) ? 1.0f : 0.0f;
  }
}

// ---- host post-processing moved to the device -------------------------------------------
// Inference.aggregate (PA/core/inference.py:574-666) for the two uses of the pipeline — speaker counting
// (PA/pipelines/utils/diarization.py:147-155) and reconstruct / to_diarization (PA/pipelines/speaker_diarization.py:
// 400-425, diarization.py:213-239) — is an overlap-add of small integers: window c adds its L frames at
// start_frame[c] (host-computed with the reference's float64 closest_frame, so no float semantics live here).
// Integer atomics: order independent, bit-reproducible.
__global__ __launch_bounds__(256) void count_accum_kernel(const uint8_t* __restrict__ seg, int64_t CL, int L, int S,
                                                          const int32_t* __restrict__ start, int T,
                                                          int32_t* __restrict__ sum, int32_t* __restrict__ cnt) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < CL; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i / L), l = (int)(i - (int64_t)c * L);
    const int t = start[c] + l;
    if (t < 0 || t >= T) continue;
    int tot = 0;
    for (int s = 0; s < S; ++s) tot += seg[i * S + s];
    if (tot) atomicAdd(sum + t, tot);
    atomicAdd(cnt + t, 1);
  }
}

// count[t] = uint8(rint(sum / max(cnt, 1e-12))) in float32 like the reference's float32 accumulators; frames no
// window covers are `missing = 0`
__global__ __launch_bounds__(256) void count_finalize_kernel(const int32_t* __restrict__ sum, const int32_t* __restrict__ cnt,
                                                             int T, uint8_t* __restrict__ count) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const float avg = cnt[t] ? __fdiv_rn((float)sum[t], fmaxf((float)cnt[t], 1e-12f)) : 0.f;
  count[t] = (uint8_t)rintf(avg);
}

// act[t, k] += max_s { seg[c, l, s] : hard[c, s] == k }   (skip_average=True; clusters absent from a window are NaN in
// the reference = contribute nothing)
__global__ __launch_bounds__(256) void cluster_accum_kernel(const uint8_t* __restrict__ seg, const int8_t* __restrict__ hard,
                                                            int64_t CL, int L, int S, const int32_t* __restrict__ start,
                                                            int T, int K, int32_t* __restrict__ act) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < CL; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i / L), l = (int)(i - (int64_t)c * L);
    const int t = start[c] + l;
    if (t < 0 || t >= T) continue;
    unsigned mask = 0;
    for (int s = 0; s < S; ++s) {
      const int k = hard[c * S + s];
      if (k >= 0 && k < K && seg[i * S + s]) mask |= 1u << k;
    }
    while (mask) {
      const int k = __ffs(mask) - 1;
      mask &= mask - 1;
      atomicAdd(act + (int64_t)t * K + k, 1);
    }
  }
}

}  // namespace

extern "C" int dzn_speaker_count(const uint8_t* d_seg, int32_t C, int32_t L, int32_t S, const int32_t* d_start_frame,
                                 int32_t T, int32_t* d_work, uint8_t* d_count, void* stream) {
  if (!d_seg || !d_start_frame || !d_work || !d_count || C < 0 || L < 1 || S < 1 || T < 1) return DZN_E_INVALID;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (hipMemsetAsync(d_work, 0, sizeof(int32_t) * 2 * (size_t)T, st) != hipSuccess) return DZN_E_HIP;
  const int64_t CL = (int64_t)C * L;
  if (CL > 0) {
    int64_t g = cdiv64(CL, 256);
    g = g > 8192 ? 8192 : g;
    hipLaunchKernelGGL(count_accum_kernel, dim3((unsigned)g), dim3(256), 0, st, d_seg, CL, L, S, d_start_frame, T,
                       d_work, d_work + T);
  }
  hipLaunchKernelGGL(count_finalize_kernel, dim3((T + 255) / 256), dim3(256), 0, st, d_work, d_work + T, T, d_count);
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}

extern "C" int dzn_cluster_activations(const uint8_t* d_seg, const int8_t* d_hard, int32_t C, int32_t L, int32_t S,
                                       const int32_t* d_start_frame, int32_t T, int32_t K, int32_t* d_act, void* stream) {
  if (!d_seg || !d_hard || !d_start_frame || !d_act || C < 0 || L < 1 || S < 1 || T < 1 || K < 1 || K > 32)
    return DZN_E_INVALID;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (hipMemsetAsync(d_act, 0, sizeof(int32_t) * (size_t)T * K, st) != hipSuccess) return DZN_E_HIP;
  const int64_t CL = (int64_t)C * L;
  if (CL > 0) {
    int64_t g = cdiv64(CL, 256);
    g = g > 8192 ? 8192 : g;
    hipLaunchKernelGGL(cluster_accum_kernel, dim3((unsigned)g), dim3(256), 0, st, d_seg, d_hard, CL, L, S, d_start_frame,
                       T, K, d_act);
  }
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}

int launch_prepare_masks(const uint8_t* ml, int B, int L, int S, int median, int exclude_overlap,
                         int min_num_frames, uint8_t* filtered, float* masks, hipStream_t st) {
  ProfScope prof_scope_(st, "prepare_masks", 0.0, (double)B * L * S * (2.0 + 4.0));
  if (B <= 0) return DZN_OK;
  if (S < 1 || S > 8 || L < 1 || (median > 1 && !(median & 1))) return DZN_E_INVALID;
  const size_t lds = 2 * (((size_t)L * S + 3) & ~(size_t)3) + 8 * sizeof(int);
  if (lds > 64 * 1024) return DZN_E_INVALID;
  hipLaunchKernelGGL(prepare_masks_kernel, dim3(B), dim3(256), lds, st, ml, L, S, median,
                     exclude_overlap, min_num_frames, filtered, masks);
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}
