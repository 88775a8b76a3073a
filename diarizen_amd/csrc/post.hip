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

}  // namespace

int launch_prepare_masks(const uint8_t* ml, int B, int L, int S, int median, int exclude_overlap,
                         int min_num_frames, uint8_t* filtered, float* masks, hipStream_t st) {
  ProfScope prof_scope_(st, "prepare_masks");
  if (B <= 0) return DZN_OK;
  if (S < 1 || S > 8 || L < 1 || (median > 1 && !(median & 1))) return DZN_E_INVALID;
  const size_t lds = 2 * (((size_t)L * S + 3) & ~(size_t)3) + 8 * sizeof(int);
  if (lds > 64 * 1024) return DZN_E_INVALID;
  hipLaunchKernelGGL(prepare_masks_kernel, dim3(B), dim3(256), lds, st, ml, L, S, median,
                     exclude_overlap, min_num_frames, filtered, masks);
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}
