// conformer.hip — the non-GEMM pieces of the EEND-Conformer head.
//
// glu_dwconv_kernel: the middle of ConvolutionModule.forward (conformer.py:203-208)
//     GLU(dim=channel) -> depthwise Conv1d(k, pad=(k-1)/2) -> BatchNorm1d(eval) -> Swish
//   BatchNorm is folded into the depthwise taps / bias when the weights are packed.  One thread
//   per channel, a 32-frame tile per workgroup; the (32 + k - 1) gated inputs of the tile live
//   in registers (statically indexed, fully unrolled), loads are coalesced across channels.
//
// classify_kernel: classifier Linear(A -> n_classes) + LogSoftmax
//   (model_wavlm_conformer.py:261-262) + Powerset.to_multilabel(soft=False)
//   (PA/utils/powerset.py:120-128: one_hot(argmax) @ mapping).  One wavefront per frame; what
//   leaves the device is 4 bytes per frame (u8 multilabel) and, optionally, the log-probs.
#include "common.h"

namespace {

constexpr int DW_TT = 32;

template <int KS, typename TO>
__global__ __launch_bounds__(256) void glu_dwconv_kernel(const float* __restrict__ u, int64_t ldu,
                                                         const float* __restrict__ w,  // [A, KS] folded
                                                         const float* __restrict__ bias,  // [A] folded
                                                         TO* __restrict__ out, int64_t ldo, int L,
                                                         int A) {
  constexpr int PAD = (KS - 1) / 2;
  constexpr int NIN = DW_TT + KS - 1;
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * DW_TT;
  for (int c = threadIdx.x; c < A; c += blockDim.x) {
    float wr[KS];
#pragma unroll
    for (int j = 0; j < KS; ++j) wr[j] = w[c * KS + j];
    const float bc = bias[c];
    float in[NIN];
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
      const int t = t0 - PAD + i;
      float v = 0.f;
      if (t >= 0 && t < L) {
        const float* up = u + ((int64_t)b * L + t) * ldu;
        v = up[c] * sigmoidf_(up[A + c]);
      }
      in[i] = v;
    }
#pragma unroll
    for (int t = 0; t < DW_TT; ++t) {
      if (t0 + t < L) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < KS; ++j) a = fmaf(in[t + j], wr[j], a);
        a += bc;
        st_act(out, ((int64_t)b * L + t0 + t) * ldo + c, swishf_(a));
      }
    }
  }
}

__global__ __launch_bounds__(256) void classify_kernel(const float* __restrict__ z, int64_t ldz,
                                                       const float* __restrict__ W,  // [NC, A]
                                                       const float* __restrict__ bias,
                                                       const uint8_t* __restrict__ mapping,  // [NC, S]
                                                       int64_t rows, int A, int NC, int S,
                                                       float* __restrict__ logp,
                                                       uint8_t* __restrict__ multilabel) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const float* zp = z + row * ldz;
  float logit[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) logit[c] = 0.f;
  for (int i = lane; i < A; i += 64) {
    const float v = zp[i];
#pragma unroll
    for (int c = 0; c < 16; ++c)
      if (c < NC) logit[c] = fmaf(v, W[c * A + i], logit[c]);
  }
  float mx = -INFINITY;
  int arg = 0;
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    if (c < NC) {
      logit[c] = wave_sum(logit[c]) + bias[c];
      if (logit[c] > mx) { mx = logit[c]; arg = c; }  // first maximum, like torch.argmax
    }
  }
  float se = 0.f;
#pragma unroll
  for (int c = 0; c < 16; ++c)
    if (c < NC) se += expf(logit[c] - mx);
  const float lse = mx + logf(se);
  if (logp) {
#pragma unroll
    for (int c = 0; c < 16; ++c)
      if (c < NC && lane == c) logp[row * NC + c] = logit[c] - lse;
  }
  if (multilabel && lane < S) multilabel[row * S + lane] = mapping[arg * S + lane];
}

}  // namespace

int launch_glu_dwconv(const float* u, int64_t ldu, const float* w, const float* bias, void* out,
                      int out_bf16, int64_t ldo, int B, int L, int A, int ks, hipStream_t st) {
  ProfScope prof_scope_(st, "glu_dwconv", 0.0, (double)B * L * A * (8.0 + (out_bf16 ? 2.0 : 4.0)));   // 2 A in, A out
  dim3 grid((L + DW_TT - 1) / DW_TT, B);
  const int threads = A >= 256 ? 256 : (A >= 128 ? 128 : 64);
#define DZN_DW(KSV)                                                                                   \
  do {                                                                                                \
    if (out_bf16)                                                                                     \
      hipLaunchKernelGGL((glu_dwconv_kernel<KSV, u16>), grid, dim3(threads), 0, st, u, ldu, w, bias,  \
                         static_cast<u16*>(out), ldo, L, A);                                          \
    else                                                                                              \
      hipLaunchKernelGGL((glu_dwconv_kernel<KSV, float>), grid, dim3(threads), 0, st, u, ldu, w, bias, \
                         static_cast<float*>(out), ldo, L, A);                                        \
  } while (0)
  if (ks == 31) DZN_DW(31);
  else if (ks == 7) DZN_DW(7);
  else if (ks == 15) DZN_DW(15);
  else return DZN_E_INVALID;
#undef DZN_DW
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}

int launch_classify(const float* z, int64_t ldz, const float* W, const float* bias,
                    const uint8_t* mapping, int64_t rows, int A, int NC, int S, float* logp,
                    uint8_t* multilabel, hipStream_t st) {
  ProfScope prof_scope_(st, "classify", 2.0 * (double)rows * A * NC, (double)rows * (A * 4.0 + NC * 4.0 + S));
  if (NC > 16 || S > 64) return DZN_E_INVALID;
  hipLaunchKernelGGL(classify_kernel, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, st, z, ldz, W,
                     bias, mapping, rows, A, NC, S, logp, multilabel);
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}
