// conv_split.hip — 3x3 stride-1 convolution, 32 -> 32 channels, for the first ResNet34 stage
// (wespeaker/resnet.py:139-144 BasicBlock.conv1 / conv2 at 32 planes, BatchNorm folded) in the
// fp32-split arithmetic of split.h (DZN_PREC_F32_SPLIT).
//
// Why a dedicated kernel: as a generic contraction (gemm_split.hip, N = 32, K = 9 x 32) every input
// pixel is split into its three bf16 terms NINE times (once per tap that reads it) and the 32-wide
// column tile leaves the in-register split as expensive as the MFMAs — 105 TFLOP/s, the slowest
// contraction class of the pipeline (8.7 % of device time).  Here:
//   * the image is walked as ONE flat pixel axis over the zero-bordered NHWC layout
//     [B][H+2][W+2][32]: a tile is 128 consecutive padded pixels, tap (dh, dw) of pixel q reads pixel
//     q + (dh-1)(W+2) + (dw-1), so a tile needs three contiguous 130-pixel segments; outputs that
//     fall on a border column are computed and masked at the store (2 / (W+2) of the work);
//   * each input pixel of the strip is split ONCE by the staging pass and kept as three bf16 planes in
//     LDS ([segment][pixel][64 B], 16-B chunk XOR ((pixel >> 1) & 3): every fragment — at any of the
//     three dw shifts — is one conflict-free ds_read_b128);
//   * the pre-split weights of a wavefront's 16 output channels (9 taps x 3 planes = 27 fragments)
//     live in registers for the whole persistent loop, so the MFMA phase reads only pixels from LDS;
//   * persistent workgroups, 2 per CU (75 KB of LDS each): the strip of the next tile is fetched into
//     registers while the current one is multiplied, and one workgroup splits while the other multiplies.
// Measured 138 TFLOP/s (2.2 ms per launch at B = 256) against 105 for the generic kernel; the MFMA
// phase still waits on its just-in-time fragment reads (register budget: 108 weight + 56 prefetch).
// Epilogue: + bias (folded BN shift), optional ReLU, optional residual, optional post-ReLU, float4 store.
#include <type_traits>

#include "common.h"
#include "split.h"

namespace {

constexpr int CS_TP = 128;               // output pixels per tile
constexpr int CS_SEG = CS_TP + 2;        // pixels per input segment
constexpr int CS_PLANE = 3 * CS_SEG * 64;  // bytes per 16-bit plane of the strip

// NP = 3: bf16 three-term split (six products per block).  NP = 2 (DZN_PREC_F32_H2): fp16 two-term split (three
// products): the strip is scaled by the exact power of two from the input image's |max| tracker (amax_in[b]) when it
// is split, the weights come pre-scaled per output channel (col_scale), the epilogue multiplies both back.
struct ConvArgs {
  const float* in;
  const u16* W3;       // [32 oc][9 taps][NP planes][32] 16-bit, k order of gemm_split.hip
  const float* bias;   // [32]
  const float* R;      // residual image (same geometry) or nullptr
  float* out;
  int B, Hs, Ws;       // interior size; images are [Hs+2][Ws+2][32]
  int relu, post_relu;
  float* amax;         // |max| tracker of `out` per image (DZN_PREC_F32_H2 consumers) or nullptr
  const float* amax_in;    // NP = 2: per-image |max| of `in`
  const float* col_scale;  // NP = 2: [32] inverse weight scales
  const int* z_count;      // device-chosen subset of the B images: z_count[0] entries of z_list, or nullptr = all
  const int* z_list;
};

template <int NP>
__global__ __launch_bounds__(256, 2) void conv3x3_c32_split_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int mh = wave >> 1, nb = wave & 1;     // wavefront -> 64-pixel half, 16-channel block
  const int P = a.Ws + 2;
  const int64_t img = (int64_t)(a.Hs + 2) * P * 32;   // elements per image
  const int npix = a.Hs * P;                          // flat pixels of the rows that hold outputs
  const int tiles_img = (npix + CS_TP - 1) / CS_TP;
  const int nB = a.z_list ? a.z_count[0] : a.B;       // images this launch works on
  const int last_pix = (a.Hs + 2) * P - 1;

  // this wavefront's weight fragments: A operand rows = output channels nb*16 + lr, k = 8 lq .. 8 lq + 7
  u32x4 wf[9][NP];
  {
    const u16* wp = a.W3 + (int64_t)(nb * 16 + lr) * (9 * NP * 32) + lq * 8;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int p = 0; p < NP; ++p) wf[t][p] = *reinterpret_cast<const u32x4*>(wp + t * NP * 32 + p * 32);
  }
  const float4 b4 = *reinterpret_cast<const float4*>(a.bias + nb * 16 + lq * 4);
  float4 cs4 = make_float4(1.f, 1.f, 1.f, 1.f);
  if constexpr (NP == 2) cs4 = *reinterpret_cast<const float4*>(a.col_scale + nb * 16 + lq * 4);
  float out_amax = 0.f;

  // staging registers: 7 items per thread (3 segments x 130 pixels x 4 chunks of 8 channels = 1560 items);
  // the strip of tile t+1 is fetched into them while tile t is multiplied
  constexpr int NITEM = 3 * CS_SEG * 4, NIT = 7;
  static_assert(NITEM <= NIT * 256, "7 items per thread");
  float4 u4[NIT], v4[NIT];
  auto fetch = [&](int tile) {
    const int bi = tile / tiles_img;
    const int b = a.z_list ? a.z_list[bi] : bi;
    const int q0 = P + (tile - bi * tiles_img) * CS_TP;
    const float* ib = a.in + (int64_t)b * img;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      int item = tid + 256 * i;
      item = item < NITEM ? item : NITEM - 1;
      const int seg = item / (CS_SEG * 4);
      const int r = item - seg * (CS_SEG * 4);
      int gq = q0 + (seg - 1) * P - 1 + (r >> 2);
      gq = gq < 0 ? 0 : (gq > last_pix ? last_pix : gq);  // only masked outputs ever see a clamped pixel
      const float* src = ib + (int64_t)gq * 32 + 4 * (r & 3);
      u4[i] = *reinterpret_cast<const float4*>(src);        // channels 4c .. 4c+3
      v4[i] = *reinterpret_cast<const float4*>(src + 16);   // channels 16+4c .. 16+4c+3
    }
  };
  auto split_store = [&](float in_scale) {
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int item = tid + 256 * i;
      const int seg = item / (CS_SEG * 4);
      const int r = item - seg * (CS_SEG * 4);
      const int px = r >> 2, c = r & 3;
      u32x4 pf[NP];
      split_np<NP>((f32x4){u4[i].x, u4[i].y, u4[i].z, u4[i].w}, (f32x4){v4[i].x, v4[i].y, v4[i].z, v4[i].w}, in_scale, pf);
      if (item < NITEM) {
        const int off = (seg * CS_SEG + px) * 64 + ((c ^ ((px >> 1) & 3)) << 4);
#pragma unroll
        for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4*>(smem + p * CS_PLANE + off) = pf[p];
      }
    }
  };

  // Work distribution: workgroup id -> (XCD = id % 8, slot = id / 8).  Each XCD (private 4 MB L2) owns
  // whole images (b = xcd, xcd + 8, ...) and its slots walk an image's tiles in order, so the three
  // input rows a tile needs are still in that L2 from the tiles one row up — without this the input
  // was fetched from HBM ~3.4 times (PMC FETCH_SIZE: 7.2 GB per launch for 2.1 GB of input; now 3.4 GB).
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = (gridDim.x + 7 - xcd) >> 3;
  const int per_img = slot < tiles_img ? (tiles_img - slot + nslot - 1) / nslot : 0;  // my tiles per image
  const int nimg = xcd < nB ? (nB - xcd + 7) / 8 : 0;
  const int nmine = per_img * nimg;
  auto tile_of = [&](int j) {   // j-th tile of this workgroup
    const int ii = j / per_img;
    return (xcd + 8 * ii) * tiles_img + slot + (j - ii * per_img) * nslot;
  };
  if (nmine > 0) fetch(tile_of(0));
  for (int j = 0; j < nmine; ++j) {
    const int tile = tile_of(j);
    const int bi = tile / tiles_img;
    const int b = a.z_list ? a.z_list[bi] : bi;
    const int q0 = P + (tile - bi * tiles_img) * CS_TP;    // first output pixel (flat, padded coords)
    float in_scale = 1.f, in_inv = 1.f;
    if constexpr (NP == 2) h2_scale(a.amax_in[b], in_scale, in_inv);
    split_store(in_scale);      // every input pixel of the strip is split exactly once
    __syncthreads();
    if (j + 1 < nmine) fetch(tile_of(j + 1));   // in flight during the MFMA phase

    // (r3) the residual rows of this tile travel during the multiply phase instead of after it (one HBM round trip of
    // the epilogue hidden; 16 more registers live across the MFMAs)
    float* ob = a.out + (int64_t)b * img;
    const float* rb = a.R ? a.R + (int64_t)b * img : nullptr;
    float4 r4[4];
    bool okm[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      const int q = q0 + mh * 64 + mb * 16 + lr;
      const int col = q % P;
      okm[mb] = q < P + npix && col >= 1 && col <= a.Ws;
      r4[mb] = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (NP == 2) {     // (the three-term variant has no registers to spare: it loads in the epilogue)
        if (rb && okm[mb]) r4[mb] = *reinterpret_cast<const float4*>(rb + (int64_t)q * 32 + nb * 16 + lq * 4);
      }
    }

    // ---- multiply: 4 pixel blocks x 9 taps x 6 products; accumulators D[oc][pixel]; two pixel blocks
    // at a time keeps the fragment registers at 24 (the strip prefetch needs the rest) ----
    f32x4 acc[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) acc[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dh = 0; dh < 3; ++dh)
#pragma unroll
      for (int dw = 0; dw < 3; ++dw) {
        const int t = dh * 3 + dw;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          u32x4 xf[2][NP];
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            const int px = mh * 64 + (2 * g + m) * 16 + lr + dw;
            const int off = (dh * CS_SEG + px) * 64 + ((lq ^ ((px >> 1) & 3)) << 4);
#pragma unroll
            for (int p = 0; p < NP; ++p) xf[m][p] = *reinterpret_cast<const u32x4*>(smem + p * CS_PLANE + off);
          }
          // product-major, smallest terms first
#pragma unroll
          for (int tt = 0; tt < SplitTerms<NP>::N; ++tt)
#pragma unroll
            for (int m = 0; m < 2; ++m)
              acc[2 * g + m] = mfma_np<NP>(wf[t][SplitTerms<NP>::A[tt]], xf[m][SplitTerms<NP>::B[tt]], acc[2 * g + m]);
        }
      }

    // ---- epilogue: lane holds pixel lr of block mb, output channels nb*16 + 4 lq .. +3 ----
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      const int q = q0 + mh * 64 + mb * 16 + lr;
      if (okm[mb]) {
        const int64_t o = (int64_t)q * 32 + nb * 16 + lq * 4;
        f32x4 v = acc[mb];
        if constexpr (NP == 2) {   // undo the exact power-of-two operand scales
          v[0] *= in_inv * cs4.x; v[1] *= in_inv * cs4.y; v[2] *= in_inv * cs4.z; v[3] *= in_inv * cs4.w;
        }
        v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
        if (a.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (rb) {
          if constexpr (NP != 2) r4[mb] = *reinterpret_cast<const float4*>(rb + o);
          v[0] += r4[mb].x; v[1] += r4[mb].y; v[2] += r4[mb].z; v[3] += r4[mb].w;
        }
        if (a.post_relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        *reinterpret_cast<float4*>(ob + o) = make_float4(v[0], v[1], v[2], v[3]);
        out_amax = fmaxf(fmaxf(out_amax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
      }
    }
    if (a.amax) {     // per-image |max| tracker of `out` (the scale unit of the consumer's fp16 split is the window)
      track_amax(a.amax + b, out_amax);
      out_amax = 0.f;
    }
    __syncthreads();  // the strip is free for the next tile's staging
  }
}

}  // namespace

// in / out / R: zero-bordered fp32 NHWC images [B][Hs+2][Ws+2][32] (image bases, not interior pointers).
// W2h / col_scale / amax_in all given -> fp16 two-term variant, else the bf16 three-term one (W3).
int launch_conv3x3_c32_split(const float* in, const void* W3, const float* bias, const float* R, float* out, int B,
                             int Hs, int Ws, int relu, int post_relu, hipStream_t s, float* amax, const void* W2h,
                             const float* col_scale, const float* amax_in, const int* z_count, const int* z_list) {
  if (B <= 0 || Hs <= 0 || Ws <= 0) return DZN_OK;
  if (!in || !W3 || !bias || !out) return DZN_E_INVALID;
  const bool h2 = W2h && col_scale && amax_in;
  static unsigned long long attr_mask = 0;  // one bit per HIP device: function attributes are per device
  const size_t lds = (h2 ? 2 : 3) * CS_PLANE;
  if (first_use_on_device(attr_mask)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c32_split_kernel<3>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c32_split_kernel<2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  ConvArgs a{in, static_cast<const u16*>(h2 ? W2h : W3), bias, R, out, B, Hs, Ws, relu, post_relu, amax, amax_in, col_scale, z_list ? z_count : nullptr,
             z_count ? z_list : nullptr};
  const int grid = 512;   // persistent: 2 workgroups per CU; XCD-major work distribution inside the kernel
  const int pid = prof_begin(s, h2 ? "conv3x3_c32_f32h" : "conv3x3_c32_f32s", 2.0 * B * Hs * (double)Ws * 32.0 * 288.0,
                             (double)B * Hs * Ws * 32.0 * 4.0 * (R ? 3.0 : 2.0));   // image in + out (+ residual), once
  if (h2) hipLaunchKernelGGL(conv3x3_c32_split_kernel<2>, dim3(grid), dim3(256), lds, s, a);
  else hipLaunchKernelGGL(conv3x3_c32_split_kernel<3>, dim3(grid), dim3(256), lds, s, a);
  prof_end(pid, s);
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}

extern "C" int dzn_op_conv3x3_c32(const float* in, const void* W3, const float* bias, const float* R, float* out,
                                  int32_t B, int32_t Hs, int32_t Ws, int32_t relu, int32_t post_relu, void* stream) {
  return launch_conv3x3_c32_split(in, W3, bias, R, out, B, Hs, Ws, relu, post_relu,
                                  reinterpret_cast<hipStream_t>(stream), nullptr, nullptr, nullptr, nullptr);
}

// fp16 two-term variant (tests): W2h / col_scale from dzn_op_split_weights_h2 of W [32][288], amax_in f32 [B]
extern "C" int dzn_op_conv3x3_c32_h2(const float* in, const void* W3, const void* W2h, const float* col_scale,
                                     const float* amax_in, const float* bias, const float* R, float* out, int32_t B,
                                     int32_t Hs, int32_t Ws, int32_t relu, int32_t post_relu, void* stream) {
  if (!W2h || !col_scale || !amax_in) return DZN_E_INVALID;
  return launch_conv3x3_c32_split(in, W3, bias, R, out, B, Hs, Ws, relu, post_relu,
                                  reinterpret_cast<hipStream_t>(stream), nullptr, W2h, col_scale, amax_in);
}
