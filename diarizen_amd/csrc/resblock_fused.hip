// resblock_fused.hip — one BasicBlock of the first ResNet34 stage in ONE kernel (r4):
//     out = relu( conv2( relu(conv1(x) + b1) ) + b2 + x )        wespeaker/resnet.py:139-144, 32 planes, stride 1,
// BatchNorm folded into the conv weights / biases, fp16 two-term arithmetic (DZN_PREC_F32_H2 / DZN_PREC_F16).
//
// Why.  conv3x3_c32_split_kernel (conv_split.hip) runs the stage's six convolutions one launch each: 8.6 % of the
// 30-min step at 162 TFLOP/s with `MfmaUtil` ~22 %.  It is bound by bytes, not by the matrix pipe: per launch it moves
// 11.9 GB through the fabric (PMC, x2 correction calibrated in profiles/r4_fetch_size_calibration.txt) for 7.6 GB
// algorithmic — every input row is fetched about twice for the 3-row halo of the flat 128-pixel tiles — at 4.4 TB/s of a
// ~6.2 TB/s HBM stream (scripts/ubench/lds_fill_rate.hip), ~80 KB per 128 output pixels.  The intermediate image of a block (3.1 GB at
// 374 windows) is written by conv1 and read back — twice — by conv2, and conv2 re-reads x as its residual.
//
// Here a workgroup owns a COLUMN STRIP of one image and marches down its rows with two rolling line buffers in LDS:
//     X ring : 4 rows of the input strip  (64 pixels, two fp16 planes, split once when the row arrives)
//     M ring : 3 rows of the intermediate (64 pixels, two fp16 planes, split once when the row is produced)
// Step r computes intermediate row r + 1 from x rows r, r + 1, r + 2 (conv1 + b1 + ReLU, stays in LDS) and then output
// row r from intermediate rows r - 1, r, r + 1 (conv2 + b2 + residual + ReLU -> HBM).  Per block and strip row the
// kernel reads one x row (8 KB) + the residual row again (8 KB, an L2 hit: the same workgroup fetched it three steps
// earlier) and writes one output row: ~24 KB per 60 output pixels for BOTH convolutions, where the per-conv kernel moves
// 2 x ~40 KB.  The intermediate never exists in HBM.  Cost: the strip overlaps its neighbours by 4 columns (60 of every
// 64 computed pixels are kept: 6.7 % more MFMA work), and the intermediate row is computed once per strip only.
//
// Geometry.  Images are zero-bordered NHWC fp32 [B][H + 2][W + 2][32].  Strip s of an image: output columns
// [60 s, 60 s + 60); intermediate pixel j of the strip = image column 60 s - 1 + j (j < 64; 62 are needed); x pixel i =
// image column 60 s - 2 + i (i < 64).  Pixels that fall outside the image are the zero padding of the NEXT conv, so
// intermediate values there are forced to zero (columns outside [0, W), rows -1 and H).  Four wavefronts: wavefront
// (mh, nb) multiplies pixel blocks 2 mh, 2 mh + 1 (16 pixels each) by output channels 16 nb .. 16 nb + 15; per phase
// 2 blocks x 9 taps x NP-products MFMAs.  Both convolutions' weight fragments live in registers (2 x 9 x NP x 4).
//
// Arithmetic.  conv1 is the per-conv kernel's arithmetic exactly (same planes, same product order), so the intermediate
// fp32 values are bit-identical to what conv3x3_c32_split_kernel would store.  What differs is the power-of-two scale
// of the intermediate's fp16 split: the per-conv pipeline takes it from the |max| tracker of the finished intermediate
// image; here the image is never finished before it is consumed, so the scale comes from the a-priori bound
//     |relu(conv1(x) + b1)| <= amax(x) * max_oc sum_k |W1[oc][k]| + max_oc |b1[oc]|
// (host: engine.cpp make_resconv).  A looser scale does not cost the two-term split its 22 bits — only elements more
// than 2^17 below the BOUND (instead of below the true maximum) get a subnormal lo term — so the result differs from the
// unfused pipeline in the last bits only (tests/test_ops_gpu.py::test_resblock32_fused_equals_two_convs: <= 2e-6 of the
// block's |max|; the model-level embedding tests run through it).
#include "checked.h"
#include "common.h"
#include "split.h"

DZN_CHECKED_TU(resblock_fused)

namespace {

constexpr int RB_PX = 64;                    // pixels computed per row and phase (4 blocks of 16)
constexpr int RB_OUT = 60;                   // output columns kept per strip
constexpr int RB_ROW = 66 * 64;              // bytes per plane of one ring row: 64 pixels + 2 pad pixels, 64 B each
constexpr int RB_XROWS = 4, RB_MROWS = 3;

struct ResBlockArgs {
  const float* in;          // x (image bases)
  float* out;
  const u16* W1;            // [32 oc][9 taps][NP planes][32] fp16 planes of conv1 (k order of gemm_split.hip), row-scaled
  const u16* W2;
  const float* b1;          // [32] folded BN shifts
  const float* b2;
  const float* cs1;         // [32] inverse weight row scales
  const float* cs2;
  const float* amax_in;     // per-image |max| of x
  float* amax_out;          // per-image |max| tracker of out, or nullptr
  float l1max1, bmax1;      // max_oc sum |W1[oc]|, max |b1|: bound of the intermediate
  int B, Hs, Ws;
  const int* z_count;       // device-chosen subset of the B images (or nullptr)
  const int* z_list;
};

template <int NP>
__global__ __launch_bounds__(256, 2) void resblock32_fused_kernel(const ResBlockArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int XPL = RB_XROWS * RB_ROW;         // bytes per plane of the X ring
  constexpr int MPL = RB_MROWS * RB_ROW;
  unsigned char* sX = smem;                      // [plane][row slot][66 px][64 B]
  unsigned char* sM = smem + NP * XPL;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int mh = wave >> 1, nb = wave & 1;
  const int P = a.Ws + 2;
  const int64_t img = (int64_t)(a.Hs + 2) * P * 32;
  const int nstrip = (a.Ws + RB_OUT - 1) / RB_OUT;
  const int nB = a.z_list ? a.z_count[0] : a.B;
  const int nitem = nB * nstrip;

  // weight fragments of this wavefront's 16 output channels, both convolutions
  u32x4 wf1[9][NP], wf2[9][NP];
  {
    constexpr int SP = NP == 3 ? 3 : 2;          // planes STORED per weight row (NP = 1 reads the leading one of two)
    const int64_t wo = (int64_t)(nb * 16 + lr) * (9 * SP * 32) + lq * 8;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        wf1[t][p] = *reinterpret_cast<const u32x4*>(a.W1 + wo + t * SP * 32 + p * 32);
        wf2[t][p] = *reinterpret_cast<const u32x4*>(a.W2 + wo + t * SP * 32 + p * 32);
      }
  }
  const float4 b1v = *reinterpret_cast<const float4*>(a.b1 + nb * 16 + lq * 4);
  const float4 b2v = *reinterpret_cast<const float4*>(a.b2 + nb * 16 + lq * 4);
  const float4 c1v = *reinterpret_cast<const float4*>(a.cs1 + nb * 16 + lq * 4);
  const float4 c2v = *reinterpret_cast<const float4*>(a.cs2 + nb * 16 + lq * 4);

  // the M ring's pad pixels (64, 65 of every row) and the whole ring start as zeros; rows are rewritten in full
  for (int i = tid; i < NP * MPL / 16; i += 256) reinterpret_cast<float4*>(sM)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = tid; i < NP * XPL / 16; i += 256) reinterpret_cast<float4*>(sX)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();

  // staging: thread -> (pixel = tid / 4, chunk c = tid % 4): channels 4c..4c+3 and 16+4c..16+4c+3 of one x pixel
  const int spx = tid >> 2, sc = tid & 3;
  const int soff = spx * 64 + ((sc ^ ((spx >> 1) & 3)) << 4);
  float4 xu, xv;

  for (int item = blockIdx.x; item < nitem; item += gridDim.x) {
    const int bi = item / nstrip, strip = item - bi * nstrip;
    const int b = a.z_list ? a.z_list[bi] : bi;
    const int c0 = strip * RB_OUT;                       // first output column (image coordinates)
    const float* ib = a.in + (int64_t)b * img;
    float* ob = a.out + (int64_t)b * img;
    float xs, xinv, ms, minv;
    DZN_CHECK(b >= 0 && b < a.B, 0x403, b);                                               // image index (device-chosen subset) inside the batch
    h2_scale(a.amax_in[b], xs, xinv);
    h2_scale(fmaf(a.amax_in[b], a.l1max1, a.bmax1), ms, minv);
    float out_amax = 0.f;

    // x pixel i of image row q: padded row q + 1, padded column c0 - 1 + i (clamped: clamped pixels only feed masked results)
    int pcol = c0 - 1 + spx;
    pcol = pcol < 0 ? 0 : (pcol > P - 1 ? P - 1 : pcol);
    const float* xsrc = ib + (int64_t)pcol * 32 + 4 * sc;
    auto fetch_row = [&](int q) {                          // q in [-1, Hs]
      const float* s = xsrc + (int64_t)(q + 1) * P * 32;
      xu = *reinterpret_cast<const float4*>(s);
      xv = *reinterpret_cast<const float4*>(s + 16);
    };
    auto store_row = [&](int q) {
      u32x4 pf[NP];
      split_np<NP>((f32x4){xu.x, xu.y, xu.z, xu.w}, (f32x4){xv.x, xv.y, xv.z, xv.w}, xs, pf);
      unsigned char* dst = sX + ((q + 1) & 3) * RB_ROW + soff;
      DZN_CHECK(soff + 16 <= RB_ROW && q >= -1 && q <= a.Hs, 0x401, q);                  // staged pixel inside its ring row, row inside the padded image
#pragma unroll
      for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4*>(dst + p * XPL) = pf[p];
    };
    // one phase: 2 pixel blocks x 9 taps against the three ring rows `rows[dh]` (byte offsets of the row slots)
    auto conv = [&](const unsigned char* ring, int plane_bytes, const int (&rowoff)[3], const u32x4 (&wf)[9][NP],
                    f32x4 (&acc)[2]) {
      acc[0] = acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dh = 0; dh < 3; ++dh)
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) {
          const int t = dh * 3 + dw;
          u32x4 xf[2][NP];
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            const int px = (2 * mh + m) * 16 + lr + dw;
            const int off = rowoff[dh] + px * 64 + ((lq ^ ((px >> 1) & 3)) << 4);
            DZN_CHECK(off >= 0 && off + 16 <= plane_bytes, 0x402, off);                      // fragment read inside its ring plane
#pragma unroll
            for (int p = 0; p < NP; ++p) xf[m][p] = *reinterpret_cast<const u32x4*>(ring + p * plane_bytes + off);
          }
#pragma unroll
          for (int tt = 0; tt < SplitTerms<NP>::N; ++tt)
#pragma unroll
            for (int m = 0; m < 2; ++m)
              acc[m] = mfma_np<NP>(wf[t][SplitTerms<NP>::A[tt]], xf[m][SplitTerms<NP>::B[tt]], acc[m]);
        }
    };

    // ---- prologue: x rows -1, 0, 1 in the ring; intermediate row -1 (the border) is zero ----
    __syncthreads();                                       // the previous item's last phase is done with both rings
#pragma unroll
    for (int q = -1; q <= 1; ++q) {
      fetch_row(q);
      store_row(q);
    }
    {
      // zero the slot of intermediate row -1 (slot 0); rows 0.. are written before they are read
      for (int i = tid; i < NP * RB_ROW / 16; i += 256) {
        const int p = i / (RB_ROW / 16), j = i - p * (RB_ROW / 16);
        reinterpret_cast<float4*>(sM + p * MPL)[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    __syncthreads();

    for (int r = -1; r < a.Hs; ++r) {
      const bool has_next = r + 3 <= a.Hs;                 // x row r + 3 exists (rows up to Hs = the bottom border)
      if (has_next) fetch_row(r + 3);                      // in flight during phase A
      // ---- phase A: intermediate row r + 1 (rows outside the image are zero, not computed) ----
      const int mrow = r + 1;
      f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
      if (mrow < a.Hs) {
        const int rowoff[3] = {((r + 1) & 3) * RB_ROW, ((r + 2) & 3) * RB_ROW, ((r + 3) & 3) * RB_ROW};   // x rows r, r+1, r+2
        conv(sX, XPL, rowoff, wf1, acc);
      }
      __syncthreads();                                     // every wavefront is done with phase B of step r - 1 (M ring) and
                                                           // with phase A of this step (the X slot of row r - 1 is free)
      {
        unsigned char* mdst = sM + ((mrow + 1) % 3) * RB_ROW;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const int j = (2 * mh + m) * 16 + lr;            // intermediate pixel of the strip
          const int col = c0 - 1 + j;                       // image column
          const bool ok = mrow < a.Hs && col >= 0 && col < a.Ws;
          float v[4];
          v[0] = fmaxf(fmaf(acc[m][0], xinv * c1v.x, b1v.x), 0.f);
          v[1] = fmaxf(fmaf(acc[m][1], xinv * c1v.y, b1v.y), 0.f);
          v[2] = fmaxf(fmaf(acc[m][2], xinv * c1v.z, b1v.z), 0.f);
          v[3] = fmaxf(fmaf(acc[m][3], xinv * c1v.w, b1v.w), 0.f);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] * ms : 0.f;
          // channels 16 nb + 4 lq + e -> chunk lq of the pixel, bytes 8 nb .. 8 nb + 7 of the chunk, per plane
          f32x2 x01 = {v[0], v[1]}, x23 = {v[2], v[3]};
          const f16x2 h01 = __builtin_convertvector(x01, f16x2), h23 = __builtin_convertvector(x23, f16x2);
          const int off = j * 64 + ((lq ^ ((j >> 1) & 3)) << 4) + nb * 8;
          *reinterpret_cast<uint2*>(mdst + off) = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
          if constexpr (NP == 2) {
            const f32x2 f01 = __builtin_convertvector(h01, f32x2), f23 = __builtin_convertvector(h23, f32x2);
            f32x2 r01 = {x01[0] - f01[0], x01[1] - f01[1]}, r23 = {x23[0] - f23[0], x23[1] - f23[1]};
            const f16x2 l01 = __builtin_convertvector(r01, f16x2), l23 = __builtin_convertvector(r23, f16x2);
            *reinterpret_cast<uint2*>(mdst + MPL + off) =
                make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
          }
        }
      }
      if (has_next) store_row(r + 3);                      // slot (r + 4) & 3 == slot of x row r - 1: free
      // residual row r for phase B travels during its multiply phase
      float4 r4[2];
      bool okp[2];
      const int64_t orow = (int64_t)(r + 1) * P * 32;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int j = (2 * mh + m) * 16 + lr;
        const int col = c0 + j;
        okp[m] = r >= 0 && j < RB_OUT && col < a.Ws;
        r4[m] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (okp[m]) r4[m] = *reinterpret_cast<const float4*>(ib + orow + (int64_t)(col + 1) * 32 + nb * 16 + lq * 4);
      }
      __syncthreads();                                     // intermediate row r + 1 and x row r + 3 are in LDS
      // ---- phase B: output row r from intermediate rows r - 1, r, r + 1 ----
      if (r >= 0) {
        const int rowoff[3] = {(r % 3) * RB_ROW, ((r + 1) % 3) * RB_ROW, ((r + 2) % 3) * RB_ROW};
        f32x4 oc[2];
        conv(sM, MPL, rowoff, wf2, oc);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          if (okp[m]) {
            const int col = c0 + (2 * mh + m) * 16 + lr;
            float4 v;
            v.x = fmaxf(fmaf(oc[m][0], minv * c2v.x, b2v.x) + r4[m].x, 0.f);
            v.y = fmaxf(fmaf(oc[m][1], minv * c2v.y, b2v.y) + r4[m].y, 0.f);
            v.z = fmaxf(fmaf(oc[m][2], minv * c2v.z, b2v.z) + r4[m].z, 0.f);
            v.w = fmaxf(fmaf(oc[m][3], minv * c2v.w, b2v.w) + r4[m].w, 0.f);
            *reinterpret_cast<float4*>(ob + orow + (int64_t)(col + 1) * 32 + nb * 16 + lq * 4) = v;
            out_amax = fmaxf(fmaxf(out_amax, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));   // post-ReLU: non-negative
          }
        }
      }
    }
    if (a.amax_out) track_amax(a.amax_out + b, out_amax);
  }
}

}  // namespace

// in / out: zero-bordered fp32 NHWC images [B][Hs+2][Ws+2][32] (image bases); W1 / W2: the fp16 planes of the two
// folded 3x3 convolutions (dzn_op_split_weights_h2 of [32][288]), cs = their inverse row scales, b = folded BN shifts.
// np = 2 (two-term, DZN_PREC_F32_H2) or 1 (leading term only, DZN_PREC_F16: reads plane 0 of the same two-plane buffers).
int launch_resblock32_fused(const float* in, float* out, const void* W1, const float* cs1, const float* b1, const void* W2,
                            const float* cs2, const float* b2, const float* amax_in, float* amax_out, float l1max1,
                            float bmax1, int B, int Hs, int Ws, int np, hipStream_t s, const int* z_count,
                            const int* z_list) {
  if (B <= 0 || Hs <= 0 || Ws <= 0) return DZN_OK;
  if (!in || !out || !W1 || !W2 || !cs1 || !cs2 || !b1 || !b2 || !amax_in || (np != 2 && np != 1)) return DZN_E_INVALID;
  static unsigned long long attr_mask = 0;
  const size_t lds = (size_t)np * (RB_XROWS + RB_MROWS) * RB_ROW;
  if (first_use_on_device(attr_mask)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(resblock32_fused_kernel<2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(resblock32_fused_kernel<1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  ResBlockArgs a{in, out, static_cast<const u16*>(W1), static_cast<const u16*>(W2), b1, b2, cs1, cs2, amax_in, amax_out,
                 l1max1, bmax1, B, Hs, Ws, z_list ? z_count : nullptr, z_count ? z_list : nullptr};
  const int nstrip = (Ws + RB_OUT - 1) / RB_OUT;
  const int64_t nitem = (int64_t)B * nstrip;
  const int grid = (int)(nitem < 512 ? nitem : 512);      // persistent: 2 workgroups per CU
  // algorithmic work of the two convolutions; bytes: x in + out, once each (the residual is x)
  const int pid = prof_begin(s, np == 2 ? "resblock32_fused_f32h" : "resblock32_fused_f16", 2.0 * 2.0 * B * Hs * (double)Ws * 32.0 * 288.0,
                             (double)B * Hs * Ws * 32.0 * 4.0 * 2.0);
  if (np == 2) hipLaunchKernelGGL(resblock32_fused_kernel<2>, dim3(grid), dim3(256), lds, s, a);
  else hipLaunchKernelGGL(resblock32_fused_kernel<1>, dim3(grid), dim3(256), lds, s, a);
  prof_end(pid, s);
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}

// terms per operand the kernel-level entry points below (and dzn_op_resblock_ws) run with: 2 (default) or 1 (the DZN_PREC_F16
// form) — a test knob, set by dzn_op_set_resblock_np
static int g_op_resblock_np = 2;
int op_resblock_np() { return g_op_resblock_np; }
extern "C" int dzn_op_set_resblock_np(int32_t np) {
  if (np != 1 && np != 2) return DZN_E_INVALID;
  g_op_resblock_np = np;
  return DZN_OK;
}

// kernel-level entry point (tests): amax_in f32 [B] per-image |max| of `in`; l1max1 / bmax1 as computed by the caller
extern "C" int dzn_op_resblock32_fused(const float* in, float* out, const void* W1, const float* cs1, const float* b1,
                                       const void* W2, const float* cs2, const float* b2, const float* amax_in,
                                       float l1max1, float bmax1, int32_t B, int32_t Hs, int32_t Ws, void* stream) {
  return launch_resblock32_fused(in, out, W1, cs1, b1, W2, cs2, b2, amax_in, nullptr, l1max1, bmax1, B, Hs, Ws, g_op_resblock_np,
                                 reinterpret_cast<hipStream_t>(stream), nullptr, nullptr);
}
