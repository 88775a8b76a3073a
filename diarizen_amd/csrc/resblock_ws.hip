// resblock_ws.hip — a stride-1 BasicBlock (wespeaker/resnet.py:139-144) of C = 32 or 64 planes in one kernel, with
// PRODUCER and CONSUMER wavefronts (r4):
//     out = relu( conv2( relu(conv1(x) + b1) ) + b2 + x )
// Same rolling line buffers as resblock_fused.hip (an X ring of input rows and an M ring of intermediate rows in LDS, both
// as two fp16 planes; a workgroup marches down a 60-column strip of one image; the intermediate image never reaches HBM),
// but the two convolutions run CONCURRENTLY instead of in alternating phases:
//   * wavefronts 0 .. C/16 - 1 (producers) hold conv1's weight fragments for 16 output channels each and, in step r,
//     multiply intermediate row r + 1 for all 64 pixels of the strip, then split it and store its two planes in the M ring;
//   * wavefronts C/16 .. 2 C/16 - 1 (consumers) hold conv2's fragments and, in the SAME step, multiply output row r - 1
//     from intermediate rows r - 2, r - 1, r (one step behind the producers), add bias + residual, ReLU, store — and stage
//     the x rows for the producers (fetch x row r + 3 before their multiply phase, split + store it after): the staging
//     registers fit the consumers' budget, the producers' epilogue (split of the intermediate row) is the heavier one.
// One s_barrier per row step instead of two; each wavefront keeps ONE weight set in registers (72 C/32 instead of
// 144 C/32), which is what makes C = 64 possible at all (stage 2: 18 k-blocks x 2 planes x 4 registers = 144); and while a
// producer waits for its x row, the consumers' MFMAs keep the matrix pipe busy.
//   C = 32: 4 wavefronts, 68 KB of LDS (2 workgroups per CU);  C = 64: 8 wavefronts, 152 KB (1 per CU).
// Geometry, masking, scales and arithmetic are those of resblock_fused.hip (intermediate scale from the a-priori bound
// amax(x) * max_oc sum|W1| + max|b1|); pixel rows are [pixel][k-block of 32 channels][64 B] per plane (WsLayout below).
#include "checked.h"
#include "common.h"
#include "split.h"

DZN_CHECKED_TU(resblock_ws)

namespace {

constexpr int WS_OUT = 60;                    // output columns kept per strip
constexpr int WS_PXR = 66;                    // pixels per ring row: 64 + 2 pad

struct ResBlockWsArgs {
  const float* in;
  float* out;
  const u16* W1;            // [C oc][9 C / 32 k-blocks][2 planes][32] fp16 planes (k = (dh*3 + dw) * C + ci, gemm_split.hip order)
  const u16* W2;
  const float* b1;
  const float* b2;
  const float* cs1;
  const float* cs2;
  const float* amax_in;
  float* amax_out;
  float l1max1, bmax1;
  int B, Hs, Ws;
  const int* z_count;
  const int* z_list;
};

// byte offset of (pixel, k-block, 16-B chunk) inside one plane of a ring row.
//   C = 32: 64-B pixels, chunk XOR (pixel >> 1) & 3 (the layout of conv_split.hip / resblock_fused.hip);
//   C = 64: 128-B pixels on a 144-B pitch, no XOR: 36 dwords per pixel put 16 consecutive pixels of a ds_read_b128 lane
//           group on 16 different 4-dword bank slots, and every fragment address is one per-lane base + a compile-time
//           constant (the XOR form needs a register per (block, tap, k-block): the 64-plane kernel spilled with it).
template <int C>
struct WsLayout {
  static constexpr int PITCH = C == 32 ? 64 : 144;
  static __device__ __forceinline__ int off(int px, int kb, int chunk) {
    if constexpr (C == 32) return px * 64 + ((chunk ^ ((px >> 1) & 3)) << 4);
    else return px * 144 + kb * 64 + (chunk << 4);
  }
};
template <int C>
__device__ __forceinline__ int ws_pix_off(int px, int kb, int chunk) { return WsLayout<C>::off(px, kb, chunk); }

// NP = 2: two fp16 terms per operand (three products); NP = 1 (r5, DZN_PREC_F16): the leading term only — one plane per ring,
// half the weight registers, a third of the MFMAs; reads plane 0 of the same two-plane weight buffers.
template <int C, int NP>
__global__ __launch_bounds__(2 * (C / 16) * 64, C == 32 ? 2 : 1) void resblock_ws_kernel(const ResBlockWsArgs a) {
  constexpr int KB = C / 32;                     // k-blocks of 32 channels per tap
  constexpr int NWH = C / 16;                    // wavefronts per role
  constexpr int NT = 2 * NWH * 64;
  constexpr int ROW = WS_PXR * WsLayout<C>::PITCH;   // bytes per plane of one ring row
  constexpr int PL = 4 * ROW;                    // bytes per plane of a ring (4 row slots each)
  constexpr int NKB = 9 * KB;                    // k-blocks of the whole 3x3 contraction
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sX = smem;                      // [plane][slot][66 px][KB][64 B]
  unsigned char* sM = smem + NP * PL;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave < NWH;              // wave-uniform role
  const int ocb = producer ? wave : wave - NWH;  // 16-channel output block of this wavefront
  const int lr = lane & 15, lq = lane >> 4;
  const int P = a.Ws + 2;
  const int64_t img = (int64_t)(a.Hs + 2) * P * C;
  const int nstrip = (a.Ws + WS_OUT - 1) / WS_OUT;
  const int nB = a.z_list ? a.z_count[0] : a.B;
  const int nitem = nB * nstrip;

  // this wavefront's weight fragments (conv1 for producers, conv2 for consumers), bias, inverse row scales
  u32x4 wf[NKB][NP];
  {
    const u16* Wp = (producer ? a.W1 : a.W2) + (int64_t)(ocb * 16 + lr) * (NKB * 2 * 32) + lq * 8;
#pragma unroll
    for (int t = 0; t < NKB; ++t)
#pragma unroll
      for (int p = 0; p < NP; ++p) wf[t][p] = *reinterpret_cast<const u32x4*>(Wp + t * 64 + p * 32);
  }
  const float4 bv = *reinterpret_cast<const float4*>((producer ? a.b1 : a.b2) + ocb * 16 + lq * 4);
  const float4 cv = *reinterpret_cast<const float4*>((producer ? a.cs1 : a.cs2) + ocb * 16 + lq * 4);

  for (int i = tid; i < 2 * NP * PL / 16; i += NT) reinterpret_cast<float4*>(smem)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();

  // x staging (consumers): item = ptid + NWH * 64 * i, i < 2 -> (pixel = item / (4 KB), k-block, chunk c): channels
  // 32 kb + 4c .. + 3 and 32 kb + 16 + 4c .. + 3 of one pixel
  const int ptid = tid - NWH * 64;                // the CONSUMER wavefronts stage the x rows (threads NWH * 64 ..): their
                                                  // register budget has the room (the producers' split / store of the
                                                  // intermediate row is the heavier epilogue)
  float4 xu[2], xv[2];

  for (int item = blockIdx.x; item < nitem; item += gridDim.x) {
    const int bi = item / nstrip, strip = item - bi * nstrip;
    const int b = a.z_list ? a.z_list[bi] : bi;
    const int c0 = strip * WS_OUT;
    const float* ib = a.in + (int64_t)b * img;
    float* ob = a.out + (int64_t)b * img;
    float xs, xinv, ms, minv;
    DZN_CHECK(b >= 0 && b < a.B, 0x502, b);                                                   // image index inside the batch
    h2_scale(a.amax_in[b], xs, xinv);
    h2_scale(fmaf(a.amax_in[b], a.l1max1, a.bmax1), ms, minv);
    float out_amax = 0.f;

    auto fetch_row = [&](int q) {                 // x row q in [-1, Hs]: padded row q + 1, padded columns c0 - 1 + pixel (clamped)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int it = ptid + NWH * 64 * i;
        const int px = it / (4 * KB), kb = (it / 4) % KB, c = it & 3;
        int pcol = c0 - 1 + px;
        pcol = pcol < 0 ? 0 : (pcol > P - 1 ? P - 1 : pcol);
        const float* s = ib + ((int64_t)(q + 1) * P + pcol) * C + kb * 32 + 4 * c;
        xu[i] = *reinterpret_cast<const float4*>(s);
        xv[i] = *reinterpret_cast<const float4*>(s + 16);
      }
    };
    auto store_row = [&](int q) {
      unsigned char* dst = sX + ((q + 1) & 3) * ROW;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int it = ptid + NWH * 64 * i;
        const int px = it / (4 * KB), kb = (it / 4) % KB, c = it & 3;
        u32x4 pf[NP];
        split_np<NP>((f32x4){xu[i].x, xu[i].y, xu[i].z, xu[i].w}, (f32x4){xv[i].x, xv[i].y, xv[i].z, xv[i].w}, xs, pf);
        const int off = ws_pix_off<C>(px, kb, c);
        *reinterpret_cast<u32x4*>(dst + off) = pf[0];
        if constexpr (NP == 2) *reinterpret_cast<u32x4*>(dst + PL + off) = pf[1];
      }
    };
    // 4 pixel blocks x 9 taps x KB k-blocks against three ring rows
    auto conv = [&](const unsigned char* ring, const int (&rowoff)[3], f32x4 (&acc)[4]) {
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dh = 0; dh < 3; ++dh)
#pragma unroll
        for (int dw = 0; dw < 3; ++dw)
#pragma unroll
          for (int kb = 0; kb < KB; ++kb) {
            const int t = (dh * 3 + dw) * KB + kb;
#pragma unroll
            for (int g = 0; g < 2; ++g) {          // two pixel blocks at a time: 16 fragment registers live
              u32x4 xf[2][NP];
#pragma unroll
              for (int m = 0; m < 2; ++m) {
                const int px = (2 * g + m) * 16 + lr + dw;
                const int off = rowoff[dh] + ws_pix_off<C>(px, kb, lq);
                DZN_CHECK(off >= 0 && off + 16 <= PL, 0x501, off);                               // fragment read inside its ring plane
                xf[m][0] = *reinterpret_cast<const u32x4*>(ring + off);
                if constexpr (NP == 2) xf[m][1] = *reinterpret_cast<const u32x4*>(ring + PL + off);
              }
#pragma unroll
              for (int tt = 0; tt < SplitTerms<NP>::N; ++tt)
#pragma unroll
                for (int m = 0; m < 2; ++m)
                  acc[2 * g + m] = mfma_np<NP>(wf[t][SplitTerms<NP>::A[tt]], xf[m][SplitTerms<NP>::B[tt]], acc[2 * g + m]);
            }
          }
    };

    // ---- prologue: x rows -1, 0, 1; the slot of intermediate row -1 (the border) is zero ----
    __syncthreads();                               // the previous item is done with both rings
    if (!producer) {
#pragma unroll
      for (int q = -1; q <= 1; ++q) {
        fetch_row(q);
        store_row(q);
      }
    } else {
      for (int i = tid; i < NP * ROW / 16; i += NWH * 64) {
        const int p = i / (ROW / 16), j = i - p * (ROW / 16);
        reinterpret_cast<float4*>(sM + p * PL)[j] = make_float4(0.f, 0.f, 0.f, 0.f);     // slot 0 = row -1
      }
    }
    __syncthreads();

    // step r: producers -> intermediate row r + 1 (M slot (r + 2) & 3) and x row r + 3 (X slot r & 3);
    //         consumers -> output row r - 1 from intermediate rows r - 2, r - 1, r (slots (r - 1) & 3, r & 3, (r + 1) & 3)
    for (int r = -1; r <= a.Hs; ++r) {
      if (producer) {
        const int mrow = r + 1;
        if (mrow <= a.Hs) {
          f32x4 acc[4];
#pragma unroll
          for (int m = 0; m < 4; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (mrow < a.Hs) {
            const int rowoff[3] = {((r + 1) & 3) * ROW, ((r + 2) & 3) * ROW, ((r + 3) & 3) * ROW};   // x rows r, r + 1, r + 2
            conv(sX, rowoff, acc);
          }
          unsigned char* mdst = sM + ((mrow + 1) & 3) * ROW;
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const int j = m * 16 + lr;               // intermediate pixel of the strip = image column c0 - 1 + j
            const int col = c0 - 1 + j;
            const bool ok = mrow < a.Hs && col >= 0 && col < a.Ws;
            float v[4];
            v[0] = fmaxf(fmaf(acc[m][0], xinv * cv.x, bv.x), 0.f);
            v[1] = fmaxf(fmaf(acc[m][1], xinv * cv.y, bv.y), 0.f);
            v[2] = fmaxf(fmaf(acc[m][2], xinv * cv.z, bv.z), 0.f);
            v[3] = fmaxf(fmaf(acc[m][3], xinv * cv.w, bv.w), 0.f);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] * ms : 0.f;
            // channels 16 ocb + 4 lq + e: k-block ocb / 2, chunk lq, bytes 8 (ocb % 2) .. + 7 of the chunk
            f32x2 x01 = {v[0], v[1]}, x23 = {v[2], v[3]};
            const f16x2 h01 = __builtin_convertvector(x01, f16x2), h23 = __builtin_convertvector(x23, f16x2);
            if constexpr (NP == 2) {
              const f32x2 f01 = __builtin_convertvector(h01, f32x2), f23 = __builtin_convertvector(h23, f32x2);
              f32x2 r01 = {x01[0] - f01[0], x01[1] - f01[1]}, r23 = {x23[0] - f23[0], x23[1] - f23[1]};
              const f16x2 l01 = __builtin_convertvector(r01, f16x2), l23 = __builtin_convertvector(r23, f16x2);
              const int off = ws_pix_off<C>(j, ocb >> 1, lq) + (ocb & 1) * 8;
              *reinterpret_cast<uint2*>(mdst + off) = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
              *reinterpret_cast<uint2*>(mdst + PL + off) = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
            } else {
              const int off = ws_pix_off<C>(j, ocb >> 1, lq) + (ocb & 1) * 8;
              *reinterpret_cast<uint2*>(mdst + off) = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
            }
          }
        }
      } else {
        // x row r + 3 for the producers' step r + 1 (X slot r & 3: row r - 1, last read in step r - 1): in flight during the
        // multiply phase below
        const bool has_next = r + 3 <= a.Hs;
        if (has_next) fetch_row(r + 3);
        const int orow_i = r - 1;                     // output row of this step
        if (orow_i >= 0 && orow_i < a.Hs) {
          const int64_t orow = (int64_t)(orow_i + 1) * P * C;
          const int rowoff[3] = {((r - 1) & 3) * ROW, (r & 3) * ROW, ((r + 1) & 3) * ROW};   // intermediate rows r - 2, r - 1, r
          f32x4 oc[4];
          conv(sM, rowoff, oc);
          if (has_next) store_row(r + 3);
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const int j = m * 16 + lr;
            const int col = c0 + j;
            if (j < WS_OUT && col < a.Ws) {
              // the residual row is an L2 hit (this workgroup staged it four steps ago); loaded after the multiply phase so
              // that it does not hold 16 registers across it
              const float4 rr = *reinterpret_cast<const float4*>(ib + orow + (int64_t)(col + 1) * C + ocb * 16 + lq * 4);
              float4 v;
              v.x = fmaxf(fmaf(oc[m][0], minv * cv.x, bv.x) + rr.x, 0.f);
              v.y = fmaxf(fmaf(oc[m][1], minv * cv.y, bv.y) + rr.y, 0.f);
              v.z = fmaxf(fmaf(oc[m][2], minv * cv.z, bv.z) + rr.z, 0.f);
              v.w = fmaxf(fmaf(oc[m][3], minv * cv.w, bv.w) + rr.w, 0.f);
              *reinterpret_cast<float4*>(ob + orow + (int64_t)(col + 1) * C + ocb * 16 + lq * 4) = v;
              out_amax = fmaxf(fmaxf(out_amax, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
            }
          }
        } else if (has_next) {
          store_row(r + 3);
        }
      }
      __syncthreads();
    }
    if (!producer && a.amax_out) track_amax(a.amax_out + b, out_amax);
  }
}

template <int C, int NP>
int launch_ws(const ResBlockWsArgs& a, int B, int Ws, hipStream_t s) {
  constexpr int KB = C / 32;
  const size_t lds = (size_t)2 * NP * 4 * WS_PXR * WsLayout<C>::PITCH;
  static unsigned long long attr_mask = 0;
  if (first_use_on_device(attr_mask))
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(resblock_ws_kernel<C, NP>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
  const int nstrip = (Ws + WS_OUT - 1) / WS_OUT;
  const int64_t nitem = (int64_t)B * nstrip;
  const int slots = C == 32 ? 512 : 256;                 // persistent: 2 / 1 workgroups per CU
  const int grid = (int)(nitem < slots ? nitem : slots);
  hipLaunchKernelGGL((resblock_ws_kernel<C, NP>), dim3(grid), dim3(2 * (C / 16) * 64), lds, s, a);
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}

}  // namespace

// in / out: zero-bordered fp32 NHWC images [B][Hs+2][Ws+2][C] (image bases), C = 32 or 64; W1 / W2: fp16 two-term planes of the
// folded [C][9 C] weights (dzn_op_split_weights_h2, k = (dh*3 + dw) * C + ci), cs = inverse row scales, b = folded BN shifts.
int launch_resblock_ws(const float* in, float* out, const void* W1, const float* cs1, const float* b1, const void* W2,
                       const float* cs2, const float* b2, const float* amax_in, float* amax_out, float l1max1, float bmax1,
                       int B, int Hs, int Ws, int C, hipStream_t s, const int* z_count, const int* z_list, int np) {
  if (B <= 0 || Hs <= 0 || Ws <= 0) return DZN_OK;
  if (!in || !out || !W1 || !W2 || !cs1 || !cs2 || !b1 || !b2 || !amax_in || (C != 32 && C != 64) || (np != 1 && np != 2)) return DZN_E_INVALID;
  ResBlockWsArgs a{in, out, static_cast<const u16*>(W1), static_cast<const u16*>(W2), b1, b2, cs1, cs2, amax_in, amax_out,
                   l1max1, bmax1, B, Hs, Ws, z_list ? z_count : nullptr, z_count ? z_list : nullptr};
  const int pid = prof_begin(s, np == 2 ? (C == 32 ? "resblock32_ws_f32h" : "resblock64_ws_f32h") : (C == 32 ? "resblock32_ws_f16" : "resblock64_ws_f16"),
                             2.0 * 2.0 * B * Hs * (double)Ws * C * (9.0 * C), (double)B * Hs * Ws * C * 4.0 * 2.0);
  const int rc = np == 2 ? (C == 32 ? launch_ws<32, 2>(a, B, Ws, s) : launch_ws<64, 2>(a, B, Ws, s))
                         : (C == 32 ? launch_ws<32, 1>(a, B, Ws, s) : launch_ws<64, 1>(a, B, Ws, s));
  prof_end(pid, s);
  return rc;
}

extern "C" int dzn_op_resblock_ws(const float* in, float* out, const void* W1, const float* cs1, const float* b1,
                                  const void* W2, const float* cs2, const float* b2, const float* amax_in, float l1max1,
                                  float bmax1, int32_t B, int32_t Hs, int32_t Ws, int32_t C, void* stream) {
  return launch_resblock_ws(in, out, W1, cs1, b1, W2, cs2, b2, amax_in, nullptr, l1max1, bmax1, B, Hs, Ws, C,
                            reinterpret_cast<hipStream_t>(stream), nullptr, nullptr, op_resblock_np());
}
