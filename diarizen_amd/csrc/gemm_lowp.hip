#ifdef DZN_TUNING   // the bf16 engine mode is quarantined: compiled only by DZN_TUNING=1 builds
// gemm_lowp.hip — bf16 MFMA contraction with BOTH operands bf16 in HBM (the "bf16" engine mode).
//
// Same contract as gemm.hip (dzn_gemm_desc: two-level K addressing, row-offset tables, z batching,
// fused epilogue) with A = bf16 activations written by the producing kernel's epilogue, W = bf16
// weights packed at load, fp32 accumulation (v_mfma_f32_16x16x32_bf16), output fp32 or bf16
// (c_bf16), residual fp32 or bf16 (r_bf16).  Because no conversion sits between HBM and LDS, both
// tiles arrive by LDS-DMA (global_load_lds_dwordx4): a K tile is 64 bf16 = one 128-B LDS row, so
// the LDS geometry, the source-side XOR swizzle and the fragment reads are those of the f32 kernel.
// A K tile may straddle a kc chunk or the end of K only at its 32-element half (K % 32 == 0,
// kc % 32 == 0); a half that lies beyond K is fetched from a zero page.
#include <cstdio>
#include <cstdlib>

#include "common.h"

namespace {

__device__ __attribute__((aligned(16))) unsigned int g_zero_page[64];  // zero-initialised

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
  return __uint_as_float(((unsigned int)b) << 16);
}
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
  __bf16 h = (__bf16)f;
  return *reinterpret_cast<unsigned short*>(&h);
}

template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(256) void gemm_lowp_kernel(const dzn_gemm_desc d) {
  static_assert(WGM * WGN == 4, "4 wavefronts per workgroup");
  constexpr int BK = 64;
  constexpr int TM = BM / WGM, TN = BN / WGN;
  constexpr int MI = TM / 16, NI = TN / 16;
  constexpr int ACH = BM / 32, WCH = BN / 32;
  constexpr int BUF = (BM + BN) * 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int tilesN = (d.N + BN - 1) / BN;
  int t;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = t / tilesN, tn = t % tilesN;
  const int z = blockIdx.y;
  int z0 = z / d.zdiv;
  const int z1 = z - z0 * d.zdiv;
  if (d.z_list) {   // device-chosen subset of the batch (dzn_gemm_desc.z_count / z_list)
    if (z0 >= d.z_count[0]) return;
    z0 = d.z_list[z0];
  }
  const u16* __restrict__ A = reinterpret_cast<const u16*>(d.A) + z0 * d.a_z0 + z1 * d.a_z1;
  const u16* __restrict__ W = reinterpret_cast<const u16*>(d.W16) + z0 * d.w_z0 + z1 * d.w_z1;
  const int64_t cz = z0 * d.c_z0 + z1 * d.c_z1;
  const int64_t bz = z0 * d.b_z0 + z1 * d.b_z1;

  // thread -> (row = tid/8 + 32 i, physical slot p = tid%8); logical 8-element chunk c = p ^ swz(row)
  const int r0 = tid >> 3;
  const int csw = (tid & 7) ^ ((r0 >> 1) & 7);
  const int half = csw >> 2;          // which 32-element half of the 64-wide K tile
  const int cin = (csw & 3) * 8;      // element offset inside the half
  int64_t abase[ACH], wbase[WCH];
#pragma unroll
  for (int i = 0; i < ACH; ++i) {
    int m = tm * BM + r0 + 32 * i;
    m = m < d.M ? m : d.M - 1;
    abase[i] = (d.a_rowoff ? (int64_t)d.a_rowoff[m] : (int64_t)m * d.lda) + cin;
  }
#pragma unroll
  for (int i = 0; i < WCH; ++i) {
    int n = tn * BN + r0 + 32 * i;
    n = n < d.N ? n : d.N - 1;
    wbase[i] = (int64_t)n * d.ldw + cin;
  }

  auto issue = [&](int k0, int buf) {
    // the two 32-element halves of this K tile: chunk offsets computed on the scalar unit
    const int ka = k0, kb = k0 + 32;
    const int cha = ka / d.kc, chb = kb / d.kc;
    const int64_t offa = (int64_t)cha * d.ldk + (ka - cha * d.kc);
    const int64_t offb = (int64_t)chb * d.ldk + (kb - chb * d.kc);
    const bool in_range = half == 0 || kb < d.K;
    const int64_t koff = half ? offb : offa;
    const int kw = half ? kb : ka;
    unsigned char* sA = smem + buf * BUF + wave * 1024;
    unsigned char* sW = smem + buf * BUF + BM * 128 + wave * 1024;
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
      const void* src = in_range ? static_cast<const void*>(A + abase[i] + koff)
                                 : static_cast<const void*>(g_zero_page);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sA + i * 4096), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
      const void* src = in_range ? static_cast<const void*>(W + wbase[i] + kw)
                                 : static_cast<const void*>(g_zero_page);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sW + i * 4096), 16, 0, 0);
    }
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int lr = lane & 15, lq = lane >> 4;

  auto read_frags = [&](int buf, int kb, bf16x8 (&af)[MI], bf16x8 (&bf)[NI]) {
    const unsigned char* sA = smem + buf * BUF;
    const unsigned char* sW = sA + BM * 128;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int row = wm * TM + i * 16 + lr;
      const int slot = (kb * 4 + lq) ^ ((row >> 1) & 7);
      af[i] = *reinterpret_cast<const bf16x8*>(sA + row * 128 + (slot << 4));
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int row = wn * TN + j * 16 + lr;
      const int slot = (kb * 4 + lq) ^ ((row >> 1) & 7);
      bf[j] = *reinterpret_cast<const bf16x8*>(sW + row * 128 + (slot << 4));
    }
  };
  auto mma = [&](const bf16x8 (&af)[MI], const bf16x8 (&bf)[NI]) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
  };

  const int nk = (d.K + BK - 1) / BK;
  bf16x8 a0[MI], b0[NI], a1[MI], b1[NI];
  issue(0, 0);
  __syncthreads();
  read_frags(0, 0, a0, b0);
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    const bool more = kt + 1 < nk;
    if (more) issue((kt + 1) * BK, buf ^ 1);
    read_frags(buf, 1, a1, b1);
    mma(a0, b0);
    __syncthreads();
    if (more) read_frags(buf ^ 1, 0, a0, b0);
    mma(a1, b1);
  }

  // ---- epilogue: lane (lr, lq) of block (i, j): row m = ..+lr, columns n0..n0+3 (N % 4 == 0) ----
  const float* __restrict__ bias = d.bias ? d.bias + bz : nullptr;
  u16* __restrict__ C16 = reinterpret_cast<u16*>(d.C);
  const u16* __restrict__ R16 = reinterpret_cast<const u16*>(d.R);
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = tm * BM + wm * TM + i * 16 + lr;
    if (m >= d.M) continue;
    const int64_t crow = cz + (d.c_rowoff ? (int64_t)d.c_rowoff[m] : (int64_t)m * d.ldc);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n0 = tn * BN + wn * TN + j * 16 + lq * 4;
      if (n0 >= d.N) continue;
      f32x4 v = acc[i][j];
      if (bias) {
        const float4 b4 = *reinterpret_cast<const float4*>(bias + n0);
        v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], d.act) * d.alpha;
      if (d.R) {
        if (d.r_bf16) {
          const ushort4 r4 = *reinterpret_cast<const ushort4*>(R16 + crow + n0);
          v[0] += bf16_bits_to_f32(r4.x); v[1] += bf16_bits_to_f32(r4.y);
          v[2] += bf16_bits_to_f32(r4.z); v[3] += bf16_bits_to_f32(r4.w);
        } else {
          const float4 r4 = *reinterpret_cast<const float4*>(d.R + crow + n0);
          v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
        }
      }
      if (d.post_relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      if (d.c_bf16) {
        ushort4 o;
        o.x = f32_to_bf16_bits(v[0]); o.y = f32_to_bf16_bits(v[1]);
        o.z = f32_to_bf16_bits(v[2]); o.w = f32_to_bf16_bits(v[3]);
        *reinterpret_cast<ushort4*>(C16 + crow + n0) = o;
      } else {
        *reinterpret_cast<float4*>(d.C + crow + n0) = make_float4(v[0], v[1], v[2], v[3]);
      }
      if (d.WS) {
        float4* w = reinterpret_cast<float4*>(d.WS + (int64_t)m * d.ldws + n0);
        float4 a = d.ws_init ? make_float4(0.f, 0.f, 0.f, 0.f) : *w;
        a.x += d.ws_w * v[0]; a.y += d.ws_w * v[1]; a.z += d.ws_w * v[2]; a.w += d.ws_w * v[3];
        *w = a;
      }
    }
  }
}

template <int BM, int BN, int WGM, int WGN>
int launch_lowp_cfg(const dzn_gemm_desc& d, hipStream_t s) {
  const int tilesM = (d.M + BM - 1) / BM, tilesN = (d.N + BN - 1) / BN;
  const size_t lds = 2 * (BM + BN) * 128;
  auto kern = gemm_lowp_kernel<BM, BN, WGM, WGN>;
  static unsigned long long attr_mask = 0;  // one bit per HIP device: function attributes are per device
  if (first_use_on_device(attr_mask)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  dim3 grid(tilesM * tilesN, d.nz > 0 ? d.nz : 1, 1);
  int pid = -1;
  if (prof_enabled()) {
    char cls[64];
    static const bool by_shape = getenv("DZN_PROFILE_SHAPES") != nullptr;
    if (by_shape)
      snprintf(cls, sizeof(cls), "gemm_bf16_%dx%d M%d N%d K%d z%d", BM, BN, d.M, d.N, d.K, d.nz);
    else
      snprintf(cls, sizeof(cls), "gemm_bf16_%dx%d", BM, BN);
    const double fl = d.alg_flops > 0 ? d.alg_flops * d.nz : 2.0 * d.M * d.N * d.K * d.nz;
    pid = prof_begin(s, cls, fl, 0.0);
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, d);
  prof_end(pid, s);
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}

}  // namespace

// A bf16 / W bf16 path.  Requirements: K % 32 == 0, kc % 32 == 0, N % 4 == 0, all strides and
// row offsets multiples of 8 elements.
int launch_gemm_lowp(const dzn_gemm_desc& d, hipStream_t s) {
  if ((d.K & 31) || (d.kc & 31) || (d.N & 3) || !d.W16) return DZN_E_INVALID;
  if (d.N <= 32) return launch_lowp_cfg<256, 32, 4, 1>(d, s);
  if (d.N <= 64) return launch_lowp_cfg<128, 64, 2, 2>(d, s);
  const int cand[4] = {192, 160, 128, 96};
  int best = 128, best_cols = 1 << 30;
  for (int c : cand) {
    const int cols = (d.N + c - 1) / c * c;
    if (cols < best_cols) { best_cols = cols; best = c; }
  }
  switch (best) {
    case 192: return launch_lowp_cfg<128, 192, 2, 2>(d, s);
    case 160: return launch_lowp_cfg<128, 160, 2, 2>(d, s);
    case 96: return launch_lowp_cfg<128, 96, 2, 2>(d, s);
    default: return launch_lowp_cfg<128, 128, 2, 2>(d, s);
  }
}

#endif  // DZN_TUNING
