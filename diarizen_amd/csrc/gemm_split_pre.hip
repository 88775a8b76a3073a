// gemm_split_pre.hip — the contraction of gemm_split.hip for an A operand that is ALREADY split
// (dzn_gemm_desc.a_split3): three bf16 planes written once by the producing kernel, in the same
// k order inside each 32-block as the weight planes.
//
// Used where one activation element feeds many K tiles, so splitting it inside the contraction
// repeats the same VALU work many times: the positional conv (W2V/components.py:366-380: every
// element of the padded copy is read by 128 taps — pad_rows_split3_kernel writes the planes once).
// With no split in the loop the kernel is bf16-MFMA + ds_read only: both operands arrive by LDS-DMA
// as [rows][64 B] plane images (slot XOR g((row >> 2) & 3), conflict free), 6 products per block.
// Same pipeline as gemm_split.hip (mid-tile raw barrier, younger LDS-DMA tiles stay in flight,
// fragments of tile kt+1 read during the second half of tile kt), same fused epilogue.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "checked.h"
#include "common.h"
#include "gemm_epilogue.h"
#include "split.h"

DZN_CHECKED_TU(gemm_split_pre)

namespace {

__device__ __forceinline__ int pswz(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }

template <int N>
__device__ __forceinline__ void wait_vm_lgkm0() {
  static_assert(N >= 0 && N < 64, "vmcnt range");
  __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | (0x7 << 4));
}

// NP = 3: bf16 planes (six products); NP = 2 (DZN_PREC_F32_H2): fp16 planes of x * 2^e written by
// pad_rows_split2_kernel, which also SNAPSHOTS the |max| it scaled by (d.a_amax points at the snapshot, not at the
// live tracker: the positional conv updates the tensor it reads, so the live tracker moves while tiles start).
template <int BM, int BN, int WGM, int WGN, int S, int NP>
__global__ __launch_bounds__(WGM * WGN * 64) void gemm_split_pre_kernel(const dzn_gemm_desc d) {
  constexpr int NW = WGM * WGN;
  constexpr int BK = 32;
  constexpr int TM = BM / WGM, TN = BN / WGN;
  constexpr int MI = TM / 16, NI = TN / 16, MH = MI / 2;
  constexpr int RB = NW * 1024;
  constexpr int ROWS = NW * 16;           // plane rows per LDS-DMA round (64-B rows)
  constexpr int AR = BM / ROWS, WR = BN / ROWS;
  constexpr int SP = NP == 3 ? 3 : 2;     // planes STORED per weight row (NP = 1, DZN_PREC_F16, reads the leading one)
  constexpr int APLANE = BM * 64, WPLANE = BN * 64, BUF = NP * (APLANE + WPLANE);
  constexpr int LPT = NP * (AR + WR);
  static_assert(BM % ROWS == 0 && BN % ROWS == 0, "whole rounds");
  static_assert(MI % 2 == 0, "two row halves per wavefront tile");
  static_assert(S >= 2 && (S - 1) * LPT < 64, "vmcnt range");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
  const u16* __restrict__ A3 = reinterpret_cast<const u16*>(d.A) + z0 * d.a_z0 + z1 * d.a_z1;
  const u16* __restrict__ W3 =
      reinterpret_cast<const u16*>(NP == 3 ? d.W3 : d.W2h) + SP * (z0 * d.w_z0 + z1 * d.w_z1);
  float row_inv[MI];     // NP = 2: inverse of the per-unit scale the producer applied (pad_rows_split2_kernel)
#pragma unroll
  for (int i = 0; i < MI; ++i) row_inv[i] = 1.f;
  if constexpr (NP <= 2) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      int m = tm * BM + wm * TM + i * 16 + (lane & 15);
      m = m < d.M ? m : d.M - 1;
      float unused;
      h2_scale(d.a_amax[d.amax_unit > 0 ? m / d.amax_unit : z0], unused, row_inv[i]);
    }
  }
  const int64_t cz = z0 * d.c_z0 + z1 * d.c_z1;
  const int64_t bz = z0 * d.b_z0 + z1 * d.b_z1;

  // thread -> (row = 16 wave + lane/4 + ROWS i, physical slot lane%4), logical 8-element chunk = slot ^ g
  const int pr0 = wave * 16 + (lane >> 2);
  const int psw = (lane & 3) ^ pswz(pr0);
  const u16* aptr[AR];
  const u16* wptr[WR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    int m = tm * BM + pr0 + ROWS * i;
    m = m < d.M ? m : d.M - 1;
    aptr[i] = A3 + (d.a_rowoff ? (int64_t)d.a_rowoff[m] : (int64_t)m * d.lda) + psw * 8;
  }
#pragma unroll
  for (int i = 0; i < WR; ++i) {
    int n = tn * BN + pr0 + ROWS * i;
    n = n < d.N ? n : d.N - 1;
    wptr[i] = W3 + (int64_t)n * SP * d.ldw + psw * 8;
  }

  int ik = 0, irem = 0;
  int64_t ikoff = 0;
  auto issue = [&](int stage) {
    unsigned char* sA = smem + stage * BUF + wave * 1024;
    unsigned char* sW = sA + NP * APLANE;
    DZN_CHECK(stage >= 0 && stage < S && ik < d.K, 0x301, stage);                              // a stage of the ring, a k tile of the operand
#pragma unroll
    for (int p = 0; p < NP; ++p) {
#pragma unroll
      for (int i = 0; i < AR; ++i)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(aptr[i] + p * d.a_plane + ikoff),
            (__attribute__((address_space(3))) void*)(sA + p * APLANE + i * RB), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < WR; ++i)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(wptr[i] + SP * ik + p * 32),
            (__attribute__((address_space(3))) void*)(sW + p * WPLANE + i * RB), 16, 0, 0);
    }
    ik += BK;
    irem += BK;
    ikoff += BK;
    if (irem == d.kc) { irem = 0; ikoff += d.ldk - d.kc; }
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int lr = lane & 15, lq = lane >> 4;
  int aoff[MI], woff[NI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int row = wm * TM + i * 16 + lr;
    aoff[i] = row * 64 + ((lq ^ pswz(row)) << 4);
  }
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int row = wn * TN + j * 16 + lr;
    woff[j] = NP * APLANE + row * 64 + ((lq ^ pswz(row)) << 4);
  }
  auto read_w = [&](int stage, u32x4 (&wf)[NI][NP]) {
    const unsigned char* base = smem + stage * BUF;
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int p = 0; p < NP; ++p) wf[j][p] = *reinterpret_cast<const u32x4*>(base + p * WPLANE + woff[j]);
  };
  auto read_a = [&](int stage, u32x4 (&af)[MI][NP]) {
    const unsigned char* base = smem + stage * BUF;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int p = 0; p < NP; ++p) af[i][p] = *reinterpret_cast<const u32x4*>(base + p * APLANE + aoff[i]);
  };
  // products of one 16-row block, smallest terms first (same order as gemm_split.hip)
  auto mma = [&](int i, const u32x4 (&wf)[NI][NP], const u32x4 (&a)[NP]) {
    if constexpr (NP == 3) {
      constexpr int PW[6] = {2, 0, 1, 1, 0, 0}, PA[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = mfma_np<NP>(wf[j][PW[t]], a[PA[t]], acc[i][j]);
    } else if constexpr (NP == 2) {
      constexpr int PW[3] = {1, 0, 0}, PA[3] = {0, 1, 0};
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = mfma_np<NP>(wf[j][PW[t]], a[PA[t]], acc[i][j]);
    } else {
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = mfma_np<NP>(wf[j][0], a[0], acc[i][j]);
    }
  };

  const int nk = d.K / BK;
#pragma unroll
  for (int s = 0; s < S; ++s)
    if (s < nk) issue(s);
  if (nk >= S) wait_vm_lgkm0<(S - 1) * LPT>();
  else wait_vm_lgkm0<0>();
  __builtin_amdgcn_s_barrier();
  u32x4 wfa[NI][NP], wfb[NI][NP], af[MI][NP];
  read_w(0, wfa);
  read_a(0, af);
  int stage = 0;
  auto step = [&](int kt, const u32x4 (&wc)[NI][NP], u32x4 (&wn_)[NI][NP]) {
    const bool more = kt + 1 < nk;
#pragma unroll
    for (int i = 0; i < MH; ++i) mma(i, wc, af[i]);
    u32x4 a2[MI - MH][NP];
#pragma unroll
    for (int i = MH; i < MI; ++i)
#pragma unroll
      for (int p = 0; p < NP; ++p) a2[i - MH][p] = af[i][p];
    const int nstage = stage + 1 == S ? 0 : stage + 1;
    __builtin_amdgcn_sched_barrier(0);
    if (more) {
      if (kt + S <= nk) wait_vm_lgkm0<(S - 2) * LPT>();
      else wait_vm_lgkm0<0>();
      __builtin_amdgcn_s_barrier();
      if (kt + S < nk) issue(stage);
      read_w(nstage, wn_);
      read_a(nstage, af);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = MH; i < MI; ++i) mma(i, wc, a2[i - MH]);
    stage = nstage;
  };
  for (int kt = 0; kt < nk; kt += 2) {
    step(kt, wfa, wfb);
    if (kt + 1 < nk) step(kt + 1, wfb, wfa);
  }
  if constexpr (true) {
    // column vectors of the epilogue in LDS (48 registers less than holding them; the stages are dead by now)
    __syncthreads();
    gemm_epilogue<BM, BN, TM, TN, MI, NI, true>(d, acc, tm, tn, wm, wn, lr, lq, cz, bz, z0, row_inv,
                                                NP <= 2 ? d.col_scale + z0 * d.b_z0 + z1 * d.b_z1 : nullptr,
                                                reinterpret_cast<float*>(smem) + (wm * WGN + wn) * 3 * TN);
  } else {
    gemm_epilogue<BM, BN, TM, TN, MI, NI>(d, acc, tm, tn, wm, wn, lr, lq, cz, bz, z0, row_inv,
                                          NP <= 2 ? d.col_scale + z0 * d.b_z0 + z1 * d.b_z1 : nullptr);
  }
}

template <int BM, int BN, int WGM, int WGN, int S, int NP>
int launch_pre_cfg(const dzn_gemm_desc& d, hipStream_t s) {
  const int tilesM = (d.M + BM - 1) / BM, tilesN = (d.N + BN - 1) / BN;
  const size_t lds = (size_t)S * NP * (BM + BN) * 64;
  auto kern = gemm_split_pre_kernel<BM, BN, WGM, WGN, S, NP>;
  static unsigned long long attr_mask = 0;  // one bit per HIP device: function attributes are per device
  if (first_use_on_device(attr_mask)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
  }
  dim3 grid(tilesM * tilesN, d.nz > 0 ? d.nz : 1, 1);
  int pid = -1;
  if (prof_enabled()) {
    char cls[64];
    static const bool by_shape = getenv("DZN_PROFILE_SHAPES") != nullptr;
    if (by_shape)
      snprintf(cls, sizeof(cls), "gemm_%s_pre_%dx%d M%d N%d K%d z%d", NP == 3 ? "f32s" : NP == 2 ? "f32h" : "f16", BM, BN,
               d.M, d.N, d.K, d.nz);
    else
      snprintf(cls, sizeof(cls), "gemm_%s_pre_%dx%d", NP == 3 ? "f32s" : NP == 2 ? "f32h" : "f16", BM, BN);
    const double fl = d.alg_flops > 0 ? d.alg_flops * d.nz : 2.0 * d.M * d.N * d.K * d.nz;
    pid = prof_begin(s, cls, fl, gemm_alg_bytes(d, NP * 2));
  }
  hipLaunchKernelGGL(kern, grid, dim3(WGM * WGN * 64), lds, s, d);
  prof_end(pid, s);
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}

// x fp32 [B, L, D] -> three bf16 planes of the zero-padded copy [B, Lp, D] (rows shifted by `pad`), the 32
// channels of every block stored in fragment order (position 8q+e <-> channel 4q+e, 16+4q+(e-4))
// cg / cgp: channels per group of the source / of the planes (cgp = cg rounded up to 32; channels cg .. cgp-1 of a group
// are written as zeros — the base models' 48-channel groups become 64, so that the contraction's K tiles never straddle
// a tap; D is the PLANE row width G * cgp, the source rows are G * cg wide)
__global__ __launch_bounds__(256) void pad_rows_split3_kernel(const float* __restrict__ x, u16* __restrict__ planes,
                                                              int64_t plane_stride, int L, int Lp, int pad, int D, int cg, int cgp) {
  const int b = blockIdx.y;
  const int Ds = D / cgp * cg;                                // source row width
  const int chunks = D / 8;                                   // 16-byte output chunks per row
  const int64_t n = (int64_t)Lp * chunks;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / chunks), c = (int)(i - (int64_t)r * chunks);
    const int blk = c >> 2, q = c & 3;
    const int t = r - pad;
    f32x4 u = (f32x4){0.f, 0.f, 0.f, 0.f}, v = u;
    if (t >= 0 && t < L) {
      const float* row = x + ((int64_t)b * L + t) * Ds;
      const int c0 = blk * 32 + 4 * q, c1 = c0 + 16;          // plane channels of the two float4s
      const int g0 = c0 / cgp, i0 = c0 - g0 * cgp, g1 = c1 / cgp, i1 = c1 - g1 * cgp;
      if (i0 < cg) {
        const float4 a = *reinterpret_cast<const float4*>(row + g0 * cg + i0);
        u = (f32x4){a.x, a.y, a.z, a.w};
      }
      if (i1 < cg) {
        const float4 e = *reinterpret_cast<const float4*>(row + g1 * cg + i1);
        v = (f32x4){e.x, e.y, e.z, e.w};
      }
    }
    bf16x8 ph, pm, pl;
    split8(u, v, ph, pm, pl);
    u16* dst = planes + ((int64_t)b * Lp + r) * D + blk * 32 + q * 8;
    *reinterpret_cast<bf16x8*>(dst) = ph;
    *reinterpret_cast<bf16x8*>(dst + plane_stride) = pm;
    *reinterpret_cast<bf16x8*>(dst + 2 * plane_stride) = pl;
  }
}

// two-term fp16 variant: planes of x * 2^e (e from the live |max| tracker of x); the |max| that was used is
// snapshotted for the consuming contraction (see gemm_split_pre_kernel)
__global__ __launch_bounds__(256) void pad_rows_split2_kernel(const float* __restrict__ x, u16* __restrict__ planes,
                                                              int64_t plane_stride, int L, int Lp, int pad, int D,
                                                              const float* __restrict__ amax, float* __restrict__ snapshot, int cg, int cgp) {
  const int b = blockIdx.y;
  const int Ds = D / cgp * cg;
  const float am = amax[b];              // per-window tracker
  float sc, inv;
  h2_scale(am, sc, inv);
  if (blockIdx.x == 0 && threadIdx.x == 0) snapshot[b] = am;
  const int chunks = D / 8;
  const int64_t n = (int64_t)Lp * chunks;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / chunks), c = (int)(i - (int64_t)r * chunks);
    const int blk = c >> 2, q = c & 3;
    const int t = r - pad;
    f32x4 u = (f32x4){0.f, 0.f, 0.f, 0.f}, v = u;
    if (t >= 0 && t < L) {
      const float* row = x + ((int64_t)b * L + t) * Ds;
      const int c0 = blk * 32 + 4 * q, c1 = c0 + 16;          // plane channels of the two float4s
      const int g0 = c0 / cgp, i0 = c0 - g0 * cgp, g1 = c1 / cgp, i1 = c1 - g1 * cgp;
      if (i0 < cg) {
        const float4 a = *reinterpret_cast<const float4*>(row + g0 * cg + i0);
        u = (f32x4){a.x, a.y, a.z, a.w};
      }
      if (i1 < cg) {
        const float4 e = *reinterpret_cast<const float4*>(row + g1 * cg + i1);
        v = (f32x4){e.x, e.y, e.z, e.w};
      }
    }
    u32x4 ph, pl;
    split8_h2(u, v, sc, ph, pl);
    u16* dst = planes + ((int64_t)b * Lp + r) * D + blk * 32 + q * 8;
    *reinterpret_cast<u32x4*>(dst) = ph;
    *reinterpret_cast<u32x4*>(dst + plane_stride) = pl;
  }
}

}  // namespace

// A = plane 0 of the pre-split operand (bf16), planes a_plane elements apart.  Requirements: K % 32 == 0,
// kc % 32 == 0, ldw == K, all A offsets multiples of 8 elements.
template <int NP>
int launch_gemm_split_pre_np(const dzn_gemm_desc& d, hipStream_t s) {
  static const char* force = getenv("DZN_GEMM_CFG");   // tuning knob
  if (force && !strcmp(force, "256x128")) return launch_pre_cfg<256, 128, 4, 2, 2, NP>(d, s);
  if (force && !strcmp(force, "128x128")) return launch_pre_cfg<128, 128, 2, 2, 2, NP>(d, s);
  if (force && !strcmp(force, "128x64")) return launch_pre_cfg<128, 64, 4, 1, 2, NP>(d, s);
  if (d.N > 64) return launch_pre_cfg<256, 128, 4, 2, 2, NP>(d, s);   // 144 KB of LDS (NP = 3), 8 wavefronts
  return launch_pre_cfg<128, 64, 4, 1, 2, NP>(d, s);   // 72 KB of LDS (NP = 3): two workgroups per CU
}

int launch_gemm_split_pre(const dzn_gemm_desc& d, hipStream_t s) {
  if ((d.K & 31) || (d.kc & 31) || d.ldw != d.K || !d.a_split3 || d.a_plane <= 0) return DZN_E_INVALID;
  if (d.a_split3 == 2) {   // two fp16 planes
    if (!d.W2h || !d.col_scale || !d.a_amax || d.w_z0 * 1 != d.w_z0) return DZN_E_INVALID;
    return d.precision == DZN_PREC_F16 ? launch_gemm_split_pre_np<1>(d, s) : launch_gemm_split_pre_np<2>(d, s);
  }
  if (!d.W3) return DZN_E_INVALID;
  return launch_gemm_split_pre_np<3>(d, s);
}

int launch_pad_rows_split2(const float* x, void* planes, int64_t plane_stride, int B, int L, int Lp, int pad, int D,
                           const float* amax, float* snapshot, hipStream_t st, int cg, int cgp) {
  if (cg <= 0) cg = cgp = 32;
  if (cg % 4 || cgp % 32 || cg > cgp || D % cgp) return DZN_E_INVALID;
  ProfScope prof_scope_(st, "pad_rows_split2", 0.0, (double)B * D * (L * 4.0 + Lp * 4.0));   // fp32 in, two fp16 planes out
  if (D % 32 || !amax || !snapshot) return DZN_E_INVALID;
  int64_t g = cdiv64((int64_t)Lp * (D / 8), 256);
  g = g > 4096 ? 4096 : g;
  hipLaunchKernelGGL(pad_rows_split2_kernel, dim3((unsigned)g, B), dim3(256), 0, st, x, static_cast<u16*>(planes),
                     plane_stride, L, Lp, pad, D, amax, snapshot, cg, cgp);
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}

extern "C" int dzn_op_split_rows(const float* x, void* planes, int64_t plane_stride, int64_t rows, int32_t D,
                                 void* stream);

int launch_pad_rows_split3(const float* x, void* planes, int64_t plane_stride, int B, int L, int Lp, int pad, int D,
                           hipStream_t st, int cg, int cgp) {
  if (cg <= 0) cg = cgp = 32;
  if (cg % 4 || cgp % 32 || cg > cgp || D % cgp) return DZN_E_INVALID;
  ProfScope prof_scope_(st, "pad_rows_split3", 0.0, (double)B * D * (L * 4.0 + Lp * 6.0));
  if (D % 32) return DZN_E_INVALID;
  int64_t g = cdiv64((int64_t)Lp * (D / 8), 256);
  g = g > 4096 ? 4096 : g;
  hipLaunchKernelGGL(pad_rows_split3_kernel, dim3((unsigned)g, B), dim3(256), 0, st, x, static_cast<u16*>(planes),
                     plane_stride, L, Lp, pad, D, cg, cgp);
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}

// x fp32 [rows, D] -> three bf16 planes [rows, D] (plane_stride elements apart) in fragment order: the
// pre-split A operand of gemm_split_pre.hip (tests / benchmarks; engines use the fused producers)
extern "C" int dzn_op_split_rows(const float* x, void* planes, int64_t plane_stride, int64_t rows, int32_t D,
                                 void* stream) {
  if (!x || !planes || rows <= 0 || rows > 0x7fffffff) return DZN_E_INVALID;
  return launch_pad_rows_split3(x, planes, plane_stride, 1, (int)rows, (int)rows, 0, D,
                                reinterpret_cast<hipStream_t>(stream));
}
