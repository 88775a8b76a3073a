// gemm_wide.hip — (r4) the f32h contraction (two-term fp16 split, three products; gemm_split.hip) on 256 x 256 tiles with ONE
// wavefront per SIMD: 4 wavefronts (2 x 2), each a 128 x 128 sub-tile = 256 accumulator registers out of a 512-register
// budget, the K tile consumed in 8 row-block phases with the next K tile's loads interleaved between the phases' MFMAs.
//
// Why this shape.  The instruction stream of the 128 x 128 kernel per wavefront and K tile is 48 MFMAs (768 matrix-pipe
// cycles) against 8 LDS-DMA pieces (60-185 issue cycles each, guide), 20 ds_read_b128 and ~100 VALU instructions of operand
// split, with two wavefronts per SIMD to overlap them: the matrix pipe is 42-45 % busy whatever is done to the prefetch depth,
// the MFMA shape or the workgroup's life time (profiles/r4_gemm_{a3,m32,persist}_probe.txt).  Per MFMA this tile needs
// 0.37 x the LDS-DMA pieces (16 per 192 MFMAs), 0.4 x the fragment reads (32 per 192) and the same split work, and the
// 512-register budget lets a wavefront hold BOTH fragment sets of the W tile plus the raw A rows, so nothing but one
// `s_waitcnt` + `s_barrier` per K tile interrupts the MFMA stream:
//
//   top of step kt:  s_waitcnt vmcnt(0) (my pieces of tile kt+1 have landed), s_barrier (everyone's have, and every wavefront
//                    holds tile kt in registers, so the stage of tile kt is free)
//   row block i = 0..7:   split A rows of block i (tile kt)  ->  [i < 4: four LDS-DMA pieces of tile kt+2 into the free stage]
//                    ->  ds_read A rows of block i and W column block i of tile kt+1 into the registers just retired / the other
//                    W set  ->  24 MFMAs (8 column blocks x 3 products)
//
// Same LDS images (XOR-swizzled 128-B A rows, [BN][64 B] W planes), same weight planes, same exact power-of-two scaling and
// the same epilogue (gemm_epilogue<256, 256, 128, 128, 8, 8>) as gemm_split_kernel.  The summation order over k inside an
// accumulator is identical (k tiles in order, products lo*hi, hi*lo, hi*hi), so results are bit-identical to the 128 x 128
// tile (tests/test_ops_gpu.py under DZN_GEMM_CFG=wide).
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "gemm_epilogue.h"
#include "split.h"

namespace {

__device__ __forceinline__ int wswz(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }

template <int N>
__device__ __forceinline__ void wait_vm_lgkm0() {
  static_assert(N >= 0 && N < 64, "vmcnt range");
  __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | (0x7 << 4));
}

// generic -> global -> generic on a pointer field read (as a pointer) from the laundered kernel-argument segment: keeps the
// address-space inference on "global" (else every access through the descriptor copy becomes a flat_load / flat_store)
template <class T>
__device__ __forceinline__ T* as_global(T* p) {
  return (T*)(__attribute__((address_space(1))) T*)p;
}

// The kernel's descriptor (argument 0) re-read from the kernel-argument segment through a LAUNDERED pointer.  Inside a loop
// that contains the epilogue its ~60 scalar fields (and the reciprocals of uniform divisions) are loop invariants: hoisted,
// they overflow the scalar register file and the spill code lands between the epilogue's stores, each reload a drain of the
// store queue (measured on the persistent form, profiles/r4_gemm_persist_probe.txt).
__device__ __forceinline__ void fresh_desc(dzn_gemm_desc& dd) {
  static_assert(sizeof(dzn_gemm_desc) % 4 == 0, "copied as dwords");
  unsigned long long ki = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(ki));
  const __attribute__((address_space(4))) unsigned* kw = (const __attribute__((address_space(4))) unsigned*)ki;
  unsigned* dw = reinterpret_cast<unsigned*>(&dd);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(dzn_gemm_desc) / 4); ++i) dw[i] = kw[i];
  const __attribute__((address_space(4))) dzn_gemm_desc* kd = (const __attribute__((address_space(4))) dzn_gemm_desc*)ki;
  dd.A = as_global(kd->A);
  dd.W = as_global(kd->W);
  dd.W16 = as_global(kd->W16);
  dd.C = as_global(kd->C);
  dd.bias = as_global(kd->bias);
  dd.R = as_global(kd->R);
  dd.WS = as_global(kd->WS);
  dd.a_rowoff = as_global(kd->a_rowoff);
  dd.c_rowoff = as_global(kd->c_rowoff);
  dd.W3 = as_global(kd->W3);
  dd.ln_stats = as_global(kd->ln_stats);
  dd.ln_colsum = as_global(kd->ln_colsum);
  dd.W2h = as_global(kd->W2h);
  dd.col_scale = as_global(kd->col_scale);
  dd.a_amax = as_global(kd->a_amax);
  dd.c_amax = as_global(kd->c_amax);
  dd.stat_partial = as_global(kd->stat_partial);
  dd.stat_final = as_global(kd->stat_final);
  dd.z_count = as_global(kd->z_count);
  dd.z_list = as_global(kd->z_list);
}

template <int NP>
__global__ __launch_bounds__(256, 1) void gemm_wide_kernel(const dzn_gemm_desc d) {
  static_assert(NP == 2, "f32h form");
  constexpr int BM = 256, BN = 256, WGM = 2, WGN = 2, NW = 4, S = 2;
  constexpr int BK = 32;
  constexpr int TM = BM / WGM, TN = BN / WGN;        // 128 x 128 per wavefront
  constexpr int MI = TM / 16, NI = TN / 16;          // 8 x 8 blocks
  constexpr int RB = NW * 1024;                      // bytes per LDS-DMA round (1 KiB per wavefront)
  constexpr int ACH = BM * 128 / RB;                 // 8 rounds of the A tile
  constexpr int WROWS = NW * 16;                     // 64 rows of one W plane per round
  constexpr int WR = BN / WROWS;                     // 4 rounds per plane
  constexpr int SP = 2;
  constexpr int ABYTES = BM * 128, WPLANE = BN * 64, BUF = ABYTES + NP * WPLANE;   // 32 + 2 x 16 = 64 KB per stage
  constexpr int LPT = ACH + NP * WR;                 // 16 pieces per thread and tile
  static_assert(MI == NI && LPT == 16, "one W column block and two pieces per row-block phase");
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
  if (d.z_list) {
    if (z0 >= d.z_count[0]) return;
    z0 = d.z_list[z0];
  }
  const float* __restrict__ A = d.A + z0 * d.a_z0 + z1 * d.a_z1;
  const u16* __restrict__ W2 = reinterpret_cast<const u16*>(d.W2h) + SP * (z0 * d.w_z0 + z1 * d.w_z1);
  const int lr = lane & 15, lq = lane >> 4;
  // exact power-of-two row scales (gemm_split.hip); the inverses are re-derived for the epilogue instead of living across the loop
  auto row_scales = [&](float (&sc)[MI], float (&inv)[MI]) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      int m = tm * BM + wm * TM + i * 16 + lr;
      m = m < d.M ? m : d.M - 1;
      h2_scale(d.a_amax[d.amax_unit > 0 ? m / d.amax_unit : z0], sc[i], inv[i]);
    }
  };
  float a_scale[MI];
  {
    float inv_[MI];
    row_scales(a_scale, inv_);
  }
  const int64_t cz = z0 * d.c_z0 + z1 * d.c_z1;
  const int64_t bz = z0 * d.b_z0 + z1 * d.b_z1;

  // A: thread -> (row = tid/8 + 32 i, physical slot tid%8), logical chunk = slot ^ ((row>>1)&7)
  const int r0 = tid >> 3;
  const int csw = (tid & 7) ^ ((r0 >> 1) & 7);
  // round i reads row r0 + 32 i of the tile: one pointer, a uniform stride of 32 rows, and for the last row tile the last
  // round whose row exists (rows past M re-read that one; their accumulators are never stored).  (No a_rowoff here: the
  // dispatcher keeps gathered-row launches on the 128 x 128 kernel.)
  const int m_first = tm * BM + r0;                      // < M: a tile has at least one row and r0 < 32 ... see launcher
  const float* const aptr0 = A + (int64_t)(m_first < d.M ? m_first : d.M - 1) * d.lda + csw * 4;
  const int a_last = m_first < d.M ? (d.M - 1 - m_first) >> 5 : 0;
  const int64_t a_stride = 32 * d.lda;
  // W planes: thread -> (row = 16 wave + lane/4 + 64 i, physical slot lane%4)
  const int wr0 = wave * 16 + (lane >> 2);
  const int wsw = (lane & 3) ^ wswz(wr0);
  const int n_first = tn * BN + wr0;
  const u16* const wptr0 = W2 + (int64_t)(n_first < d.N ? n_first : d.N - 1) * SP * d.ldw + wsw * 8;
  const int w_last = n_first < d.N ? (d.N - 1 - n_first) >> 6 : 0;
  const int64_t w_stride = (int64_t)WROWS * SP * d.ldw;
  // K cursor of the next tile to fetch (two-level K addressing: kc contiguous, then a jump of ldk)
  int ik = 0, irem = 0;
  int64_t ikoff = 0;
  // piece j of 16: 0..7 = A rounds, 8..15 = (plane, W round)
  auto piece = [&](int stage, int j) {
    unsigned char* sbase = smem + stage * BUF + wave * 1024;
    if (j < ACH) {
      const float* src = aptr0 + (j < a_last ? j : a_last) * a_stride + ikoff;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sbase + j * RB), 16, 0, 0);
    } else {
      const int p = (j - ACH) / WR, i = (j - ACH) % WR;
      const u16* src = wptr0 + (i < w_last ? i : w_last) * w_stride + (SP * ik + p * 32);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sbase + ABYTES + p * WPLANE + i * RB), 16, 0, 0);
    }
  };
  auto advance = [&]() {
    ik += BK;
    irem += BK;
    ikoff += BK;
    if (irem == d.kc) { irem = 0; ikoff += d.ldk - d.kc; }
  };

  // fragment addresses: the swizzle terms do not depend on the block index (16 i rows: (row >> 1) & 7 and (row >> 2) & 3 see
  // lr only), so block i is a compile-time offset from ONE per-lane base per operand
  const int arow = wm * TM + lr, wrow = wn * TN + lr;
  const int abase = arow * 128 + ((lq ^ ((arow >> 1) & 7)) << 4);       // second half of the k range: abase ^ 64
  const int wbase = ABYTES + wrow * 64 + ((lq ^ wswz(wrow)) << 4);
  auto read_a = [&](int stage, int i, f32x4 (&a)[2]) {
    const unsigned char* base = smem + stage * BUF + i * 16 * 128;
    a[0] = *reinterpret_cast<const f32x4*>(base + abase);
    a[1] = *reinterpret_cast<const f32x4*>(base + (abase ^ 64));
  };
  auto read_w = [&](int stage, int j, u32x4 (&w)[NP]) {
    const unsigned char* base = smem + stage * BUF + j * 16 * 64;
#pragma unroll
    for (int p = 0; p < NP; ++p) w[p] = *reinterpret_cast<const u32x4*>(base + p * WPLANE + wbase);
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = d.K / BK;
  // prologue: tiles 0 and 1 in flight, tile 0's fragments into registers
#pragma unroll
  for (int j = 0; j < LPT; ++j) piece(0, j);
  advance();
  if (nk > 1) {
#pragma unroll
    for (int j = 0; j < LPT; ++j) piece(1, j);
    advance();
    wait_vm_lgkm0<LPT>();
  } else {
    wait_vm_lgkm0<0>();
  }
  __builtin_amdgcn_s_barrier();
  u32x4 wc[NI][NP];          // W fragments of the tile being multiplied (ONE set: refilled inside the last phase, see below)
  f32x4 ar[MI][2];           // its raw A rows; block i is refilled right after its split
#pragma unroll
  for (int j = 0; j < NI; ++j) read_w(0, j, wc[j]);
#pragma unroll
  for (int i = 0; i < MI; ++i) read_a(0, i, ar[i]);

  // One K tile.  MORE = a tile kt+1 exists, FETCH = a tile kt+2 exists: compile-time, so that the fragment registers are
  // overwritten unconditionally (a conditional ds_read is a phi of two register sets).
  //   first half  — row blocks 0..3, one phase each: [lo*hi over the 8 column blocks, hi*lo ..., hi*hi ...] (8 independent
  //                 accumulators between dependent MFMAs); each phase also splits row block i + 4 for the second half, issues
  //                 four LDS-DMA pieces of tile kt+2 and refills row blocks i, i + 4 of the raw A registers from tile kt+1;
  //   second half — row blocks 4..7 COLUMN-major: for column block j the 3 x 4 products of the four row blocks (4 independent
  //                 accumulators between dependent MFMAs), after which W fragment j is dead and is refilled from tile kt+1 —
  //                 a second W register set (64 registers) does not fit beside 256 accumulators.
  // Per accumulator the order of the three products and of the K tiles is the same in both halves and the same as in
  // gemm_split_kernel.
  auto step = [&](auto more_c, auto fetch_c, int kt) {
    constexpr bool MORE = decltype(more_c)::value, FETCH = decltype(fetch_c)::value;
    constexpr int PW[3] = {1, 0, 0}, PA[3] = {0, 1, 0};                     // lo*hi hi*lo hi*hi
    constexpr int HM = MI / 2;
    const int cur = kt & 1, nxt = cur ^ 1;
    if constexpr (MORE) {
      wait_vm_lgkm0<0>();                 // my pieces of tile kt+1 landed (the only loads in flight)
      __builtin_amdgcn_s_barrier();       // ... everyone's; and every wavefront holds tile kt in registers: stage `cur` is free
    }
    u32x4 af2[HM][NP];
#pragma unroll
    for (int i = 0; i < HM; ++i) {
      u32x4 af[NP];
      split8_h2(ar[i][0], ar[i][1], a_scale[i], af[0], af[1]);
      split8_h2(ar[i + HM][0], ar[i + HM][1], a_scale[i + HM], af2[i][0], af2[i][1]);
      if constexpr (FETCH) {
#pragma unroll
        for (int j = 0; j < LPT / HM; ++j) piece(cur, (LPT / HM) * i + j);
      }
      if constexpr (MORE) {
        read_a(nxt, i, ar[i]);
        read_a(nxt, i + HM, ar[i + HM]);
      }
#pragma unroll
      for (int t3 = 0; t3 < 3; ++t3)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = mfma_np<NP>(wc[j][PW[t3]], af[PA[t3]], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);   // phases stay phases: the scheduler must not pull later phases' loads up front
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
#pragma unroll
      for (int t3 = 0; t3 < 3; ++t3)
#pragma unroll
        for (int i = 0; i < HM; ++i) acc[HM + i][j] = mfma_np<NP>(wc[j][PW[t3]], af2[i][PA[t3]], acc[HM + i][j]);
      if constexpr (MORE) read_w(nxt, j, wc[j]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (FETCH) advance();
  };
  constexpr std::true_type T{};
  constexpr std::false_type F{};
  int kt = 0;
  for (; kt + 2 < nk; ++kt) step(T, T, kt);
  if (nk > 1) step(T, F, nk - 2);
  step(F, F, nk - 1);
  __syncthreads();   // the stages are dead: the epilogue's column vectors go there
#ifdef DZN_WIDE_NOEPI
  {
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    d.C[cz + (int64_t)(tm * BM + wm * TM + lr) * d.ldc + tn * BN + wn * TN + lq] = sum;
    return;
  }
#endif
  // Epilogue as a ROLLED loop over pairs of row blocks: gemm_epilogue<.., TM = 32, .., MI = 2, NI = 8> on the leading 64
  // accumulator registers, then the accumulator file rotates by one pair.  Fully unrolled over 8 x 8 blocks the epilogue is
  // 130 k instructions with 600 spilled registers; rolled it is the code of the 128 x 128 tile's, 4 x per tile.
  constexpr int EM = 2;
#pragma nounroll
  for (int it = 0; it < MI / EM; ++it) {
    dzn_gemm_desc de;
    fresh_desc(de);
    f32x4 blk[EM][NI];
    float inv[EM];
#pragma unroll
    for (int i = 0; i < EM; ++i) {
#pragma unroll
      for (int j = 0; j < NI; ++j) blk[i][j] = acc[i][j];
      int m = tm * BM + wm * TM + (it * EM + i) * 16 + lr;
      m = m < de.M ? m : de.M - 1;
      float sc_;
      h2_scale(de.a_amax[de.amax_unit > 0 ? m / de.amax_unit : z0], sc_, inv[i]);
    }
    gemm_epilogue<BM, BN, 16 * EM, TN, EM, NI, true>(de, blk, tm, tn, wm * (MI / EM) + it, wn, lr, lq, cz, bz, z0, inv, de.col_scale,
                                                    reinterpret_cast<float*>(smem) + wave * 3 * TN);
#pragma unroll
    for (int i = 0; i + EM < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = acc[i + EM][j];
  }
}

}  // namespace

int launch_gemm_wide(const dzn_gemm_desc& d, hipStream_t s) {
  constexpr int BM = 256, BN = 256, NP = 2;
  if ((d.K & 31) || (d.kc & 31) || d.ldw != d.K || d.a_rowoff || !d.W2h || !d.col_scale || !d.a_amax || d.w_z0 || d.w_z1) return DZN_E_INVALID;
  const int tilesM = (d.M + BM - 1) / BM, tilesN = (d.N + BN - 1) / BN;
  auto kern = gemm_wide_kernel<NP>;
  const size_t lds = 2 * (BM * 128 + NP * BN * 64);
  static unsigned long long attr_mask = 0;
  if (first_use_on_device(attr_mask))
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  dim3 grid(tilesM * tilesN, d.nz > 0 ? d.nz : 1, 1);
  int pid = -1;
  if (prof_enabled()) {
    char cls[64];
    static const bool by_shape = getenv("DZN_PROFILE_SHAPES") != nullptr;
    if (by_shape) snprintf(cls, sizeof(cls), "gemm_f32h_256x256 M%d N%d K%d z%d", d.M, d.N, d.K, d.nz);
    else snprintf(cls, sizeof(cls), "gemm_f32h_256x256");
    const double fl = d.alg_flops > 0 ? d.alg_flops * d.nz : 2.0 * d.M * d.N * d.K * d.nz;
    pid = prof_begin(s, cls, fl, gemm_alg_bytes(d, NP * 2));
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, d);
  prof_end(pid, s);
  if (hipGetLastError() != hipSuccess) return DZN_E_HIP;
  if (d.stat_partial && d.stat_final)
    return launch_stats_finalize(d.stat_partial, d.M, tilesN * 2, d.stat_C, d.stat_eps, d.stat_final, s);
  return DZN_OK;
}
