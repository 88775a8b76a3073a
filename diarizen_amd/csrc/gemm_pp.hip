// gemm_pp.hip — the fp32-grade split contraction (see gemm_split.hip for the arithmetic) re-scheduled as an
// 8-wavefront "ping-pong" workgroup (round 3).
//
// Why.  gemm_split_kernel<128,128> (r2) sat at MfmaUtil 35 %: its ablation (profiles/r2_gemm_ablation.txt) showed no
// saturated resource — every wavefront alternates ~770 cycles of MFMAs with an equally long serial stretch of
// LDS-DMA issue (8 x 1 KiB pieces, 60-185 cycles each), 20 fragment ds_reads, a vmcnt wait, a barrier and ~50 VALU
// ops of operand split, and with two such wavefronts per SIMD from unrelated workgroups the stretches overlap only by
// chance.  This kernel makes the overlap structural:
//
//   * 256 x BN tile, 512 threads = 8 wavefronts, one workgroup per CU.  Wavefronts 0-3 (group 0) own rows 0..127,
//     wavefronts 4-7 (group 1) rows 128..255; the hardware places wavefront w and w+4 on the same SIMD.  Group 1 runs
//     ONE barrier interval behind group 0, so in every interval one wavefront of each SIMD is in a pure-MFMA
//     "compute" segment (48 back-to-back v_mfma_f32_16x16x32_f16 for NP = 2, BN = 128) while its partner is in the
//     "load" segment of its next K tile: W fragment ds_reads, its share of the LDS-DMA, its own A loads, and the
//     operand split of the A rows — everything that is not an MFMA.
//   * A never touches LDS.  With 8 x 1 wavefront tiles (32 rows x BN columns each) an A row is consumed by exactly one
//     wavefront, so each lane loads its MFMA fragment (row lr, k = 4q..4q+3 and 16+4q..16+4q+3 — the k order the
//     weight planes are packed in) straight into registers with two global_load_dwordx4, AD K tiles ahead.  That
//     halves the LDS-DMA pieces per K tile, removes the A tile's LDS write + read, and leaves LDS to W alone:
//     S stages x NP planes x BN x 64 B (48 KB at BN = 128, NP = 2, S = 3).
//   * W arrives by LDS-DMA exactly as in gemm_split.hip (same plane images, same XOR swizzle, so the fragment reads
//     are conflict-free ds_read_b128), spread over all 8 wavefronts; a stage is refilled S-1 tiles ahead.
//
// Hazards (I(n) = the interval after barrier n; group 0 loads tile t in I(2t) and multiplies it in I(2t+1), group 1 one
// interval later):  stage t % S is read in I(2t) and I(2t+1) and refilled for tile t+S by the load segments of tile
// t+1, i.e. in I(2t+2) / I(2t+3) — after both readers passed a barrier.  A wavefront leaves a load segment only when its
// own pieces of tile t+1 have landed (s_waitcnt vmcnt(INFL): VMEM returns in order, so "all but the INFL youngest"),
// and the barrier that ends the segment publishes them: group 1's segment of tile t ends at barrier 2t+2, the first
// reader of tile t+1 (group 0) starts after it.  The same wait covers the A registers of the tile about to be split.
//
// Contract: dzn_gemm_desc + the shared fused epilogue (gemm_epilogue.h), weight planes of dzn_op_split_weights[_h2].
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "common.h"
#include "gemm_epilogue.h"
#include "split.h"

namespace {

__device__ __forceinline__ int wswz_pp(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }

// s_waitcnt vmcnt(N) alone (gfx9 encoding: vmcnt = [3:0] + [15:14]; expcnt / lgkmcnt left at their maxima)
template <int N>
__device__ __forceinline__ void wait_vm_only() {
  static_assert(N >= 0 && N < 64, "vmcnt range");
  __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | (0x7 << 4) | (0xF << 8));
}
__device__ __forceinline__ void wait_lgkm0() { __builtin_amdgcn_s_waitcnt(0xF | (0x3 << 14) | (0x7 << 4)); }

// The A fragments are loaded by INLINE-ASM global_load_dwordx4 and waited for by an inline-asm s_waitcnt that names the
// destination registers as in/out operands.  Compiler-visible loads do not work here: its waitcnt pass re-derives the
// wait for every register it saw loaded and, across the rotating register sets of the unrolled K loop, falls back to
// vmcnt(0) — which drains the whole prefetch pipeline every K tile (seen in the ISA of the first version of this file).
// With asm the compiler believes the registers are valid right after the load; the tied wait is what makes that true
// before their first real use, and volatile asm keeps loads, LDS-DMA builtins, waits and barriers in program order.
template <int OFF>
__device__ __forceinline__ void asm_load16(f32x4& dst, const float* p) {
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(p), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm_tied(f32x4 (&a)[2][2]) {
  static_assert(N >= 0 && N < 64, "vmcnt range");
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]) : "n"(N) : "memory");
}

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// BN columns per workgroup, NP terms per operand (3 = bf16, 2 = fp16, 1 = fp16 leading term only), S LDS stages of W,
// A fragments loaded AD K tiles ahead.
template <int BN, int NP, int S, int AD>
__global__ __launch_bounds__(512) void gemm_pp_kernel(const dzn_gemm_desc d) {
  constexpr int BM = 256, TM = 32, MI = 2, NI = BN / 16, BK = 32;
  constexpr int SP = NP == 3 ? 3 : 2;                      // planes STORED per weight row
  constexpr int WPLANE = BN * 64, STAGE = NP * WPLANE;     // bytes
  constexpr int NPIECE = STAGE / 1024;                     // 1 KiB LDS-DMA pieces per stage
  static_assert(NPIECE % 8 == 0, "every wavefront issues the same number of pieces");
  constexpr int PPW = NPIECE / 8;
  constexpr int OPS = PPW + 2 * MI;                        // VMEM instructions per wavefront per load segment
  constexpr int INFL = (AD < S - 2 ? AD : S - 2) * OPS;    // ... that may still be in flight when a segment ends
  constexpr int AHEAD = (S - 1 > AD ? S - 1 : AD);
  constexpr int NR = AD + 1;                               // rotating register sets of raw A fragments
  static_assert(S >= 3 && AD >= 1 && INFL < 64, "pipeline depths");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int lr = lane & 15, lq = lane >> 4;
  const int tilesN = (d.N + BN - 1) / BN;
  int tile;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = tile / tilesN, tn = tile % tilesN;
  const int z = blockIdx.y;
  const int z0 = z / d.zdiv, z1 = z - z0 * d.zdiv;
  const float* __restrict__ A = d.A + z0 * d.a_z0 + z1 * d.a_z1;
  const u16* __restrict__ W3 =
      reinterpret_cast<const u16*>(NP == 3 ? d.W3 : d.W2h) + SP * (z0 * d.w_z0 + z1 * d.w_z1);
  const int64_t cz = z0 * d.c_z0 + z1 * d.c_z1;
  const int64_t bz = z0 * d.b_z0 + z1 * d.b_z1;

  // A row pointers (the optional row-offset table is a compiler-visible load: forced to resolve HERE, before any of
  // our own loads is in flight), and the |max| tracker of every row's unit as an asm load that the prologue wait covers
  float a_scale[MI], row_inv[MI], a_max[MI];
  const float* aptr[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    int m = tm * BM + wave * TM + i * 16 + lr;
    m = m < d.M ? m : d.M - 1;
    a_scale[i] = row_inv[i] = 1.f;
    a_max[i] = 0.f;
    aptr[i] = A + (d.a_rowoff ? (int64_t)d.a_rowoff[m] : (int64_t)m * d.lda) + lq * 4;
    asm volatile("" : "+v"(aptr[i]));
    if constexpr (NP <= 2) {
      const float* tp = d.a_amax + (d.amax_unit > 0 ? m / d.amax_unit : z0);
      asm volatile("global_load_dword %0, %1, off" : "=v"(a_max[i]) : "v"(tp) : "memory");
    }
  }
  // W pieces of this wavefront: piece c = wave + 8 i is rows 16 g .. 16 g + 15 of plane p, c = p (BN / 16) + g
  const u16* wptr[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int c = wave + 8 * i, p = c / (BN / 16), g = c % (BN / 16);
    const int row = g * 16 + (lane >> 2);
    int n = tn * BN + row;
    n = n < d.N ? n : d.N - 1;
    wptr[i] = W3 + (int64_t)n * SP * d.ldw + p * 32 + (((lane & 3) ^ wswz_pp(row)) << 3);
  }
  const int woff0 = lr * 64 + ((lq ^ wswz_pp(lr)) << 4);   // fragment (j, p) sits at + p WPLANE + j 1024

  const int nk = d.K / BK;
  int wk = 0, wst = 0;                       // W stream: next k to fetch, stage it goes to
  int64_t a_koff = 0;                        // A stream: element offset of the next K tile (two-level K addressing)
  int a_rem = 0;
  f32x4 araw[NR][MI][2];
  u32x4 af[MI][NP];
  u32x4 wf[NI][NP];
  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto issue_w = [&]() {
    unsigned char* dst = smem + wst * STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < PPW; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wptr[i] + SP * wk),
                                       (__attribute__((address_space(3))) void*)(dst + i * 8192), 16, 0, 0);
    wk += BK;
    wst = wst + 1 == S ? 0 : wst + 1;
  };
  auto issue_a = [&](auto rc) {
    constexpr int R = decltype(rc)::value;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      asm_load16<0>(araw[R][i][0], aptr[i] + a_koff);
      asm_load16<64>(araw[R][i][1], aptr[i] + a_koff);
    }
    a_koff += BK;
    a_rem += BK;
    if (a_rem == d.kc) { a_rem = 0; a_koff += d.ldk - d.kc; }
  };

  // ---- prologue: W tiles 0 .. S-2, A tiles 0 .. AD-1; tile 0 of W must be visible to everybody ----
  // W tiles 0 .. S-2 first, then A tiles 0 .. AD-1; the wait below leaves only A tiles 1 .. AD-1 in flight, which makes
  // the in-order VMEM queue of the first real segments at least as drained as the steady state assumes.  The common
  // case (nk > AHEAD) has no conditionals between here and the end of the steady-state loop.
  const bool deep = nk > AHEAD;
  static_assert(AD <= S - 1, "the W prologue is issued first");
  if (deep) {
#pragma unroll
    for (int s = 0; s < S - 1; ++s) issue_w();
  } else {
#pragma unroll
    for (int s = 0; s < S - 1; ++s)
      if (s < nk) issue_w();
  }
  if (deep) {
    static_for<AD>([&](auto ac) { issue_a(ac); });
    wait_vm_only<(AD - 1) * 2 * MI>();
  } else {
    static_for<AD>([&](auto ac) {
      if (decltype(ac)::value < nk) issue_a(ac);
    });
    wait_vm_only<0>();
  }
  if constexpr (NP <= 2) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      asm volatile("" : "+v"(a_max[i]));     // the trackers were the oldest loads in the queue: landed by now
      h2_scale(a_max[i], a_scale[i], row_inv[i]);
    }
  }
  __builtin_amdgcn_s_barrier();
  if (grp) __builtin_amdgcn_s_barrier();     // group 1 runs one interval behind

  int rst = 0;                               // stage of the tile being read
  // one K tile of this wavefront: load segment, barrier, compute segment, barrier.  FULL = steady state (tile t + AHEAD
  // exists: both issues happen, INFL operations stay in flight); otherwise the tail (conditional issues, full drain).
  auto segment = [&](auto rc, auto fullc, int t) {
    constexpr int R = decltype(rc)::value;
    constexpr bool FULL = decltype(fullc)::value;
    // -------- load segment of tile t --------
    {
      const unsigned char* base = smem + rst * STAGE + woff0;
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int p = 0; p < NP; ++p) wf[j][p] = *reinterpret_cast<const u32x4*>(base + p * WPLANE + j * 1024);
      rst = rst + 1 == S ? 0 : rst + 1;
    }
    if constexpr (FULL) {
      issue_w();
      issue_a(std::integral_constant<int, (R + AD) % NR>{});
      wait_vm_tied<INFL>(araw[R]);
    } else {
      if (t + S - 1 < nk) issue_w();
      if (t + AD < nk) issue_a(std::integral_constant<int, (R + AD) % NR>{});
      wait_vm_tied<0>(araw[R]);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      if constexpr (NP == 3) {
        bf16x8 h_, m_, l_;
        split8(araw[R][i][0], araw[R][i][1], h_, m_, l_);
        af[i][0] = __builtin_bit_cast(u32x4, h_);
        af[i][1] = __builtin_bit_cast(u32x4, m_);
        af[i][2] = __builtin_bit_cast(u32x4, l_);
      } else if constexpr (NP == 2) {
        split8_h2(araw[R][i][0], araw[R][i][1], a_scale[i], af[i][0], af[i][1]);
      } else {
        cvt8_h1(araw[R][i][0], araw[R][i][1], a_scale[i], af[i][0]);
      }
    }
    wait_lgkm0();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // -------- compute segment of tile t: MFMAs only --------
    if constexpr (NP == 3) {
      constexpr int PW[6] = {2, 0, 1, 1, 0, 0}, PA[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) acc[i][j] = mfma_np<NP>(wf[j][PW[q]], af[i][PA[q]], acc[i][j]);
    } else if constexpr (NP == 2) {
      constexpr int PW[3] = {1, 0, 0}, PA[3] = {0, 1, 0};   // lo*hi hi*lo hi*hi
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) acc[i][j] = mfma_np<NP>(wf[j][PW[q]], af[i][PA[q]], acc[i][j]);
    } else {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = mfma_np<NP>(wf[j][0], af[i][0], acc[i][j]);
    }
    __builtin_amdgcn_sched_barrier(0);
    // group 1 skips its very last barrier: both groups execute 2 nk + 1 of them
    if (FULL || !(grp && t + 1 == nk)) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  int t0 = 0;
  if (deep)
    for (; t0 + NR - 1 + AHEAD < nk; t0 += NR)
      static_for<NR>([&](auto rc) { segment(rc, std::true_type{}, t0 + decltype(rc)::value); });
  for (; t0 < nk; t0 += NR)
    static_for<NR>([&](auto rc) {
      const int t = t0 + decltype(rc)::value;
      if (t < nk) segment(rc, std::false_type{}, t);
    });

  // column vectors of the epilogue: a wavefront-private scratch BEHIND the W stages (group 1 may still be multiplying)
  float* lds_cols = reinterpret_cast<float*>(smem + S * STAGE) + wave * 3 * BN;
  gemm_epilogue<BM, BN, TM, BN, MI, NI, true>(d, acc, tm, tn, wave, 0, lr, lq, cz, bz, z0, row_inv, NP <= 2 ? d.col_scale : nullptr,
                                        lds_cols);
}

// ---------------------------------------------------------------------------------------------------------------------
// Second form ("pq"): the W fragments are STREAMED through a small register ring inside the compute segment instead of
// being read wholesale in the load segment.  Measured on the first form (profiles/r3_gemm_pp_probe.txt): time per launch
// = 300 us + 0.70 us x K at M = 149 k, N = 1024, i.e. the loop ran at ~52 % of the matrix pipe — its load segment
// (16 fragment reads + DMA + A loads + wait + 50 VALU of split) was LONGER than the 768-cycle compute segment of its
// partner.  Here the load segment keeps only the VMEM issue, the wait and the split; the compute segment multiplies
// column-block pairs while the ds_reads of the next pair are in flight (ring of RD pairs, first RD-1 pairs read at the
// end of the load segment).  64 registers of fragments become 16 x RD, which is what lets the tile grow to 256 x 192 /
// 256 x 256 (128 accumulator registers): per K tile 96 MFMAs per wavefront against the same A traffic and split work.
// Stage t % S is now read during the compute segments of tile t (I(2t+1), I(2t+2)), so it is refilled with tile t+S by
// the load segments of tile t+2 (W runs S-2 tiles ahead) and S >= 4.
template <int BN, int NP, int S, int AD, int RD>
__global__ __launch_bounds__(512) void gemm_pq_kernel(const dzn_gemm_desc d) {
  constexpr int BM = 256, TM = 32, MI = 2, NI = BN / 16, BK = 32, NPAIR = NI / 2;
  constexpr int SP = NP == 3 ? 3 : 2;
  constexpr int WPLANE = BN * 64, STAGE = NP * WPLANE;
  constexpr int NPIECE = STAGE / 1024;
  static_assert(NPIECE % 8 == 0 && NI % 2 == 0, "tile shape");
  constexpr int PPW = NPIECE / 8;
  // A is loaded ONE tile ahead into a single register set: a load segment first waits for its own tile (issued by the
  // previous segment, after that segment's W pieces: so only the PPW pieces it issues itself stay in flight), splits it
  // into the fp16 fragments, and only then issues the next tile's loads into the same registers.
  constexpr int AHEAD = S - 2;
  static_assert(S >= 4 && AD == 1 && RD >= 2 && RD <= NPAIR, "pipeline depths");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int lr = lane & 15, lq = lane >> 4;
  const int tilesN = (d.N + BN - 1) / BN;
  int tile;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = tile / tilesN, tn = tile % tilesN;
  const int z = blockIdx.y;
  const int z0 = z / d.zdiv, z1 = z - z0 * d.zdiv;
  const float* __restrict__ A = d.A + z0 * d.a_z0 + z1 * d.a_z1;
  const u16* __restrict__ W3 =
      reinterpret_cast<const u16*>(NP == 3 ? d.W3 : d.W2h) + SP * (z0 * d.w_z0 + z1 * d.w_z1);
  const int64_t cz = z0 * d.c_z0 + z1 * d.c_z1;
  const int64_t bz = z0 * d.b_z0 + z1 * d.b_z1;

  float a_scale[MI], row_inv[MI], a_max[MI];
  const float* aptr[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    int m = tm * BM + wave * TM + i * 16 + lr;
    m = m < d.M ? m : d.M - 1;
    a_scale[i] = row_inv[i] = 1.f;
    a_max[i] = 0.f;
    aptr[i] = A + (d.a_rowoff ? (int64_t)d.a_rowoff[m] : (int64_t)m * d.lda) + lq * 4;
    asm volatile("" : "+v"(aptr[i]));
    if constexpr (NP <= 2) {
      const float* tp = d.a_amax + (d.amax_unit > 0 ? m / d.amax_unit : z0);
      asm volatile("global_load_dword %0, %1, off" : "=v"(a_max[i]) : "v"(tp) : "memory");
    }
  }
  const u16* wptr[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int c = wave + 8 * i, p = c / (BN / 16), g = c % (BN / 16);
    const int row = g * 16 + (lane >> 2);
    int n = tn * BN + row;
    n = n < d.N ? n : d.N - 1;
    wptr[i] = W3 + (int64_t)n * SP * d.ldw + p * 32 + (((lane & 3) ^ wswz_pp(row)) << 3);
  }
  const int woff0 = lr * 64 + ((lq ^ wswz_pp(lr)) << 4);

  const int nk = d.K / BK;
  int wk = 0, wst = 0;
  int64_t a_koff = 0;
  int a_rem = 0;
  f32x4 araw[MI][2];
  u32x4 af[MI][NP];
  u32x4 ring[RD][2][NP];                     // W fragments of RD column-block pairs
  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto issue_w = [&]() {
    unsigned char* dst = smem + wst * STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < PPW; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wptr[i] + SP * wk),
                                       (__attribute__((address_space(3))) void*)(dst + i * 8192), 16, 0, 0);
    wk += BK;
    wst = wst + 1 == S ? 0 : wst + 1;
  };
  auto issue_a = [&]() {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      asm_load16<0>(araw[i][0], aptr[i] + a_koff);
      asm_load16<64>(araw[i][1], aptr[i] + a_koff);
    }
    a_koff += BK;
    a_rem += BK;
    if (a_rem == d.kc) { a_rem = 0; a_koff += d.ldk - d.kc; }
  };

  const bool deep = nk > AHEAD;
  // prologue: W tiles 0 .. S-3, then A tile 0; W tile 0 must be visible to everybody before the first segment
  if (deep) {
#pragma unroll
    for (int s = 0; s < S - 2; ++s) issue_w();
  } else {
#pragma unroll
    for (int s = 0; s < S - 2; ++s)
      if (s < nk) issue_w();
  }
  issue_a();
  wait_vm_only<2 * MI>();                    // everything but A tile 0 (the first segment waits for that itself)
  if constexpr (NP <= 2) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      asm volatile("" : "+v"(a_max[i]));
      h2_scale(a_max[i], a_scale[i], row_inv[i]);
    }
  }
  __builtin_amdgcn_s_barrier();
  if (grp) __builtin_amdgcn_s_barrier();

  int rst = 0;
  auto read_pair = [&](const unsigned char* base, auto slotc, const int pr) {
    constexpr int SL = decltype(slotc)::value;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int p = 0; p < NP; ++p)
        ring[SL][jj][p] = *reinterpret_cast<const u32x4*>(base + p * WPLANE + (2 * pr + jj) * 1024);
  };
  auto segment = [&](int, auto fullc, int t) {
    constexpr bool FULL = decltype(fullc)::value;
    const unsigned char* base = smem + rst * STAGE + woff0;
    rst = rst + 1 == S ? 0 : rst + 1;
    // -------- load segment of tile t --------
    // (1) A tile t: issued last by the previous segment -> nothing of ours is younger
    wait_vm_tied<0>(araw);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      if constexpr (NP == 3) {
        bf16x8 h_, m_, l_;
        split8(araw[i][0], araw[i][1], h_, m_, l_);
        af[i][0] = __builtin_bit_cast(u32x4, h_);
        af[i][1] = __builtin_bit_cast(u32x4, m_);
        af[i][2] = __builtin_bit_cast(u32x4, l_);
      } else if constexpr (NP == 2) {
        split8_h2(araw[i][0], araw[i][1], a_scale[i], af[i][0], af[i][1]);
      } else {
        cvt8_h1(araw[i][0], araw[i][1], a_scale[i], af[i][0]);
      }
    }
    // the split must have consumed the raw registers before the next loads are issued into them
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int p = 0; p < NP; ++p) asm volatile("" : "+v"(af[i][p]));
    // (2) W tile t+S-2 and A tile t+1; everything the previous segments issued has landed (in-order return), and the
    //     barrier below publishes this wavefront's pieces of tile t+1 to the other group
    if constexpr (FULL) {
      issue_w();
      issue_a();
    } else {
      if (t + S - 2 < nk) issue_w();
      if (t + 1 < nk) issue_a();
    }
    // (3) first fragment pairs of tile t
    static_for<RD - 1>([&](auto pc) { read_pair(base, pc, decltype(pc)::value); });
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // -------- compute segment of tile t: MFMAs + the fragment reads of the pairs ahead --------
    static_for<NPAIR>([&](auto pc) {
      constexpr int P = decltype(pc)::value;
      if constexpr (P + RD - 1 < NPAIR) read_pair(base, std::integral_constant<int, (P + RD - 1) % RD>{}, P + RD - 1);
      constexpr int SL = P % RD;
      constexpr int NQ = NP == 3 ? 6 : NP == 2 ? 3 : 1;
      constexpr int PW6[6] = {2, 0, 1, 1, 0, 0}, PA6[6] = {0, 2, 1, 0, 1, 0};
      constexpr int PW3[3] = {1, 0, 0}, PA3[3] = {0, 1, 0};
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int pw = NP == 3 ? PW6[q] : NP == 2 ? PW3[q] : 0, pa = NP == 3 ? PA6[q] : NP == 2 ? PA3[q] : 0;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int i = 0; i < MI; ++i) acc[i][2 * P + jj] = mfma_np<NP>(ring[SL][jj][pw], af[i][pa], acc[i][2 * P + jj]);
      }
    });
    __builtin_amdgcn_sched_barrier(0);
    if (FULL || !(grp && t + 1 == nk)) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  int t0 = 0;
  if (deep)
    for (; t0 + AHEAD < nk; ++t0) segment(0, std::true_type{}, t0);
  for (; t0 < nk; ++t0) segment(0, std::false_type{}, t0);

  float* lds_cols = reinterpret_cast<float*>(smem + S * STAGE) + wave * 3 * BN;
  gemm_epilogue<BM, BN, TM, BN, MI, NI, true>(d, acc, tm, tn, wave, 0, lr, lq, cz, bz, z0, row_inv, NP <= 2 ? d.col_scale : nullptr,
                                              lds_cols);
}

template <int BN, int NP, int S, int AD, int RD>
int launch_pq_cfg(const dzn_gemm_desc& d, hipStream_t s) {
  constexpr int BM = 256;
  const int tilesM = (d.M + BM - 1) / BM, tilesN = (d.N + BN - 1) / BN;
  const size_t lds = (size_t)S * NP * BN * 64 + 8 * 3 * BN * sizeof(float);
  static_assert((size_t)S * NP * BN * 64 + 8 * 3 * BN * sizeof(float) <= 160 * 1024, "LDS");
  auto kern = gemm_pq_kernel<BN, NP, S, AD, RD>;
  static unsigned long long attr_mask = 0;
  if (first_use_on_device(attr_mask))
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  dim3 grid(tilesM * tilesN, d.nz > 0 ? d.nz : 1, 1);
  int pid = -1;
  if (prof_enabled()) {
    char cls[64];
    static const bool by_shape = getenv("DZN_PROFILE_SHAPES") != nullptr;
    const char* pn = NP == 3 ? "f32s" : NP == 2 ? "f32h" : "f16";
    if (by_shape) snprintf(cls, sizeof(cls), "gemm_%s_pq256x%d M%d N%d K%d z%d", pn, BN, d.M, d.N, d.K, d.nz);
    else snprintf(cls, sizeof(cls), "gemm_%s_pq256x%d", pn, BN);
    const double fl = d.alg_flops > 0 ? d.alg_flops * d.nz : 2.0 * d.M * d.N * d.K * d.nz;
    pid = prof_begin(s, cls, fl, gemm_alg_bytes(d, NP * 2));
  }
  hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, d);
  prof_end(pid, s);
  if (hipGetLastError() != hipSuccess) return DZN_E_HIP;
  if (d.stat_partial && d.stat_final)
    return launch_stats_finalize(d.stat_partial, d.M, tilesN, d.stat_C, d.stat_eps, d.stat_final, s);
  return DZN_OK;
}

template <int BN, int NP, int S, int AD>
int launch_pp_cfg(const dzn_gemm_desc& d, hipStream_t s) {
  constexpr int BM = 256;
  const int tilesM = (d.M + BM - 1) / BM, tilesN = (d.N + BN - 1) / BN;
  const size_t lds = (size_t)S * NP * BN * 64 + 8 * 3 * BN * sizeof(float);   // W stages + epilogue column scratch
  auto kern = gemm_pp_kernel<BN, NP, S, AD>;
  static unsigned long long attr_mask = 0;
  if (first_use_on_device(attr_mask))
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  dim3 grid(tilesM * tilesN, d.nz > 0 ? d.nz : 1, 1);
  int pid = -1;
  if (prof_enabled()) {
    char cls[64];
    static const bool by_shape = getenv("DZN_PROFILE_SHAPES") != nullptr;
    const char* pn = NP == 3 ? "f32s" : NP == 2 ? "f32h" : "f16";
    if (by_shape) snprintf(cls, sizeof(cls), "gemm_%s_pp256x%d M%d N%d K%d z%d", pn, BN, d.M, d.N, d.K, d.nz);
    else snprintf(cls, sizeof(cls), "gemm_%s_pp256x%d", pn, BN);
    const double fl = d.alg_flops > 0 ? d.alg_flops * d.nz : 2.0 * d.M * d.N * d.K * d.nz;
    pid = prof_begin(s, cls, fl, gemm_alg_bytes(d, NP * 2));
  }
  hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, d);
  prof_end(pid, s);
  if (hipGetLastError() != hipSuccess) return DZN_E_HIP;
  if (d.stat_partial && d.stat_final)
    return launch_stats_finalize(d.stat_partial, d.M, tilesN, d.stat_C, d.stat_eps, d.stat_final, s);
  return DZN_OK;
}

}  // namespace

// cfg: first form "pp128" / "pp64" (+ "s4" for 4 stages, "a2" for A two tiles ahead), e.g. "pp128s4a2";
//      streamed form "pq128" / "pq192" / "pq256" (+ "r3": ring of 3 pairs, "a2": A two tiles ahead with 5 stages)
int launch_gemm_pp(const dzn_gemm_desc& d, hipStream_t s, int np, const char* cfg) {
  if (cfg && !strncmp(cfg, "pq", 2)) {
    const bool r3 = strstr(cfg, "r3");
    const int bn = atoi(cfg + 2);
    if (np == 2) {
      if (bn == 128) return r3 ? launch_pq_cfg<128, 2, 4, 1, 3>(d, s) : launch_pq_cfg<128, 2, 4, 1, 2>(d, s);
      if (bn == 192) return r3 ? launch_pq_cfg<192, 2, 4, 1, 3>(d, s) : launch_pq_cfg<192, 2, 4, 1, 2>(d, s);
      // (256 x 256: 128 accumulator registers + ring + fragments do not fit 256 VGPRs without spilling)
    }
    if (np == 1 && bn == 128) return launch_pq_cfg<128, 1, 4, 1, 2>(d, s);
    if (np == 3 && bn == 128) return launch_pq_cfg<128, 3, 4, 1, 2>(d, s);
    return DZN_E_INVALID;
  }
  const bool n64 = cfg && !strncmp(cfg, "pp64", 4);
  const bool s4 = cfg && strstr(cfg, "s4"), a2 = cfg && strstr(cfg, "a2");
  if (np == 2) {
    if (n64) return s4 ? launch_pp_cfg<64, 2, 4, 2>(d, s) : launch_pp_cfg<64, 2, 3, 1>(d, s);
    if (s4 && a2) return launch_pp_cfg<128, 2, 4, 2>(d, s);
    return launch_pp_cfg<128, 2, 3, 1>(d, s);
  }
  if (np == 1 && !n64) return launch_pp_cfg<128, 1, 3, 1>(d, s);
  if (np == 3 && !n64) return launch_pp_cfg<128, 3, 3, 1>(d, s);
  return DZN_E_INVALID;
}
