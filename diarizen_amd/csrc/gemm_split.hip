// gemm_split.hip — fp32 contraction on the bf16 matrix pipe: exact 3-way operand split
// ("f32s" engine mode, DZN_PREC_F32_SPLIT).
//
// Every fp32 value x is written as x = hi + mid + lo with hi = bf16(x), mid = bf16(x - hi),
// lo = bf16(x - hi - mid) (round-to-nearest at each level; the two subtractions are exact and
// lo needs <= 8 significant bits, so the decomposition is EXACT).  A product a*w is then
//     a_hi w_hi + (a_hi w_mid + a_mid w_hi) + (a_hi w_lo + a_mid w_mid + a_lo w_hi)  [+ O(2^-24) terms]
// i.e. six v_mfma_f32_16x16x32_bf16 products accumulated in fp32; the three dropped cross terms
// (mid*lo, lo*mid, lo*lo) are bounded by 2^-22 |a w| — the size of one or two fp32 roundings of the
// product.  6 bf16 MFMAs of 16 cycles replace 8 fp32 MFMAs (16x16x4) of 32 cycles for the same
// 16x16x32 block: 2.67x less matrix-pipe time at fp32-grade accuracy (tests/test_ops_gpu.py
// measures both kernels against a float64 product).
//
// Same contract as gemm.hip (dzn_gemm_desc, fused epilogue).  Data movement:
//   A (activations) : fp32 in HBM, fp32 tile in LDS by LDS-DMA (128-B rows, XOR-swizzled exactly
//       like the fp32 kernel); each wavefront splits the fragments it reads in registers
//       (v_cvt_pk_bf16_f32 / shifts / packed fp32 subtracts, overlapped with the MFMAs).
//   W (weights)     : split ONCE when the weights are packed (dzn_op_split_weights) into
//       [N][K/32][plane 3][32] bf16; the 32 k of a block are stored in the order the fragment
//       reads want (lane group q owns k = 4q..4q+3 and 16+4q..16+4q+3, matching the two 16-B
//       slots q and 4+q of the fp32 A row), so every fragment is one ds_read_b128.  Each plane
//       of a W tile is a [BN][64 B] LDS image; slot s of row r is stored at s ^ g((r>>2)&3),
//       g = (0,2,3,1), which makes the four 16-lane groups of ds_read_b128 conflict free.
// Requires K % 32 == 0 and kc % 32 == 0 (else the caller falls back to the fp32 MFMA kernel).
//
// NP = 2 ("f32h", DZN_PREC_F32_H2): the same kernel with a TWO-term fp16 split and THREE products
// (hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16) — the error-corrected tensor-core scheme of Ootomo &
// Yokota (and of "3xTF32": fp16 and TF32 both carry 11 significant bits).  x*s = hi + lo + r with
// hi = fp16(x*s), lo = fp16(x*s - hi), |r| <= 2^-22 |x*s|; the dropped terms (lo*lo and the two residuals) are
// <= 3 * 2^-22 |a w| — the size of a few fp32 roundings of the product — and half the matrix-pipe work of
// NP = 3.  fp16's 5-bit exponent is handled by EXACT power-of-two scaling on both sides: the weights are
// scaled per output row when they are split (max |w| of the row lands in [2^14, 2^15), dzn_op_split_weights_h2),
// the activations by s = 2^(14 - floor(log2 amax)) from the running |max| the PRODUCER of the tensor tracked
// (dzn_gemm_desc.a_amax / c_amax); the epilogue multiplies the accumulator by the exact inverse powers of two.
// Elements more than 2^17 below the tensor's maximum keep a lo term in fp16's subnormal range: their ABSOLUTE
// error stays <= 2^-25 * max / 2^14, i.e. below fp32 resolution of any dot product that contains the maximum.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "checked.h"
#include "common.h"
#include "gemm_epilogue.h"
#include "split.h"

DZN_CHECKED_TU(gemm_split)

namespace {

__device__ __forceinline__ int wswz(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }

// s_waitcnt vmcnt(N) lgkmcnt(0), expcnt left at its maximum (gfx9 encoding: vmcnt = [3:0] + [15:14])
template <int N>
__device__ __forceinline__ void wait_vm_lgkm0() {
  static_assert(N >= 0 && N < 64, "vmcnt range");
  __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | (0x7 << 4));
}

// BM x BN tile per workgroup of WGM x WGN wavefronts, S LDS stages of one 32-k tile each.
//
// Pipeline.  The matrix pipe retires a K tile in ~1.5k cycles per wavefront-tile while an LDS-DMA
// fill needs 1.1-1.7 us (2.6k-4k cycles) to land, and a wavefront that has just passed a barrier
// needs ~0.5k cycles of ds_read + split before its first MFMA.  So:
//   * tiles are prefetched S tiles ahead (raw s_barrier + s_waitcnt vmcnt(N) that leaves the
//     younger tiles in flight; __syncthreads would drain them);
//   * the barrier of K tile kt sits in the MIDDLE of the tile's MFMAs: the first half of the
//     rows is multiplied, then [wait tile kt+1, barrier, refill the stage of tile kt], then the
//     fragments of tile kt+1 are read into the second register set while the second half of
//     tile kt is multiplied — the matrix pipe has work queued across the barrier.
// ABL != 0: ABLATION probes for scripts/bench_gemm_cfgs.py (wrong results on purpose; never launched by the engines):
//   1 = no operand split (raw bits as fragments), 2 = no MFMAs, 3 = no steady-state LDS-DMA refills, 4 = no barrier,
//   5 = no fragment reads after the first tile
// RPF (r5): the residual of the whole wavefront tile is requested before the first operand tile (gemm_prefetch_residual): the
// short-K launches' epilogue is then stores only.
template <int BM, int BN, int WGM, int WGN, int S, int NP, int OCC = 1, int ABL = 0, bool RPF = false>
__global__ __launch_bounds__(WGM * WGN * 64, OCC) void gemm_split_kernel(const dzn_gemm_desc d, const int ngroups) {
  constexpr int NW = WGM * WGN;           // wavefronts per workgroup
  constexpr int BK = 32;
  constexpr int TM = BM / WGM, TN = BN / WGN;
  constexpr int MI = TM / 16, NI = TN / 16, MH = MI / 2;
  constexpr int RB = NW * 1024;           // bytes per LDS-DMA round (1 KiB per wavefront)
  constexpr int ACH = BM * 128 / RB;      // rounds of the A tile (BM rows x 128 B)
  constexpr int WROWS = NW * 16;          // rows of one W plane per round (64-B rows)
  constexpr int WR = (BN + WROWS - 1) / WROWS;
  constexpr int SP = NP == 3 ? 3 : 2;     // planes STORED per weight row (NP = 1 reads the leading one of two)
  constexpr int ABYTES = BM * 128, WPLANE = BN * 64, BUF = ABYTES + NP * WPLANE;
  constexpr int LPT = ACH + NP * WR;      // LDS-DMA instructions per thread per tile (NP fewer for the
                                          // wavefronts that sit out a partial last W round)
  constexpr bool WPART = BN % WROWS != 0;
  static_assert(BM * 128 % RB == 0, "A tile must be whole rounds");
  static_assert(MI % 2 == 0, "two row halves per wavefront tile");
  static_assert(S >= 2 && (S - 1) * LPT < 64, "vmcnt range");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: LDS-DMA bases stay scalar
  const int wm = wave / WGN, wn = wave % WGN;
  const int tilesN = (d.N + BN - 1) / BN;
  int t;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // (r4) COLUMN GROUPS.  t walks the tiles XCD by XCD (contiguous ranges).  With ngroups == 1 that order is row-block-major:
  // an XCD works on a few row blocks with ALL their column tiles, so it streams the whole weight-plane set (N x K x 2 NP bytes)
  // once per row block — and from ~4 MB on that set no longer survives in the XCD's 4 MB L2 between two row blocks
  // (profiles/r4_gemm_refetch_probe.txt: fabric reads 1.02 x the algorithmic bytes at N = 128, 2.1 x at N = K = 1024, 4.7 x at
  // N = 2048; the excess is 0.27 / 0.69 of the plane set PER ROW BLOCK).  With ngroups = G the column tiles are cut into G
  // contiguous groups and the order is group-major, so an XCD meets only 1 / G of the planes (they stay in its L2) while a row
  // block of A is fetched by G XCDs instead of one; the launcher picks G from that trade (choose_column_groups).  A
  // permutation of the tile index: every tile is computed exactly once, by the same code — results do not change.
  int tm, tn;
  if (ngroups <= 1) {
    tm = t / tilesN;
    tn = t % tilesN;
  } else {
    const int tilesM_ = (int)gridDim.x / tilesN;
    const int base = tilesN / ngroups, rem = tilesN % ngroups;
    int g = 0, cg = base + (rem > 0), cs = 0, r = t;
    while (r >= tilesM_ * cg) {     // ngroups <= 8 iterations, scalar
      r -= tilesM_ * cg;
      cs += cg;
      ++g;
      cg = base + (g < rem);
    }
    tm = r / cg;
    tn = cs + r - tm * cg;
  }
  const int z = blockIdx.y;
  int z0 = z / d.zdiv;
  const int z1 = z - z0 * d.zdiv;
  if (d.z_list) {   // device-chosen subset of the batch (dzn_gemm_desc.z_count / z_list)
    if (z0 >= d.z_count[0]) return;
    z0 = d.z_list[z0];
  }
  const float* __restrict__ A = d.A + z0 * d.a_z0 + z1 * d.a_z1;
  const u16* __restrict__ W3 =
      reinterpret_cast<const u16*>(NP == 3 ? d.W3 : d.W2h) + SP * (z0 * d.w_z0 + z1 * d.w_z1);
  // NP = 2: exact power-of-two scale of every A row from the |max| tracker of the UNIT (window / image) the row
  // belongs to — unit = m / amax_unit, or the z batch index when amax_unit == 0 — so that a window's result does
  // not depend on what else is in the batch
  float a_scale[MI], row_inv[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) a_scale[i] = row_inv[i] = 1.f;
  if constexpr (NP <= 2) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      int m = tm * BM + wm * TM + i * 16 + (lane & 15);
      m = m < d.M ? m : d.M - 1;
      const int unit = d.amax_unit > 0 ? m / d.amax_unit : z0;
      DZN_CHECK(d.amax_count <= 0 || (unit >= 0 && unit < d.amax_count), 0x101, unit);   // tracker index inside its array
      h2_scale(d.a_amax[unit], a_scale[i], row_inv[i]);
    }
  }
  // NP = 1 with a folded LayerNorm (d.ln_centered): the row mean is subtracted before the fp16 rounding; |x - mean| <=
  // 2 amax, so the scale gives up one bit of headroom
  float a_mean[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) a_mean[i] = 0.f;
  if constexpr (NP == 1) {
    if (d.ln_centered && d.ln_stats) {
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        int m = tm * BM + wm * TM + i * 16 + (lane & 15);
        m = m < d.M ? m : d.M - 1;
        a_mean[i] = d.ln_stats[2 * (int64_t)m];
        a_scale[i] *= 0.5f;
        row_inv[i] *= 2.f;
      }
    }
  }
  const int64_t cz = z0 * d.c_z0 + z1 * d.c_z1;
  const int64_t bz = z0 * d.b_z0 + z1 * d.b_z1;

  // A: thread -> (row = tid/8 + 8 NW i, physical slot tid%8), logical chunk = slot ^ ((row>>1)&7)
  const int r0 = tid >> 3;
  const int csw = (tid & 7) ^ ((r0 >> 1) & 7);
  const float* aptr[ACH];   // per-thread source of K tile 0; a K tile adds a uniform offset
#pragma unroll
  for (int i = 0; i < ACH; ++i) {
    int m = tm * BM + r0 + 8 * NW * i;
    m = m < d.M ? m : d.M - 1;
    aptr[i] = A + (d.a_rowoff ? (int64_t)d.a_rowoff[m] : (int64_t)m * d.lda) + csw * 4;
  }
  // W planes: thread -> (row = 16 wave + lane/4 + 16 NW i, physical slot lane%4); rows past N re-read
  // row N-1 (their accumulators are never stored); a partial last round is fetched by the first waves only
  const bool wfull = !WPART || (WR - 1) * WROWS + wave * 16 < BN;   // wave-uniform
  const int wr0 = wave * 16 + (lane >> 2);
  const int wsw = (lane & 3) ^ wswz(wr0);
  const u16* wptr[WR];
#pragma unroll
  for (int i = 0; i < WR; ++i) {
    int n = tn * BN + wr0 + WROWS * i;
    n = n < d.N ? n : d.N - 1;
    wptr[i] = W3 + (int64_t)n * SP * d.ldw + wsw * 8;
  }

  // the next K tile to fetch: k index and its A element offset (two-level K addressing), advanced
  // incrementally on the scalar unit
  int ik = 0, irem = 0;
  int64_t ikoff = 0;
  auto issue = [&](int stage) {
    unsigned char* sA = smem + stage * BUF + wave * 1024;
    unsigned char* sW = smem + stage * BUF + ABYTES + wave * 1024;
    DZN_CHECK(stage >= 0 && stage < S && ik < d.K, 0x102, stage);                              // a stage of the ring, a k tile of the operand
    DZN_CHECK(wave * 1024 + (ACH - 1) * RB + 1024 <= ABYTES, 0x103, wave);                      // A fill stays inside the A image
    DZN_CHECK(!wfull || wave * 1024 + (WR - 1) * RB + 1024 <= WPLANE, 0x104, wave);                  // W fill stays inside its plane image
#pragma unroll
    for (int i = 0; i < ACH; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(aptr[i] + ikoff),
                                       (__attribute__((address_space(3))) void*)(sA + i * RB), 16, 0, 0);
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int i = 0; i < WR; ++i)
        if (i + 1 < WR || wfull)
          __builtin_amdgcn_global_load_lds(
              (const __attribute__((address_space(1))) void*)(wptr[i] + SP * ik + p * 32),
              (__attribute__((address_space(3))) void*)(sW + p * WPLANE + i * RB), 16, 0, 0);
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
  f32x4 rpre[RPF ? MI : 1][RPF ? NI : 1];
  bool use_rpre = false;
  if constexpr (RPF) {
    use_rpre = d.R != nullptr && gemm_epilogue_vec(d, cz, bz);
    if (use_rpre) gemm_prefetch_residual<BM, BN, TM, TN, MI, NI>(d, rpre, tm, tn, wm, wn, lr, lq, cz);
  }

  // per-lane LDS byte offsets of the fragments inside a stage
  int woff[NI], aoff0[MI], aoff1[MI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int row = wn * TN + j * 16 + lr;
    woff[j] = ABYTES + row * 64 + ((lq ^ wswz(row)) << 4);
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int row = wm * TM + i * 16 + lr;
    const int sw = (row >> 1) & 7;
    aoff0[i] = row * 128 + ((lq ^ sw) << 4);
    aoff1[i] = row * 128 + (((4 + lq) ^ sw) << 4);
  }
#pragma unroll
  for (int j = 0; j < NI; ++j) DZN_CHECK(woff[j] >= ABYTES && woff[j] + (NP - 1) * WPLANE + 16 <= BUF, 0x105, woff[j]);   // fragment reads inside the stage
#pragma unroll
  for (int i = 0; i < MI; ++i) DZN_CHECK(aoff0[i] + 16 <= ABYTES && aoff1[i] + 16 <= ABYTES, 0x106, aoff1[i]);
  DZN_CHECK(tm * BM < d.M && tn * BN < d.N, 0x107, t);                                            // the tile exists
  auto read_w = [&](int stage, u32x4 (&wf)[NI][NP]) {
    const unsigned char* base = smem + stage * BUF;
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int p = 0; p < NP; ++p) wf[j][p] = *reinterpret_cast<const u32x4*>(base + p * WPLANE + woff[j]);
  };
  auto read_a = [&](int stage, f32x4 (&ar)[MI][2]) {
    const unsigned char* base = smem + stage * BUF;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      ar[i][0] = *reinterpret_cast<const f32x4*>(base + aoff0[i]);
      ar[i][1] = *reinterpret_cast<const f32x4*>(base + aoff1[i]);
    }
  };
  // the products of one 16-row block against all NI column blocks: smallest terms first, NI independent
  // accumulators between dependent MFMAs.  af[] = A terms (hi, [mid,] lo), wf[j][] = W planes (hi, [mid,] lo).
  auto mma = [&](int i, const u32x4 (&wf)[NI][NP], const u32x4 (&af)[NP]) {
    if constexpr (ABL == 2) {
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j][0] += __uint_as_float(wf[j][0][0] ^ af[0][0]);   // keep the operands live
    } else if constexpr (NP == 3) {
      constexpr int PW[6] = {2, 0, 1, 1, 0, 0}, PA[6] = {0, 2, 1, 0, 1, 0};   // lo*hi hi*lo mid*mid mid*hi hi*mid hi*hi
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = mfma_np<NP>(wf[j][PW[t]], af[PA[t]], acc[i][j]);
    } else if constexpr (NP == 2) {
      constexpr int PW[3] = {1, 0, 0}, PA[3] = {0, 1, 0};                     // lo*hi hi*lo hi*hi
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = mfma_np<NP>(wf[j][PW[t]], af[PA[t]], acc[i][j]);
    } else {                                                                  // hi*hi (DZN_PREC_F16)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = mfma_np<NP>(wf[j][0], af[0], acc[i][j]);
    }
  };
  auto split = [&](const f32x4 (&a)[2], u32x4 (&af)[NP], float sc, float mu = 0.f) {
    if constexpr (ABL == 1) {
#pragma unroll
      for (int p = 0; p < NP; ++p) af[p] = __builtin_bit_cast(u32x4, a[p & 1]);
    } else if constexpr (NP == 3) {
      bf16x8 h_, m_, l_;
      split8(a[0], a[1], h_, m_, l_);
      af[0] = __builtin_bit_cast(u32x4, h_);
      af[1] = __builtin_bit_cast(u32x4, m_);
      af[2] = __builtin_bit_cast(u32x4, l_);
    } else if constexpr (NP == 2) {
      split8_h2(a[0], a[1], sc, af[0], af[1]);
    } else {
      cvt8_h1(a[0] - mu, a[1] - mu, sc, af[0]);      // mu = 0 unless d.ln_centered (x - 0 is exact)
    }
  };

  const int nk = d.K / BK;
#pragma unroll
  for (int s = 0; s < S; ++s)
    if (s < nk) issue(s);
  // wait until all but the `T` youngest tiles of this wavefront's LDS-DMA have landed
  auto wait_tiles = [&](auto tiles) {
    constexpr int T = decltype(tiles)::value;
    if (wfull) wait_vm_lgkm0<T * LPT>();
    else wait_vm_lgkm0<T * (LPT - NP)>();
  };
  if (nk >= S) wait_tiles(std::integral_constant<int, S - 1>{});
  else wait_vm_lgkm0<0>();
  __builtin_amdgcn_s_barrier();
  u32x4 wfa[NI][NP], wfb[NI][NP];
  f32x4 ar[MI][2];
  read_w(0, wfa);
  read_a(0, ar);
  if constexpr (ABL == 5) read_w(0, wfb);
  int stage = 0;

  // one K tile: `wc` holds its W fragments, `ar` its raw A fragments; leaves tile kt+1 in (wn_, ar)
  auto step = [&](int kt, const u32x4 (&wc)[NI][NP], u32x4 (&wn_)[NI][NP]) {
    const bool more = kt + 1 < nk;
#pragma unroll
    for (int i = 0; i < MH; ++i) {
      u32x4 af[NP];
      split(ar[i], af, a_scale[i], a_mean[i]);
      mma(i, wc, af);
    }
    u32x4 af2[MI - MH][NP];
#pragma unroll
    for (int i = MH; i < MI; ++i) split(ar[i], af2[i - MH], a_scale[i], a_mean[i]);
    const int nstage = stage + 1 == S ? 0 : stage + 1;
    __builtin_amdgcn_sched_barrier(0);  // keep the second half of the MFMAs BEHIND the barrier block
    if (more) {
      // tile kt+1 landed (tiles kt+2 .. kt+S-1 may stay in flight); all my reads of tile kt retired
      if (kt + S <= nk) wait_tiles(std::integral_constant<int, S - 2>{});
      else wait_vm_lgkm0<0>();
      if constexpr (ABL != 4) __builtin_amdgcn_s_barrier();
      if constexpr (ABL != 3)
        if (kt + S < nk) issue(stage);  // every wave is past its reads of tile kt
      if constexpr (ABL != 5) {
        read_w(nstage, wn_);
        read_a(nstage, ar);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = MH; i < MI; ++i) mma(i, wc, af2[i - MH]);
    stage = nstage;
  };
  for (int kt = 0; kt < nk; kt += 2) {
    step(kt, wfa, wfb);
    if (kt + 1 < nk) step(kt + 1, wfb, wfa);
  }
  if constexpr (true) {
    // the epilogue's column vectors live in LDS (48 registers less than holding them): the stages are dead once every
    // wavefront left the loop
    __syncthreads();
    if constexpr (RPF) {
      if (use_rpre) {
        gemm_epilogue<BM, BN, TM, TN, MI, NI, true>(d, acc, tm, tn, wm, wn, lr, lq, cz, bz, z0, row_inv, NP <= 2 ? d.col_scale : nullptr,
                                                    reinterpret_cast<float*>(smem) + wave * 3 * TN, rpre);
        return;
      }
    }
    gemm_epilogue<BM, BN, TM, TN, MI, NI, true>(d, acc, tm, tn, wm, wn, lr, lq, cz, bz, z0, row_inv, NP <= 2 ? d.col_scale : nullptr,
                                                reinterpret_cast<float*>(smem) + wave * 3 * TN);
  } else {
    gemm_epilogue<BM, BN, TM, TN, MI, NI>(d, acc, tm, tn, wm, wn, lr, lq, cz, bz, z0, row_inv, NP <= 2 ? d.col_scale : nullptr);
  }
}

#ifdef DZN_TUNING
// ---- (r4) the same contraction on 32x32x16 MFMA blocks: a MEASURED NEGATIVE, kept as a DZN_TUNING probe ------------------
// VERDICT r3 item 4 asked for it.  Result (profiles/r4_gemm_m32_probe.txt, M = 149 226): 253-272 TFLOP/s where the
// 16x16x32 tile below has 285-305 on the same shapes (N = K = 1024: 270 vs 305; N 1920: 272 vs 304; K = 256: 149 vs 170),
// s_setprio around the MFMA cluster changes nothing, 2 x 2 / 256 x 128 wavefront layouts are slower still, and in the
// pipeline the step loses 4 % (1160 vs 1116 ms).  Same LDS bytes, same DMA schedule, half the MFMA instructions: the
// contraction is evidently not bound by MFMA issue or operand-register reads.  (Not isolated further: the form has 4
// independent accumulator blocks per wavefront where the 16x16 form has 16, against a 16-pass dependent latency.)
// v_mfma_f32_32x32x16_f16 does twice the flops of v_mfma_f32_16x16x32_f16 from the same 4 + 4 operand registers: half
// the operand-register reads per flop, and the instruction retires 2 x 16 passes where two 16x16x32 need 2 x 8 + issue gaps
// (guide: 2178 vs 1955 TFLOP/s fp16 micro-benchmark ceilings).  Same tile, same LDS images, same LDS-DMA schedule and the
// same weight planes as gemm_split_kernel — only the lane -> element map changes:
//   * lane (l31 = lane & 31, lh = lane >> 5) owns row / column l31 of a 32-wide block and the k subset of chunk
//     c = 2 kh + lh of the 32-k tile, kh = 0, 1 being the two 16-k MFMAs of the tile.  The weight planes keep their k
//     order (chunk c = k in {4c..4c+3} u {16+4c..16+4c+3}); the A fragment of chunk c is the fp32 slots c and 4 + c of the
//     row — exactly what the 16x16 form reads for lq = c — so both operands of an MFMA cover the same 16 k.
//   * ds_read_b128 stays conflict free: a 16-lane group of the instruction now holds 16 different rows of ONE chunk; the
//     A swizzle (slot ^ (row >> 1) & 7) and the W swizzle (slot ^ g((row >> 2) & 3)) spread those over all 16 slots.
//   * accumulators: lane holds row l31, columns 32 j + 8 g + 4 lh + (0..3) for g = 0..3: four float4 per block; the
//     epilogue sees them as 8-column blocks (gemm_epilogue<..., RS = 32, CS = 8>).
// The K tile is multiplied in its two 16-k halves with the barrier block between them (the 16x16 form splits by rows).
// NP = 2 (f32h) and NP = 1 (f16) only.
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int BM, int BN, int WGM, int WGN, int S, int NP, int OCC = 1, int PRIO = 0>
__global__ __launch_bounds__(WGM * WGN * 64, OCC) void gemm_split32_kernel(const dzn_gemm_desc d) {
  static_assert(NP == 1 || NP == 2, "fp16 forms only");
  constexpr int NW = WGM * WGN;
  constexpr int BK = 32;
  constexpr int TM = BM / WGM, TN = BN / WGN;
  constexpr int MI = TM / 32, NJ = TN / 32;
  constexpr int RB = NW * 1024;
  constexpr int ACH = BM * 128 / RB;
  constexpr int WROWS = NW * 16;
  constexpr int WR = (BN + WROWS - 1) / WROWS;
  constexpr int ABYTES = BM * 128, WPLANE = BN * 64, BUF = ABYTES + NP * WPLANE;
  constexpr int LPT = ACH + NP * WR;
  constexpr bool WPART = BN % WROWS != 0;
  static_assert(BM * 128 % RB == 0 && TM % 32 == 0 && TN % 32 == 0, "tile geometry");
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
  if (d.z_list) {
    if (z0 >= d.z_count[0]) return;
    z0 = d.z_list[z0];
  }
  const float* __restrict__ A = d.A + z0 * d.a_z0 + z1 * d.a_z1;
  const u16* __restrict__ W2 = reinterpret_cast<const u16*>(d.W2h) + 2 * (z0 * d.w_z0 + z1 * d.w_z1);
  const int l31 = lane & 31, lh = lane >> 5;
  float a_scale[MI], row_inv[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    int m = tm * BM + wm * TM + i * 32 + l31;
    m = m < d.M ? m : d.M - 1;
    h2_scale(d.a_amax[d.amax_unit > 0 ? m / d.amax_unit : z0], a_scale[i], row_inv[i]);
  }
  const int64_t cz = z0 * d.c_z0 + z1 * d.c_z1;
  const int64_t bz = z0 * d.b_z0 + z1 * d.b_z1;

  // ---- LDS-DMA sources: identical to gemm_split_kernel ----
  const int r0 = tid >> 3;
  const int csw = (tid & 7) ^ ((r0 >> 1) & 7);
  const float* aptr[ACH];
#pragma unroll
  for (int i = 0; i < ACH; ++i) {
    int m = tm * BM + r0 + 8 * NW * i;
    m = m < d.M ? m : d.M - 1;
    aptr[i] = A + (d.a_rowoff ? (int64_t)d.a_rowoff[m] : (int64_t)m * d.lda) + csw * 4;
  }
  const bool wfull = !WPART || (WR - 1) * WROWS + wave * 16 < BN;
  const int wr0 = wave * 16 + (lane >> 2);
  const int wsw = (lane & 3) ^ wswz(wr0);
  const u16* wptr[WR];
#pragma unroll
  for (int i = 0; i < WR; ++i) {
    int n = tn * BN + wr0 + WROWS * i;
    n = n < d.N ? n : d.N - 1;
    wptr[i] = W2 + (int64_t)n * 2 * d.ldw + wsw * 8;
  }
  int ik = 0, irem = 0;
  int64_t ikoff = 0;
  auto issue = [&](int stage) {
    unsigned char* sA = smem + stage * BUF + wave * 1024;
    unsigned char* sW = smem + stage * BUF + ABYTES + wave * 1024;
#pragma unroll
    for (int i = 0; i < ACH; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(aptr[i] + ikoff),
                                       (__attribute__((address_space(3))) void*)(sA + i * RB), 16, 0, 0);
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int i = 0; i < WR; ++i)
        if (i + 1 < WR || wfull)
          __builtin_amdgcn_global_load_lds(
              (const __attribute__((address_space(1))) void*)(wptr[i] + 2 * ik + p * 32),
              (__attribute__((address_space(3))) void*)(sW + p * WPLANE + i * RB), 16, 0, 0);
    ik += BK;
    irem += BK;
    ikoff += BK;
    if (irem == d.kc) { irem = 0; ikoff += d.ldk - d.kc; }
  };

  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // per-lane LDS byte offsets: [block][k half]
  int woff[NJ][2], aoff0[MI][2], aoff1[MI][2];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int row = wn * TN + j * 32 + l31;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) woff[j][kh] = ABYTES + row * 64 + (((2 * kh + lh) ^ wswz(row)) << 4);
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int row = wm * TM + i * 32 + l31;
    const int sw = (row >> 1) & 7;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      const int c = 2 * kh + lh;
      aoff0[i][kh] = row * 128 + ((c ^ sw) << 4);
      aoff1[i][kh] = row * 128 + (((4 + c) ^ sw) << 4);
    }
  }
  auto read_w = [&](int stage, u32x4 (&wf)[NJ][2][NP]) {
    const unsigned char* base = smem + stage * BUF;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int p = 0; p < NP; ++p) wf[j][kh][p] = *reinterpret_cast<const u32x4*>(base + p * WPLANE + woff[j][kh]);
  };
  auto read_a = [&](int stage, f32x4 (&ar)[MI][2][2]) {
    const unsigned char* base = smem + stage * BUF;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        ar[i][kh][0] = *reinterpret_cast<const f32x4*>(base + aoff0[i][kh]);
        ar[i][kh][1] = *reinterpret_cast<const f32x4*>(base + aoff1[i][kh]);
      }
  };
  auto mfma32 = [&](const u32x4& a, const u32x4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  };
  // the products of row block i against all NJ column blocks for one 16-k half: smallest terms first, NJ independent
  // accumulators between dependent MFMAs (32x32x16: 16 passes, the next MFMA on the same block is 4 instructions away)
  auto mma = [&](int i, int kh, const u32x4 (&wf)[NJ][2][NP], const u32x4 (&af)[NP]) {
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
    if constexpr (NP == 2) {
      constexpr int PW[3] = {1, 0, 0}, PA[3] = {0, 1, 0};                     // lo*hi hi*lo hi*hi
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = mfma32(wf[j][kh][PW[t]], af[PA[t]], acc[i][j]);
    } else {
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = mfma32(wf[j][kh][0], af[0], acc[i][j]);
    }
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
  };
  auto split = [&](const f32x4 (&a)[2], u32x4 (&af)[NP], float sc) {
    if constexpr (NP == 2) split8_h2(a[0], a[1], sc, af[0], af[1]);
    else cvt8_h1(a[0], a[1], sc, af[0]);
  };

  const int nk = d.K / BK;
#pragma unroll
  for (int s = 0; s < S; ++s)
    if (s < nk) issue(s);
  auto wait_tiles = [&](auto tiles) {
    constexpr int T = decltype(tiles)::value;
    if (wfull) wait_vm_lgkm0<T * LPT>();
    else wait_vm_lgkm0<T * (LPT - NP)>();
  };
  if (nk >= S) wait_tiles(std::integral_constant<int, S - 1>{});
  else wait_vm_lgkm0<0>();
  __builtin_amdgcn_s_barrier();
  u32x4 wfa[NJ][2][NP], wfb[NJ][2][NP];
  f32x4 ar[MI][2][2];
  read_w(0, wfa);
  read_a(0, ar);
  int stage = 0;

  auto step = [&](int kt, const u32x4 (&wc)[NJ][2][NP], u32x4 (&wn_)[NJ][2][NP]) {
    const bool more = kt + 1 < nk;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      u32x4 af[NP];
      split(ar[i][0], af, a_scale[i]);
      mma(i, 0, wc, af);
    }
    u32x4 af2[MI][NP];
#pragma unroll
    for (int i = 0; i < MI; ++i) split(ar[i][1], af2[i], a_scale[i]);
    const int nstage = stage + 1 == S ? 0 : stage + 1;
    __builtin_amdgcn_sched_barrier(0);
    if (more) {
      if (kt + S <= nk) wait_tiles(std::integral_constant<int, S - 2>{});
      else wait_vm_lgkm0<0>();
      __builtin_amdgcn_s_barrier();
      if (kt + S < nk) issue(stage);
      read_w(nstage, wn_);
      read_a(nstage, ar);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < MI; ++i) mma(i, 1, wc, af2[i]);
    stage = nstage;
  };
  for (int kt = 0; kt < nk; kt += 2) {
    step(kt, wfa, wfb);
    if (kt + 1 < nk) step(kt + 1, wfb, wfa);
  }
  __syncthreads();
  // accumulators as 8-column blocks: block 4 j + g of lane (l31, lh) = columns 32 j + 8 g + 4 lh .. + 3 of row 32 i + l31
  f32x4 accv[MI][4 * NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) accv[i][4 * j + g][e] = acc[i][j][4 * g + e];
  gemm_epilogue<BM, BN, TM, TN, MI, 4 * NJ, true, 32, 8>(d, accv, tm, tn, wm, wn, l31, lh, cz, bz, z0, row_inv, d.col_scale,
                                                       reinterpret_cast<float*>(smem) + wave * 3 * TN);
}

template <int BM, int BN, int WGM, int WGN, int S, int NP, int OCC = 1, int PRIO = 0>
int launch_split32_cfg(const dzn_gemm_desc& d, hipStream_t s) {
  const int tilesM = (d.M + BM - 1) / BM, tilesN = (d.N + BN - 1) / BN;
  const size_t lds = (size_t)S * (BM * 128 + NP * BN * 64);
  auto kern = gemm_split32_kernel<BM, BN, WGM, WGN, S, NP, OCC, PRIO>;
  static unsigned long long attr_mask = 0;
  if (first_use_on_device(attr_mask))
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  dim3 grid(tilesM * tilesN, d.nz > 0 ? d.nz : 1, 1);
  int pid = -1;
  if (prof_enabled()) {
    char cls[64];
    static const bool by_shape = getenv("DZN_PROFILE_SHAPES") != nullptr;
    if (by_shape)
      snprintf(cls, sizeof(cls), "gemm_%s_%dx%d M%d N%d K%d z%d", NP == 2 ? "f32h" : "f16", BM, BN, d.M, d.N, d.K, d.nz);
    else
      snprintf(cls, sizeof(cls), "gemm_%s_%dx%d", NP == 2 ? "f32h" : "f16", BM, BN);
    const double fl = d.alg_flops > 0 ? d.alg_flops * d.nz : 2.0 * d.M * d.N * d.K * d.nz;
    pid = prof_begin(s, cls, fl, gemm_alg_bytes(d, NP * 2));
  }
  hipLaunchKernelGGL(kern, grid, dim3(WGM * WGN * 64), lds, s, d);
  prof_end(pid, s);
  if (hipGetLastError() != hipSuccess) return DZN_E_HIP;
  if (d.stat_partial && d.stat_final)
    return launch_stats_finalize(d.stat_partial, d.M, tilesN * WGN, d.stat_C, d.stat_eps, d.stat_final, s);
  return DZN_OK;
}

#endif  // DZN_TUNING (32x32x16 probe)

// Column groups of a launch (gemm_split_kernel, "COLUMN GROUPS").  Fabric reads of a launch as a function of G, from the
// measured miss fractions of the weight-plane set per row block (profiles/r4_gemm_refetch_probe.txt; M = 149226, K = 1024,
// one XCD = 4 MB of L2 shared with the streaming A / C / residual traffic):
//     plane set seen by an XCD   0.5 MB   1 MB    2 MB    4 MB    8 MB
//     fraction re-fetched        0.02     0.055   0.12    0.27    0.69        (log-linear between, 1.0 from 12 MB on)
//     reads(G) = G x A  +  row_blocks x W x miss(W / G)
// G in {1, 2, 4, 8}, G <= column tiles, the smallest reads(G) wins and G = 1 unless the saving is worth 5 % of the launch's reads.
// STATE (end of r4): built, correct (kernel tests + the turn-taking goldens pass under forced G = 3 / 4), and NEUTRAL in time — the
// step is 1040.8 ms with the model's choice against 1039.2 ms with G = 1 — although the fabric reads DO drop as modelled (N = 2048:
// 8.59 -> 4.82 GB per launch, model 4.9; profiles/r4_gemm_refetch_probe.txt): the re-fetched planes come from the 256 MB
// Infinity Cache, which costs neither time nor measurable power.  So the default stays G = 1, the r1-r3 order;
// DZN_GEMM_NGROUPS (read once) = "auto" takes the model's choice, a number forces it.
int choose_column_groups(const dzn_gemm_desc& d, int tilesM, int tilesN, int BM, int NP) {
  static const char* env = getenv("DZN_GEMM_NGROUPS");
  static const bool automatic = env && !strcmp(env, "auto");
  static const int forced = env && !automatic ? atoi(env) : 0;
  if (tilesN < 2 || d.w_z0 || d.w_z1 || (!automatic && forced <= 1)) return 1;
  if (forced > 0) return forced < tilesN ? (forced > 8 ? 8 : forced) : tilesN > 8 ? 8 : tilesN;
  if ((d.nz > 1) || (int64_t)tilesM * tilesN < 1024) return 1;   // z-batched / small launches: the planes are small or the chip is not full
  const double W = (double)d.N * d.K * 2.0 * NP;
  const double a_cols = d.a_rowoff ? (double)d.kc : (d.lda > 0 && d.lda < d.K ? (double)d.lda : (double)d.K);
  const double A = (double)d.M * a_cols * 4.0;
  auto miss = [](double bytes) {
    static const double mb[] = {0.5, 1.0, 2.0, 4.0, 8.0, 12.0}, f[] = {0.02, 0.055, 0.12, 0.27, 0.69, 1.0};
    const double x = bytes / (1024.0 * 1024.0);
    if (x <= mb[0]) return f[0] * x / mb[0];
    for (int i = 1; i < 6; ++i)
      if (x <= mb[i]) return f[i - 1] + (f[i] - f[i - 1]) * (log2(x) - log2(mb[i - 1])) / (log2(mb[i]) - log2(mb[i - 1]));
    return 1.0;
  };
  int best = 1;
  const double r1 = A + tilesM * W * miss(W);
  double rbest = r1;
  for (int G = 2; G <= 8 && G <= tilesN; G *= 2) {
    const double r = G * A + tilesM * W * miss(W / G);
    if (r < rbest) { rbest = r; best = G; }
  }
  (void)BM;
  return rbest < 0.95 * r1 ? best : 1;
}

template <int BM, int BN, int WGM, int WGN, int S, int NP, int OCC = 1, int ABL = 0, bool RPF = false>
int launch_split_cfg(const dzn_gemm_desc& d, hipStream_t s) {
  const int tilesM = (d.M + BM - 1) / BM, tilesN = (d.N + BN - 1) / BN;
  const size_t lds = (size_t)S * (BM * 128 + NP * BN * 64);
  auto kern = gemm_split_kernel<BM, BN, WGM, WGN, S, NP, OCC, ABL, RPF>;
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
      snprintf(cls, sizeof(cls), "gemm_%s_%dx%d M%d N%d K%d z%d", NP == 3 ? "f32s" : NP == 2 ? "f32h" : "f16", BM, BN, d.M,
               d.N, d.K, d.nz);
    else
      snprintf(cls, sizeof(cls), "gemm_%s_%dx%d%s", NP == 3 ? "f32s" : NP == 2 ? "f32h" : "f16", BM, BN, RPF ? "_rpf" : "");
    const double fl = d.alg_flops > 0 ? d.alg_flops * d.nz : 2.0 * d.M * d.N * d.K * d.nz;
    pid = prof_begin(s, cls, fl, gemm_alg_bytes(d, NP * 2));
  }
  hipLaunchKernelGGL(kern, grid, dim3(WGM * WGN * 64), lds, s, d, choose_column_groups(d, tilesM, tilesN, BM, NP));
  prof_end(pid, s);
  if (hipGetLastError() != hipSuccess) return DZN_E_HIP;
  if (d.stat_partial && d.stat_final)
    return launch_stats_finalize(d.stat_partial, d.M, tilesN * WGN, d.stat_C, d.stat_eps, d.stat_final, s);
  return DZN_OK;
}

// W [rows][K] fp32 (row stride ldw)  ->  W3 [rows][K/32][3][32] bf16, k permuted inside each block
__global__ __launch_bounds__(256) void split_weights_kernel(const float* __restrict__ W, int64_t rows, int K,
                                                            int64_t ldw, u16* __restrict__ W3) {
  const int64_t n = rows * K;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / K;
    const int k = (int)(i - r * K);
    const float x = W[r * ldw + k];
    const __bf16 h = (__bf16)x;
    const float r1 = x - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    const __bf16 l = (__bf16)r2;
    const int kk = k & 31;
    const int pos = 8 * ((kk & 15) >> 2) + (kk & 3) + 4 * (kk >> 4);
    u16* o = W3 + r * 3 * K + (int64_t)(k >> 5) * 96 + pos;
    o[0] = *reinterpret_cast<const u16*>(&h);
    o[32] = *reinterpret_cast<const u16*>(&m);
    o[64] = *reinterpret_cast<const u16*>(&l);
  }
}

// DZN_GEMM_CFG (read once) or dzn_op_set_gemm_cfg() (tuning scripts: several shapes in one process)
char g_force_buf[32] = {0};
bool g_force_init = false;
const char* g_force_cfg() {
  if (!g_force_init) {
    const char* e = getenv("DZN_GEMM_CFG");
    if (e) snprintf(g_force_buf, sizeof(g_force_buf), "%s", e);
    g_force_init = true;
  }
  return g_force_buf[0] ? g_force_buf : nullptr;
}

template <int NP>
int launch_gemm_split_np(const dzn_gemm_desc& d, hipStream_t s) {
  const char* force = g_force_cfg();                    // tuning knob: force one tile shape
  if (force) {
    // production tiles by name
    if (!strcmp(force, "128x64")) return launch_split_cfg<128, 64, 4, 1, 2, NP, NP <= 2 ? 3 : 2>(d, s);
    if (!strcmp(force, "128x80")) return launch_split_cfg<128, 80, 4, 1, 2, NP, 2>(d, s);
    if (!strcmp(force, "128x32")) return launch_split_cfg<128, 32, 4, 1, 2, NP>(d, s);
    // (r5 probes, measured negative and removed: 64x64 tiles of 2 wavefronts, 128x32 at 4 workgroups per CU, a first-round stagger of
    //  the narrow tile — profiles/r5_short_k_probes.txt, code at commit 2a14565)
    if constexpr (NP == 3) {
      if (!strcmp(force, "128x128")) return launch_split_cfg<128, 128, 2, 2, 2, NP, 2>(d, s);
    } else {
      if (!strcmp(force, "128x128w4")) return launch_split_cfg<128, 128, 4, 1, 2, NP, 2>(d, s);
    }
    if constexpr (NP == 1) {
      if (!strcmp(force, "256x128w8s3")) return launch_split_cfg<256, 128, 8, 1, 3, NP, 2>(d, s);
    }
#ifdef DZN_TUNING
    if constexpr (NP <= 2) {    // (r4) 32x32x16 MFMA forms
      if (!strcmp(force, "m32_128x128")) return launch_split32_cfg<128, 128, 4, 1, 2, NP, 2>(d, s);
      if (!strcmp(force, "m32_128x64")) return launch_split32_cfg<128, 64, 4, 1, 2, NP, 3>(d, s);
    }   // probe / ablation instantiations (profiles/r2_gemm_cfg_probe.txt, r2_gemm_ablation.txt): build with
                    // DZN_TUNING=1 (diarizen_amd/build.py); they triple the compile time of this file
    if (!strcmp(force, "256x128")) return launch_split_cfg<256, 128, 4, 2, 2, NP>(d, s);
    if constexpr (NP != 3) {
      if (!strcmp(force, "128x128")) return launch_split_cfg<128, 128, 2, 2, 2, NP>(d, s);
    }
    if constexpr (NP <= 2) {
      if (!strcmp(force, "m32_128x128p")) return launch_split32_cfg<128, 128, 4, 1, 2, NP, 2, 1>(d, s);   // + s_setprio around the MFMAs
      if (!strcmp(force, "m32_128x128w22")) return launch_split32_cfg<128, 128, 2, 2, 2, NP, 2>(d, s);    // 2 x 2 wavefronts of 64 x 64
      if (!strcmp(force, "m32_256x128")) return launch_split32_cfg<256, 128, 4, 2, 2, NP, 1>(d, s);       // 8 wavefronts of 64 x 64
      if (!strcmp(force, "m32_256x128w8")) return launch_split32_cfg<256, 128, 8, 1, 2, NP, 1>(d, s);     // 8 wavefronts of 32 x 128
      if (!strcmp(force, "m32_128x64p")) return launch_split32_cfg<128, 64, 4, 1, 2, NP, 3, 1>(d, s);
      // deeper LDS-DMA pipelines (more K tiles in flight per CU): probes of the load-latency bound.  (128x128 with
      // S = 3 / 4 were probed too — one wavefront per SIMD in the two-term kernel, 30 % slower)
      if (!strcmp(force, "256x128s3")) return launch_split_cfg<256, 128, 4, 2, 3, NP>(d, s);
      if (!strcmp(force, "256x128w8s3")) return launch_split_cfg<256, 128, 8, 1, 3, NP, 2>(d, s);
      if (!strcmp(force, "128x64s3")) return launch_split_cfg<128, 64, 4, 1, 3, NP>(d, s);
      if (!strcmp(force, "256x64s3")) return launch_split_cfg<256, 64, 8, 1, 3, NP>(d, s);
      if (!strcmp(force, "abl1")) return launch_split_cfg<128, 128, 4, 1, 2, NP, 2, 1>(d, s);
      if (!strcmp(force, "abl2")) return launch_split_cfg<128, 128, 4, 1, 2, NP, 2, 2>(d, s);
      if (!strcmp(force, "abl3")) return launch_split_cfg<128, 128, 4, 1, 2, NP, 2, 3>(d, s);
      if (!strcmp(force, "abl4")) return launch_split_cfg<128, 128, 4, 1, 2, NP, 2, 4>(d, s);
      if (!strcmp(force, "abl5")) return launch_split_cfg<128, 128, 4, 1, 2, NP, 2, 5>(d, s);
    }
#endif
  }
  // (r3's 8-wavefront ping-pong forms — gemm_pp.hip, 256 x 192 tiles — measured +0.5 % on the step and were removed in r4;
  // the A/B record is profiles/r3_gemm_pq_probe.txt, the source is in the history at 7bb9ad7)
  // (r4, all bit-identical, all measured neutral or negative, sources in the history: a separate three-stage A ring — aea83fd,
  // profiles/r4_gemm_a3_probe.txt; persistent workgroups with the next tile's prologue fill under the epilogue — 6ff2eda,
  // profiles/r4_gemm_persist_probe.txt; 256 x 256 tiles at one wavefront per SIMD — e863011, profiles/r4_gemm_wide_probe.txt,
  // whose PMC pass shows clock x matrix-pipe-busy constant across both forms: the K loop sits on a power-limited MFMA rate)
#ifdef DZN_TUNING
  // (r4) DZN_GEMM_M32 (read once): bit 0 = the 128 x 128 class, bit 1 = the 128 x 64 class run on 32x32x16 MFMA blocks
  static const int m32 = getenv("DZN_GEMM_M32") ? atoi(getenv("DZN_GEMM_M32")) : 0;
#else
  constexpr int m32 = 0;
#endif
  // launch bounds pin the occupancy the tile was tuned at (r3: the pipelined epilogue gives the register allocator
  // room to trade occupancy for more loads in flight; 128x64 tiles want 3 workgroups per CU, 128x128 two)
  constexpr int OCC64 = NP <= 2 ? 3 : 2;
  if (d.N <= 32) return launch_split_cfg<128, 32, 4, 1, 2, NP>(d, s);
  // 128x64 tiles run 4 wavefronts as 4x1 (32 rows x 64 columns each): the in-register operand
  // split is per A row, so wide-and-short wavefront tiles halve the VALU work per MFMA
  // (r3 probe, profiles/r3_tile_choice_short_k.txt: in isolation the 128x128 tile is 4-7 % faster on the K = 256 .. 512
  // shapes now that the epilogue is pipelined — 424 vs 444 us at 149226 x 1024 x 256; in the pipeline the step time did
  // not move (1139 vs 1124-1142 ms), so the short-K launches stay on the narrow tile and the 128x128 symbol stays a
  // homogeneous K >= 768 class for the roofline line)
#ifdef DZN_TUNING
  if constexpr (NP <= 2) {
    if ((m32 & 2) && d.N > 32 && (d.N <= 64 || d.K <= 512)) return launch_split32_cfg<128, 64, 4, 1, 2, NP, 3>(d, s);
  }
#endif
  if constexpr (NP == 2) {
    // (r5) short-K launches with a residual (out_proj / FFN-output) with the residual requested at kernel START: a MEASURED
    // NEGATIVE, kept as a switch (DZN_GEMM_RPF=1, read once).  The 32 residual registers cost the third workgroup per CU: class
    // 168 -> 134 TFLOP/s, device step 1743 -> 1705 audio-s/s on one box (profiles/r5_rpf_probe.txt); at three workgroups per CU
    // the form spills (probed as DZN_GEMM_RPF=2, instantiation removed).  The epilogue's load latency is not what these launches wait for.
    static const int rpf = getenv("DZN_GEMM_RPF") ? atoi(getenv("DZN_GEMM_RPF")) : 0;
    if (rpf && d.R && d.N > 64 && d.K <= 512) {
      return launch_split_cfg<128, 64, 4, 1, 2, NP, 2, 0, true>(d, s);   // (at 3 workgroups per CU the form spills 24 registers: probed, removed)
    }
  }
  if (d.N <= 64 || d.K <= 512) return launch_split_cfg<128, 64, 4, 1, 2, NP, OCC64>(d, s);
  // 128-wide column tiles unless 64-wide ones save more than ~1/8 of the (padded) columns; widths that
  // are multiples of 80 but not of 64 (conv1 of the extractor: 153 -> 160) get exact 80-wide tiles
  const int cols128 = (d.N + 127) / 128 * 128, cols64 = (d.N + 63) / 64 * 64;
  // small launches (BASELINE configs[1]: 32 windows of 5 s = 7968 rows): 128 x 128 tiles would leave most of the 512
  // workgroup slots (256 CUs x 2) empty — halve the tile so that twice as many workgroups exist
  if ((int64_t)((d.M + 127) / 128) * (cols128 / 128) * (d.nz > 0 ? d.nz : 1) < 448)
    return launch_split_cfg<128, 64, 4, 1, 2, NP, OCC64>(d, s);
  if (d.N % 80 == 0 && d.N < cols64 && d.N * 9 < cols128 * 8) return launch_split_cfg<128, 80, 4, 1, 2, NP, 2>(d, s);
  if (cols64 * 9 < cols128 * 8) return launch_split_cfg<128, 64, 4, 1, 2, NP, OCC64>(d, s);
  // NP = 2: 4 x 1 wavefronts (32 x 128 each): every A row is split by ONE wavefront instead of two; measured +3 %
  // over 2 x 2 on the pipeline's K = 1024 shapes (scripts/bench_gemm_h2.py).  NP = 3 keeps 2 x 2 (register budget).
  // NP = 1 (DZN_PREC_F16) is bound by the global -> LDS fill, not by MFMA / VALU (ablation: profiles/r2_gemm_ablation.txt):
  // 256 x 128 tiles halve the W bytes per flop and a third stage keeps two K tiles in flight: +7..13 % (r2_gemm_cfg_probe.txt)
  if constexpr (NP == 1) return launch_split_cfg<256, 128, 8, 1, 3, NP, 2>(d, s);
  if constexpr (NP == 2) {
#ifdef DZN_TUNING
    if (m32 & 1) return launch_split32_cfg<128, 128, 4, 1, 2, NP, 2>(d, s);
#endif
    return launch_split_cfg<128, 128, 4, 1, 2, NP, 2>(d, s);   // held to 2 wavefronts per SIMD
  }
  return launch_split_cfg<128, 128, 2, 2, 2, NP, 2>(d, s);
}

// W [rows][K] fp32 -> W2h [rows][K/32][2][32] fp16 (k permuted as above) of w * 2^e_row, e_row chosen so that
// the row's max |w| lands in [2^14, 2^15); col_scale[row] = 2^-e_row (exact).  One wavefront per row.
__global__ __launch_bounds__(256) void split_weights_h2_kernel(const float* __restrict__ W, int64_t rows, int K,
                                                               int64_t ldw, u16* __restrict__ W2, float* __restrict__ col_scale) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float m = 0.f;
  for (int k = lane; k < K; k += 64) m = fmaxf(m, fabsf(W[r * ldw + k]));
  m = wave_max(m);
  float sc, inv;
  h2_scale(m, sc, inv);
  if (lane == 0) col_scale[r] = inv;
  for (int k = lane; k < K; k += 64) {
    const float x = W[r * ldw + k] * sc;
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)(x - (float)h);
    const int kk = k & 31;
    const int pos = 8 * ((kk & 15) >> 2) + (kk & 3) + 4 * (kk >> 4);
    u16* o = W2 + r * 2 * K + (int64_t)(k >> 5) * 64 + pos;
    o[0] = __builtin_bit_cast(u16, h);
    o[32] = __builtin_bit_cast(u16, l);
  }
}

}  // namespace

int launch_gemm_split(const dzn_gemm_desc& d, hipStream_t s) {
  if ((d.K & 31) || (d.kc & 31) || d.ldw != d.K) return DZN_E_INVALID;
  // fp16 two-term path: needs the fp16 planes + their row scales, the producer-tracked |max| of A, and weights
  // that do not move with z (col_scale is indexed by the output column alone)
  static const bool no_h2 = getenv("DZN_NO_H2") != nullptr;
  // DZN_PREC_F16 with the MX planes: fp16 hi*hi + fp8 cross terms (gemm_mx.hip); without them the single-term fp16 kernel
  // (ln_centered — the row mean subtracted before the fp16 rounding — is implemented by the single-term kernel alone: any
  // other dispatch with it set would silently drop the mean term of the folded LayerNorm, so it is refused)
  if (d.precision == DZN_PREC_F16 && d.Wmx && d.col_scale_mx && d.a_amax && !d.w_z0 && !d.w_z1 && !no_h2)
    return d.ln_centered ? DZN_E_INVALID : launch_gemm_mx(d, s);
  if (prec_is_h2(d.precision) && d.W2h && d.col_scale && d.a_amax && !d.w_z0 && !d.w_z1 && !no_h2) {
    if (d.precision == DZN_PREC_F16) return launch_gemm_split_np<1>(d, s);
    return d.ln_centered ? DZN_E_INVALID : launch_gemm_split_np<2>(d, s);
  }
  if (!d.W3 || d.ln_centered) return DZN_E_INVALID;
  return launch_gemm_split_np<3>(d, s);
}

extern "C" int dzn_op_set_gemm_cfg(const char* cfg) {
  g_force_init = true;
  snprintf(g_force_buf, sizeof(g_force_buf), "%s", cfg && strcmp(cfg, "auto") ? cfg : "");
  return DZN_OK;
}

int launch_split_weights_h2(const float* W, int64_t rows, int K, int64_t ldw, void* W2, float* col_scale, hipStream_t s) {
  if (rows <= 0) return DZN_OK;
  if (K <= 0 || (K & 31)) return DZN_E_INVALID;
  hipLaunchKernelGGL(split_weights_h2_kernel, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, s, W, rows, K, ldw,
                     static_cast<u16*>(W2), col_scale);
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}

extern "C" int dzn_op_split_weights_h2(const float* W, int64_t rows, int32_t K, int64_t ldw, void* W2, float* col_scale,
                                       void* stream) {
  if (!W || !W2 || !col_scale) return DZN_E_INVALID;
  return launch_split_weights_h2(W, rows, K, ldw, W2, col_scale, reinterpret_cast<hipStream_t>(stream));
}

int launch_split_weights(const float* W, int64_t rows, int K, int64_t ldw, void* W3, hipStream_t s) {
  if (rows <= 0) return DZN_OK;
  if (K <= 0 || (K & 31)) return DZN_E_INVALID;
  int64_t g = cdiv64(rows * K, 256);
  g = g > 8192 ? 8192 : g;
  hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)g), dim3(256), 0, s, W, rows, K, ldw,
                     static_cast<u16*>(W3));
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}

extern "C" int dzn_op_split_weights(const float* W, int64_t rows, int32_t K, int64_t ldw, void* W3, void* stream) {
  if (!W || !W3) return DZN_E_INVALID;
  return launch_split_weights(W, rows, K, ldw, W3, reinterpret_cast<hipStream_t>(stream));
}
