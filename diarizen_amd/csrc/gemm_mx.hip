// gemm_mx.hip — the REDUCED-precision contraction of the DZN_PREC_F16 engine mode (BASELINE configs[4], SURVEY 8d reduced bar):
//
//     C = A W^T  ~  hi(A) hi(W)^T  [fp16, v_mfma_f32_32x32x16_f16]
//                 + hi8(A) lo8(W)^T + lo8(A) hi8(W)^T  [fp8 e4m3, v_mfma_scale_f32_32x32x64_f8f6f4, block scales 2^0 / 2^-11]
//
// x s = hi + lo with hi = fp16(x s), lo = x s - hi (exact in fp32, |lo| <= 2^-11 |x s|).  The single-term mode (hi hi only) fails
// SURVEY 8d's bar on non-degenerate weights (max |dlogp| 0.15-0.18 vs 5e-2: r4, profiles/r4_f16_sensitivity.json) because BOTH
// operands are rounded at 2^-11; the two cross terms bring the result back to 2^-11 x 2^-4 (fp8 keeps 4 significant bits OF THE
// CROSS TERMS): emulated on the oracle 6.3e-3, 8 x inside the bar (profiles/r4_reduced_mode_emulation.txt).  Matrix-pipe work per
// 32 x 32 x 64 block: 4 fp16 MFMAs of 8 passes + 2 scaled fp8 MFMAs of 16 passes = 64 passes against f32h's 96 — two thirds.
//
// Why the block-scaled instruction: gfx950's plain fp8 MFMA (16x16x32 / 32x32x16) runs at the fp16 rate; only the MX form
// (K = 64 per 32 x 32 block) has twice the rate.  Its per-32-k block scales (E8M0, one byte per lane) are used as two exact
// constants: 2^0 for a hi8 operand and 2^-11 for a lo8 operand, which is how the cross terms land in the SAME fp32 accumulator as
// hi hi without a multiply (lo is stored as fp8(lo 2^11), see below).
//
// Scaling (exact powers of two, undone exactly in the epilogue, as in gemm_split.hip but with the |max| in [2^7, 2^8) so that
// fp8's range — largest finite value 448 — holds the hi operand without a second multiply): weights per output row at pack time
// (split_weights_mx_kernel), activations per window from the producer's |max| tracker (dzn_gemm_desc.a_amax).  fp16 keeps its
// 11 bits for everything within 2^21 of the row / window maximum.
//
// Data movement is gemm_split.hip's, unchanged: A fp32 by LDS-DMA (XOR-swizzled 128-B rows), two 64-B weight planes per 32 k by
// LDS-DMA, two LDS stages, the barrier of K tile kt in the middle of its MFMAs.  Weight plane 0 = hi16 (fragment order of
// gemm_split.hip), plane 1 = per 16-B slot c: [fp8(w s) of the slot's 8 k | fp8((w s - hi16) 2^11) of the same 8 k] — one
// ds_read_b128 per (column block, k half) delivers both fp8 operands' bytes for that tile.  The fp8 operands of the MX
// instruction span TWO K tiles (64 k): they are assembled in registers — the A side by the in-register split, which converts
// each fp32 value once into (fp16 hi, fp8 hi, fp8 lo), the W side straight from the LDS reads — and the two MX products of a
// 32 x 32 block are issued at the end of every second tile.  Lane (l31 = lane & 31, lh = lane >> 5) holds, in both operands,
// byte 16 t + 8 kh + e = element e of chunk 2 kh + lh of tile t: the same k on both sides, which is all the instruction needs
// when every block scale of an operand is the same constant (scripts/ubench/mx_probe.hip pins the operand layout, the C layout,
// the scale semantics and the fp8 conversion on the device).
// The 32 x 32 block shape is what makes the register budget: an fp8 operand register covers 32 rows (16x16x128 would need twice
// the operand registers for the same wavefront tile: 128 for the 32 x 128 tile's weight side alone).
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "checked.h"
#include "common.h"
#include "gemm_epilogue.h"
#include "split.h"

DZN_CHECKED_TU(gemm_mx)

namespace {

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int wswz_mx(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }

template <int N>
__device__ __forceinline__ void wait_vm_lgkm0_mx() {
  static_assert(N >= 0 && N < 64, "vmcnt range");
  __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | (0x7 << 4));
}

// exact power-of-two scale that puts `amax` into [2^7, 2^8) (e4m3's largest finite value is 448), and its inverse
__device__ __forceinline__ void mx_scale(float amax, float& s, float& inv) {
  int e = (int)((__float_as_uint(amax) >> 23) & 0xff) - 127;   // floor(log2 amax) for normal amax
  if (!(amax > 0.f) || e > 100) e = 7;                          // empty / non-finite tracker: scale 1
  e = e < -100 ? -100 : e;
  s = __uint_as_float((unsigned)(7 - e + 127) << 23);
  inv = __uint_as_float((unsigned)(e - 7 + 127) << 23);
}

constexpr float MX_LO_UP = 2048.f;       // lo is stored as fp8(lo 2^11); the instruction's block scale 2^-11 undoes it
constexpr int MX_SCALE_ONE = 127;        // E8M0: 2^(byte - 127)
constexpr int MX_SCALE_LO = 127 - 11;

// 8 fp32 values of one k chunk (two float4), scaled by the exact power of two s  ->  the fp16 fragment, 8 bytes fp8(x s) and
// 8 bytes fp8((x s - fp16(x s)) 2^11).  Per pair: 2 v_mul, v_cvt_pk_f16_f32 (nearest even), 2 v_cvt_f32_f16, 2 v_sub (exact),
// 2 v_mul, 2 v_cvt_pk_fp8_f32.
// (r5 probe, removed: applying the remainder's 2^11 through v_cvt_scalef32_pk_fp8_f32's scale operand — the conversion DIVIDES by
// it — saves two of the eleven VALU operations per pair and measured +1 %, 314 -> 317 TFLOP/s at N = K = 1024:
// profiles/r5_mx_cvtscale_probe.txt.  The kernel is bound by its LDS-DMA fill, not by this chain.)
__device__ __forceinline__ void split8_mx(const f32x4& u, const f32x4& v, float s, u32x4& hi, int& h8a, int& h8b, int& l8a, int& l8b) {
  h8a = h8b = l8a = l8b = 0;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    f32x2 x;
    x[0] = (p < 2 ? u[2 * p] : v[2 * p - 4]) * s;
    x[1] = (p < 2 ? u[2 * p + 1] : v[2 * p - 3]) * s;
    const f16x2 h = __builtin_convertvector(x, f16x2);
    const f32x2 hf = __builtin_convertvector(h, f32x2);
    hi[p] = __builtin_bit_cast(unsigned, h);
    int& h8 = p < 2 ? h8a : h8b;
    int& l8 = p < 2 ? l8a : l8b;
    // (the word-select argument of the conversions must be a literal)
    if (p & 1) h8 = __builtin_amdgcn_cvt_pk_fp8_f32(x[0], x[1], h8, true);
    else h8 = __builtin_amdgcn_cvt_pk_fp8_f32(x[0], x[1], h8, false);
    const float r0 = (x[0] - hf[0]) * MX_LO_UP, r1 = (x[1] - hf[1]) * MX_LO_UP;
    if (p & 1) l8 = __builtin_amdgcn_cvt_pk_fp8_f32(r0, r1, l8, true);
    else l8 = __builtin_amdgcn_cvt_pk_fp8_f32(r0, r1, l8, false);
  }
}

// BM x BN tile per workgroup of WGM x WGN wavefronts, S LDS stages of one 32-k tile each; wavefront tiles of 32 x 32 blocks.
// sc_one / sc_lo arrive as kernel arguments so that the scale operands of the MX instruction are registers (a literal there is
// taken as an f32 constant by the compiler).
// RPF: residual prefetch at kernel start (gemm_epilogue.h), for the short-K launches whose time is their epilogue.
template <int BM, int BN, int WGM, int WGN, int S, int OCC = 1, bool RPF = false>
__global__ __launch_bounds__(WGM * WGN * 64, OCC) void gemm_mx_kernel(const dzn_gemm_desc d, const int sc_one, const int sc_lo) {
  constexpr int NW = WGM * WGN;
  constexpr int BK = 32;
  constexpr int TM = BM / WGM, TN = BN / WGN;
  constexpr int MI = TM / 32, NJ = TN / 32;
  constexpr int RB = NW * 1024;
  constexpr int ACH = BM * 128 / RB;
  constexpr int WROWS = NW * 16;
  constexpr int WR = (BN + WROWS - 1) / WROWS;
  constexpr int ABYTES = BM * 128, WPLANE = BN * 64, BUF = ABYTES + 2 * WPLANE;
  constexpr int LPT = ACH + 2 * WR;
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
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;   // XCD-contiguous tile ranges
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
  const u16* __restrict__ Wm = reinterpret_cast<const u16*>(d.Wmx);
  const int l31 = lane & 31, lh = lane >> 5;
  float a_scale[MI], row_inv[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    int m = tm * BM + wm * TM + i * 32 + l31;
    m = m < d.M ? m : d.M - 1;
    const int unit = d.amax_unit > 0 ? m / d.amax_unit : z0;
    DZN_CHECK(d.amax_count <= 0 || (unit >= 0 && unit < d.amax_count), 0x201, unit);
    mx_scale(d.a_amax[unit], a_scale[i], row_inv[i]);
  }
  const int64_t cz = z0 * d.c_z0 + z1 * d.c_z1;
  const int64_t bz = z0 * d.b_z0 + z1 * d.b_z1;

  f32x4 rpre[RPF ? MI : 1][RPF ? 4 * NJ : 1];
  bool use_rpre = false;
  if constexpr (RPF) {
    use_rpre = d.R != nullptr && gemm_epilogue_vec(d, cz, bz);
    if (use_rpre) gemm_prefetch_residual<BM, BN, TM, TN, MI, 4 * NJ, 32, 8>(d, rpre, tm, tn, wm, wn, l31, lh, cz);
  }
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
  const int wsw = (lane & 3) ^ wswz_mx(wr0);
  const u16* wptr[WR];
#pragma unroll
  for (int i = 0; i < WR; ++i) {
    int n = tn * BN + wr0 + WROWS * i;
    n = n < d.N ? n : d.N - 1;
    wptr[i] = Wm + (int64_t)n * 2 * d.ldw + wsw * 8;
  }
  int ik = 0, irem = 0;
  int64_t ikoff = 0;
  auto issue = [&](int stage) {
    unsigned char* sA = smem + stage * BUF + wave * 1024;
    unsigned char* sW = smem + stage * BUF + ABYTES + wave * 1024;
    DZN_CHECK(stage >= 0 && stage < S && ik < d.K, 0x205, stage);
#pragma unroll
    for (int i = 0; i < ACH; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(aptr[i] + ikoff),
                                       (__attribute__((address_space(3))) void*)(sA + i * RB), 16, 0, 0);
#pragma unroll
    for (int p = 0; p < 2; ++p)
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

  // per-lane LDS byte offsets: [block][k half]; chunk c = 2 kh + lh
  int woff[NJ][2], aoff0[MI][2], aoff1[MI][2];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int row = wn * TN + j * 32 + l31;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) woff[j][kh] = ABYTES + row * 64 + (((2 * kh + lh) ^ wswz_mx(row)) << 4);
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
  // fp8 operands of the current 64-k group: register 4 t + 2 kh + (0, 1) = the 8 bytes of (tile t, k half kh)
  i32x8 wh8[NJ], wl8[NJ], ah8[MI], al8[MI];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) wh8[j][e] = wl8[j][e] = 0;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) ah8[i][e] = al8[i][e] = 0;

#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) DZN_CHECK(woff[j][kh] >= ABYTES && woff[j][kh] + WPLANE + 16 <= BUF, 0x202, woff[j][kh]);
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) DZN_CHECK(aoff0[i][kh] + 16 <= ABYTES && aoff1[i][kh] + 16 <= ABYTES, 0x203, aoff1[i][kh]);
  DZN_CHECK(tm * BM < d.M && tn * BN < d.N, 0x204, t);
  auto read_w16 = [&](int stage, u32x4 (&wf)[NJ][2]) {
    const unsigned char* base = smem + stage * BUF;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) wf[j][kh] = *reinterpret_cast<const u32x4*>(base + woff[j][kh]);
  };
  auto read_w8 = [&](int stage, auto tt) {
    constexpr int T = decltype(tt)::value;
    const unsigned char* base = smem + stage * BUF + WPLANE;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(base + woff[j][kh]);
        wh8[j][4 * T + 2 * kh] = (int)v[0];
        wh8[j][4 * T + 2 * kh + 1] = (int)v[1];
        wl8[j][4 * T + 2 * kh] = (int)v[2];
        wl8[j][4 * T + 2 * kh + 1] = (int)v[3];
      }
  };
  // raw A fragments of one k half (kh) of a tile
  auto read_a = [&](int stage, int kh, f32x4 (&ar)[MI][2]) {
    const unsigned char* base = smem + stage * BUF;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      ar[i][0] = *reinterpret_cast<const f32x4*>(base + aoff0[i][kh]);
      ar[i][1] = *reinterpret_cast<const f32x4*>(base + aoff1[i][kh]);
    }
  };
  // hi hi of row block i against all NJ column blocks for one 16-k half (NJ independent accumulators between dependent MFMAs)
  auto mma16 = [&](int i, int kh, const u32x4 (&wf)[NJ][2], const u32x4& af) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wf[j][kh]), __builtin_bit_cast(f16x8, af), acc[i][j], 0, 0, 0);
  };
  // the two cross terms of the 64-k group: lo8(W) hi8(A) and hi8(W) lo8(A); first operand = W (the result's M index is the
  // output column, as in the fp16 products above)
  auto mx_all = [&]() {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wl8[j], ah8[i], acc[i][j], 0, 0, 0, sc_lo, 0, sc_one);
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wh8[j], al8[i], acc[i][j], 0, 0, 0, sc_one, 0, sc_lo);
    }
  };
  auto split = [&](const f32x4 (&a)[2], float sc, u32x4& af, i32x8& h8, i32x8& l8, auto tt, auto khc) {
    constexpr int R = 4 * decltype(tt)::value + 2 * decltype(khc)::value;
    int ha, hb, la, lb;
    split8_mx(a[0], a[1], sc, af, ha, hb, la, lb);
    // the fp8 bytes are first USED by the group's MX products, up to two tiles later: pin the conversions to this tile's step
    // (the compiler otherwise sinks them, and every tile's scaled fp32 values with them, in front of the MX products)
    asm volatile("" : "+v"(ha), "+v"(hb), "+v"(la), "+v"(lb));
    h8[R] = ha;
    h8[R + 1] = hb;
    l8[R] = la;
    l8[R + 1] = lb;
  };

  const int nk = d.K / BK;
#pragma unroll
  for (int s = 0; s < S; ++s)
    if (s < nk) issue(s);
  auto wait_tiles = [&](auto tiles) {
    constexpr int T = decltype(tiles)::value;
    if (wfull) wait_vm_lgkm0_mx<T * LPT>();
    else wait_vm_lgkm0_mx<T * (LPT - 2)>();
  };
  if (nk >= S) wait_tiles(std::integral_constant<int, S - 1>{});
  else wait_vm_lgkm0_mx<0>();
  __builtin_amdgcn_s_barrier();
  // Register budget (128 x 128 tile, 32 x 128 per wavefront): 64 accumulators + 64 + 16 fp8 operand registers leave no room to
  // hold the NEXT tile's hi16 fragments beside the current ones (gemm_split.hip prefetches both across the barrier).  Only the
  // first k half of the next tile's RAW A rows is prefetched (8 registers); the tile's weight fragments and its second A half
  // are read at the top of its own step, where the split of the prefetched half (VALU) covers their LDS latency.
  f32x4 ar0[MI][2];
  read_a(0, 0, ar0);
  int stage = 0;

  // one K tile (tile T of its 64-k group)
  auto step = [&](int kt, auto tt) {
    constexpr int T = decltype(tt)::value;
    const bool more = kt + 1 < nk;
    u32x4 wf[NJ][2];
    f32x4 ar1[MI][2];
    read_w16(stage, wf);
    read_w8(stage, tt);      // this tile's fp8 bytes: needed by the group's MX products only, but must leave LDS before the refill
    read_a(stage, 1, ar1);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      u32x4 af;
      split(ar0[i], a_scale[i], af, ah8[i], al8[i], tt, std::integral_constant<int, 0>{});
      mma16(i, 0, wf, af);
    }
    u32x4 af2[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) split(ar1[i], a_scale[i], af2[i], ah8[i], al8[i], tt, std::integral_constant<int, 1>{});
    const int nstage = stage + 1 == S ? 0 : stage + 1;
    __builtin_amdgcn_sched_barrier(0);
    if (more) {
      // tile kt+1 landed (younger tiles may stay in flight); all my reads of tile kt retired (lgkmcnt(0))
      if (kt + S <= nk) wait_tiles(std::integral_constant<int, S - 2>{});
      else wait_vm_lgkm0_mx<0>();
      __builtin_amdgcn_s_barrier();
      if (kt + S < nk) issue(stage);
      read_a(nstage, 0, ar0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < MI; ++i) mma16(i, 1, wf, af2[i]);
    if constexpr (T == 1) mx_all();
    stage = nstage;
  };
  int kt = 0;
  for (; kt + 1 < nk; kt += 2) {
    step(kt, std::integral_constant<int, 0>{});
    step(kt + 1, std::integral_constant<int, 1>{});
  }
  if (kt < nk) {   // K % 64 == 32: a last group of one tile, upper halves zero
    step(kt, std::integral_constant<int, 0>{});
#pragma unroll
    for (int e = 4; e < 8; ++e) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) wh8[j][e] = wl8[j][e] = 0;
#pragma unroll
      for (int i = 0; i < MI; ++i) ah8[i][e] = al8[i][e] = 0;
    }
    mx_all();
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
  if constexpr (RPF) {
    if (use_rpre) {
      gemm_epilogue<BM, BN, TM, TN, MI, 4 * NJ, true, 32, 8>(d, accv, tm, tn, wm, wn, l31, lh, cz, bz, z0, row_inv, d.col_scale_mx,
                                                           reinterpret_cast<float*>(smem) + wave * 3 * TN, rpre);
      return;
    }
  }
  gemm_epilogue<BM, BN, TM, TN, MI, 4 * NJ, true, 32, 8>(d, accv, tm, tn, wm, wn, l31, lh, cz, bz, z0, row_inv, d.col_scale_mx,
                                                       reinterpret_cast<float*>(smem) + wave * 3 * TN);
}

template <int BM, int BN, int WGM, int WGN, int S, int OCC, bool RPF = false>
int launch_mx_cfg(const dzn_gemm_desc& d, hipStream_t s) {
  const int tilesM = (d.M + BM - 1) / BM, tilesN = (d.N + BN - 1) / BN;
  const size_t lds = (size_t)S * (BM * 128 + 2 * BN * 64);
  auto kern = gemm_mx_kernel<BM, BN, WGM, WGN, S, OCC, RPF>;
  static unsigned long long attr_mask = 0;
  if (first_use_on_device(attr_mask))
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  dim3 grid(tilesM * tilesN, d.nz > 0 ? d.nz : 1, 1);
  int pid = -1;
  if (prof_enabled()) {
    char cls[64];
    static const bool by_shape = getenv("DZN_PROFILE_SHAPES") != nullptr;
    if (by_shape) snprintf(cls, sizeof(cls), "gemm_mx_%dx%d M%d N%d K%d z%d", BM, BN, d.M, d.N, d.K, d.nz);
    else snprintf(cls, sizeof(cls), "gemm_mx_%dx%d%s", BM, BN, RPF ? "_rpf" : "");
    const double fl = d.alg_flops > 0 ? d.alg_flops * d.nz : 2.0 * d.M * d.N * d.K * d.nz;
    pid = prof_begin(s, cls, fl, gemm_alg_bytes(d, 4));
  }
  hipLaunchKernelGGL(kern, grid, dim3(WGM * WGN * 64), lds, s, d, MX_SCALE_ONE, MX_SCALE_LO);
  prof_end(pid, s);
  if (hipGetLastError() != hipSuccess) return DZN_E_HIP;
  if (d.stat_partial && d.stat_final)
    return launch_stats_finalize(d.stat_partial, d.M, tilesN * WGN, d.stat_C, d.stat_eps, d.stat_final, s);
  return DZN_OK;
}

// W [rows][K] fp32 -> Wmx [rows][K/32][128 B]: bytes 0..63 = fp16 of w 2^e_row in gemm_split.hip's fragment order, bytes
// 64..127 = four 16-B slots c, each [fp8(w 2^e_row) x 8 | fp8((w 2^e_row - hi16) 2^11) x 8] in the slot's element order;
// e_row puts the row's max |w| into [2^7, 2^8); col_scale[row] = 2^-e_row (exact).  One wavefront per row.
__global__ __launch_bounds__(256) void split_weights_mx_kernel(const float* __restrict__ W, int64_t rows, int K, int64_t ldw,
                                                               unsigned char* __restrict__ Wmx, float* __restrict__ col_scale) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float m = 0.f;
  for (int k = lane; k < K; k += 64) m = fmaxf(m, fabsf(W[r * ldw + k]));
  m = wave_max(m);
  float sc, inv;
  mx_scale(m, sc, inv);
  if (lane == 0) col_scale[r] = inv;
  for (int k = lane; k < K; k += 64) {
    const float x = W[r * ldw + k] * sc;
    const _Float16 h = (_Float16)x;
    const float lo = (x - (float)h) * MX_LO_UP;
    const int kk = k & 31;
    const int c = (kk & 15) >> 2, e = (kk & 3) + 4 * (kk >> 4);
    unsigned char* blk = Wmx + (r * (int64_t)(K >> 5) + (k >> 5)) * 128;
    reinterpret_cast<u16*>(blk)[8 * c + e] = __builtin_bit_cast(u16, h);
    const int b8 = __builtin_amdgcn_cvt_pk_fp8_f32(x, lo, 0, false);   // byte 0 = fp8(x), byte 1 = fp8(lo 2^11)
    blk[64 + 16 * c + e] = (unsigned char)(b8 & 0xff);
    blk[64 + 16 * c + 8 + e] = (unsigned char)((b8 >> 8) & 0xff);
  }
}

// DZN_GEMM_MX_CFG (read once) or dzn_op_set_gemm_mx_cfg() (tests / tuning scripts): force one tile shape
char g_mx_force_buf[32] = {0};
bool g_mx_force_init = false;
const char* g_mx_force() {
  if (!g_mx_force_init) {
    const char* e = getenv("DZN_GEMM_MX_CFG");
    if (e) snprintf(g_mx_force_buf, sizeof(g_mx_force_buf), "%s", e);
    g_mx_force_init = true;
  }
  return g_mx_force_buf[0] ? g_mx_force_buf : nullptr;
}

}  // namespace

extern "C" int dzn_op_set_gemm_mx_cfg(const char* cfg) {
  g_mx_force_init = true;
  snprintf(g_mx_force_buf, sizeof(g_mx_force_buf), "%s", cfg && strcmp(cfg, "auto") ? cfg : "");
  return DZN_OK;
}

// the reduced-precision contraction: caller (launch_gemm_split) has checked K % 32 == 0, kc % 32 == 0, ldw == K
int launch_gemm_mx(const dzn_gemm_desc& d, hipStream_t s) {
  if (!d.Wmx || !d.col_scale_mx || !d.a_amax || d.w_z0 || d.w_z1) return DZN_E_INVALID;
  const int cols128 = (d.N + 127) / 128 * 128;
  // 128 x 128 tiles unless the launch is narrow or too small to fill the chip with them.  gemm_split.hip's other rules — narrow
  // tiles for K <= 512 and for widths that 64-wide tiles pad less — do NOT carry over: measured at M = 223 839 (first GPU run of
  // the round, profiles/r5_gemm_mx_first_bench.txt) the wide tile wins on all of them (N 1024 K 256: 186 vs 166 TFLOP/s, K 480:
  // 243 vs 214, N 320 K 1024: 256 vs 244): a 32 x 64 wavefront tile has half the MFMAs per converted A value.
  const bool narrow = d.N <= 64 || (int64_t)((d.M + 127) / 128) * (cols128 / 128) * (d.nz > 0 ? d.nz : 1) < 448;
  const char* force = g_mx_force();
  if (force && !strcmp(force, "128x64")) return launch_mx_cfg<128, 64, 4, 1, 2, 3>(d, s);
  if (force && !strcmp(force, "128x128")) return launch_mx_cfg<128, 128, 4, 1, 2, 2>(d, s);
  if (force && !strcmp(force, "128x64rpf")) return launch_mx_cfg<128, 64, 4, 1, 2, 2, true>(d, s);
  if (force && !strcmp(force, "256x128")) return launch_mx_cfg<256, 128, 8, 1, 2, 2>(d, s);     // probe: 8 wavefronts, one workgroup per CU
  if (narrow) return launch_mx_cfg<128, 64, 4, 1, 2, 3>(d, s);
  return launch_mx_cfg<128, 128, 4, 1, 2, 2>(d, s);
}

int launch_split_weights_mx(const float* W, int64_t rows, int K, int64_t ldw, void* Wmx, float* col_scale, hipStream_t s) {
  if (rows <= 0) return DZN_OK;
  if (K <= 0 || (K & 31)) return DZN_E_INVALID;
  hipLaunchKernelGGL(split_weights_mx_kernel, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, s, W, rows, K, ldw,
                     static_cast<unsigned char*>(Wmx), col_scale);
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}

extern "C" int dzn_op_split_weights_mx(const float* W, int64_t rows, int32_t K, int64_t ldw, void* Wmx, float* col_scale,
                                       void* stream) {
  if (!W || !Wmx || !col_scale) return DZN_E_INVALID;
  return launch_split_weights_mx(W, rows, K, ldw, Wmx, col_scale, reinterpret_cast<hipStream_t>(stream));
}
