// gemm_persist.hip — (r4) the f32h contraction of gemm_split.hip as a PERSISTENT kernel: one workgroup per workgroup slot
// of the chip (2 per CU), each walking a list of output tiles, with the first two K tiles of the NEXT output tile issued
// (LDS-DMA) before the epilogue of the current one.
//
// Why.  Timing gemm_split_kernel<128, 128, 4, 1, 2, 2> over K at M = 149 226, N = 1024 (profiles/r4_gemm_a3_probe.txt) gives
// t = 195 us + 52 us per 64 k: a fifth of a K = 1024 launch does not depend on K.  Per output tile that is the start of a
// fresh workgroup (kernel arguments, tracker and row-offset loads, address arithmetic), the prologue fill (nothing to
// multiply until K tile 0 has landed), the epilogue (no loads in flight for the matrix pipe behind it) and the tail of the
// grid.  The other workgroup of the CU covers part of that, not all.  Here the stages of the last K tiles are refilled with
// the next output tile's first K tiles as soon as every wavefront has read them, so the fill latency runs under the
// epilogue's residual loads and stores, and the workgroup never leaves the CU.
//
// Same arithmetic, same LDS images, same fragment order and the same epilogue as gemm_split_kernel: results are bit-identical
// (tests/test_ops_gpu.py runs every contraction test under DZN_GEMM_CFG=persist as well).
//
// Work order: XCD x (blockIdx.x & 7 — workgroups are dealt round-robin over the 8 XCDs) owns the contiguous item range
// [x Q, (x + 1) Q), Q = ceil(items / 8); in its step s the G / 8 workgroups of the XCD take items x Q + s G / 8 + (blockIdx.x >> 3)
// — G / 8 consecutive tiles, i.e. whole row blocks of A with all their column tiles, share one L2 at one time (the same
// locality as gemm_split_kernel's tile remap).  item = z * tiles + tile.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "gemm_epilogue.h"
#include "split.h"

namespace {

__device__ __forceinline__ int wswz(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }

// threadIdx.x behind an empty volatile asm: per-thread geometry derived from it is RECOMPUTED where it is needed instead of
// being kept in registers across the K loop and the epilogue (the kernel sits at the 256-register budget of 2 wavefronts/SIMD)
__device__ __forceinline__ int opaque_tid() {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  return t;
}

// generic -> global -> generic on a pointer field read (as a pointer) from the laundered kernel-argument segment: keeps the
// address-space inference on "global" (else every access through the descriptor copy becomes a flat_load / flat_store)
template <class T>
__device__ __forceinline__ T* as_global(T* p) {
  return (T*)(__attribute__((address_space(1))) T*)p;
}

template <int N>
__device__ __forceinline__ void wait_vm_lgkm0() {
  static_assert(N >= 0 && N < 64, "vmcnt range");
  __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | (0x7 << 4));
}

template <int BM, int BN, int WGM, int WGN, int NP, int OCC, int EJC = 0>
__global__ __launch_bounds__(WGM * WGN * 64, OCC) void gemm_persist_kernel(const dzn_gemm_desc d, const int tilesN,
                                                                            const int tiles, const int total) {
  static_assert(NP == 2, "f32h form");
  constexpr int S = 2;
  constexpr int NW = WGM * WGN;
  constexpr int BK = 32;
  constexpr int TM = BM / WGM, TN = BN / WGN;
  constexpr int MI = TM / 16, NI = TN / 16, MH = MI / 2;
  constexpr int RB = NW * 1024;
  constexpr int ACH = BM * 128 / RB;
  constexpr int WROWS = NW * 16;
  constexpr int WR = (BN + WROWS - 1) / WROWS;
  constexpr int SP = 2;
  constexpr int ABYTES = BM * 128, WPLANE = BN * 64, BUF = ABYTES + NP * WPLANE;
  constexpr int LPT = ACH + NP * WR;
  static_assert(BN % WROWS == 0, "whole W rounds");
  static_assert(BM * 128 % RB == 0 && MI % 2 == 0, "tile geometry");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* const lds_cols = reinterpret_cast<float*>(smem + S * BUF);   // the epilogue's column vectors: 3 TN floats per wavefront

  // The descriptor as the loop body sees it: re-read from the kernel-argument segment through a LAUNDERED pointer at the top
  // of every item and again before the epilogue.  Used directly, its ~60 scalar fields (and the reciprocals of every uniform
  // division) are loop invariants: hoisted out of the persistent loop they overflow the scalar register file, and the spill
  // code lands between the epilogue's stores, each reload a drain of the store queue.
  dzn_gemm_desc dd;
  auto refresh_desc = [&]() {
    static_assert(sizeof(dzn_gemm_desc) % 4 == 0, "copied as dwords");
    unsigned long long ki = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();   // d is argument 0
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
  };
  auto fresh = [](int v) {     // a uniform value the optimiser cannot see through (no hoisted division reciprocals)
    asm volatile("" : "+s"(v));
    return v;
  };
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int xcd = blockIdx.x & 7, gx = gridDim.x >> 3;
  const int Q = (total + 7) >> 3;
  const int zc = d.z_list ? d.z_count[0] : 0x7fffffff;
  // the next item of this workgroup at or after `local` whose z is inside the device-chosen subset; -1: none
  auto next_valid = [&](int local) {
    for (;; local += gx) {
      if (local >= Q) return -1;
      const int item = xcd * Q + local;
      if (item >= total) return -1;
      if (item / fresh(tiles) / dd.zdiv < zc) return local;
    }
  };
  refresh_desc();
  int local = next_valid(blockIdx.x >> 3);
  if (local < 0) return;

  // ---- per-item state: (tm, tn, z0, cz, bz) of the tile being multiplied; the n_ set belongs to the tile whose K tiles are
  // already being fetched while the current one is in its epilogue
  int tm, tn, z0;
  int64_t cz, bz;
  const float* aptr[ACH];   // per-thread source of K tile 0; a K tile adds a uniform offset
  const u16* wptr[WR];
  int ik = 0, irem = 0;
  int64_t ikoff = 0;
  auto setup_ptrs = [&](int loc) {
    const int tid = opaque_tid(), lane = tid & 63;
    const int r0 = tid >> 3;
    const int csw = (tid & 7) ^ ((r0 >> 1) & 7);
    const int wr0 = wave * 16 + (lane >> 2);
    const int wsw = (lane & 3) ^ wswz(wr0);
    const int item = xcd * Q + loc;
    const int tl = fresh(tiles), tlN = fresh(tilesN);
    const int z = item / tl, t = item - z * tl;
    tm = t / tlN;
    tn = t - tm * tlN;
    int zz0 = z / dd.zdiv;
    const int z1 = z - zz0 * dd.zdiv;
    if (dd.z_list) zz0 = dd.z_list[zz0];
    z0 = zz0;
    const float* __restrict__ A = dd.A + z0 * dd.a_z0 + z1 * dd.a_z1;
    const u16* __restrict__ W2 = reinterpret_cast<const u16*>(dd.W2h) + SP * (z0 * dd.w_z0 + z1 * dd.w_z1);
    cz = z0 * dd.c_z0 + z1 * dd.c_z1;
    bz = z0 * dd.b_z0 + z1 * dd.b_z1;
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
      int m = tm * BM + r0 + 8 * NW * i;
      m = m < dd.M ? m : dd.M - 1;
      aptr[i] = A + (dd.a_rowoff ? (int64_t)dd.a_rowoff[m] : (int64_t)m * dd.lda) + csw * 4;
    }
#pragma unroll
    for (int i = 0; i < WR; ++i) {
      int n = tn * BN + wr0 + WROWS * i;
      n = n < dd.N ? n : dd.N - 1;
      wptr[i] = W2 + (int64_t)n * SP * dd.ldw + wsw * 8;
    }
    ik = irem = 0;
    ikoff = 0;
  };
  // the scalar K cursor after `n` issued tiles (the per-thread offsets are rebuilt after the epilogue, see below)
  auto advance_k = [&](int n) {
    for (int t = 0; t < n; ++t) {
      ik += BK;
      irem += BK;
      ikoff += BK;
      if (irem == dd.kc) { irem = 0; ikoff += dd.ldk - dd.kc; }
    }
  };
  // exact power-of-two row scales from the |max| trackers (gemm_split.hip); the K loop keeps the scale, the epilogue re-derives
  // the inverse for ITS tile (two registers less across the loop)
  auto row_scales = [&](int tm_, int z0_, float (&sc)[MI], float (&inv)[MI]) {
    const int lr = opaque_tid() & 15;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      int m = tm_ * BM + wm * TM + i * 16 + lr;
      m = m < dd.M ? m : dd.M - 1;
      h2_scale(dd.a_amax[dd.amax_unit > 0 ? m / dd.amax_unit : z0_], sc[i], inv[i]);
    }
  };
  float a_scale[MI];
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
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wptr[i] + SP * ik + p * 32),
                                         (__attribute__((address_space(3))) void*)(sW + p * WPLANE + i * RB), 16, 0, 0);
    ik += BK;
    irem += BK;
    ikoff += BK;
    if (irem == dd.kc) { irem = 0; ikoff += dd.ldk - dd.kc; }
  };

  // per-lane LDS byte offsets of the fragments inside a stage (recomputed per item: dead during the epilogue)
  int woff[NI], aoff0[MI], aoff1[MI];
  auto setup_frag_offsets = [&]() {
    const int lane = opaque_tid() & 63, lr = lane & 15, lq = lane >> 4;
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
  };
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

  f32x4 acc[MI][NI];
  auto mma = [&](int i, const u32x4 (&wf)[NI][NP], const u32x4 (&af)[NP]) {
    constexpr int PW[3] = {1, 0, 0}, PA[3] = {0, 1, 0};                     // lo*hi hi*lo hi*hi
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = mfma_np<NP>(wf[j][PW[t]], af[PA[t]], acc[i][j]);
  };

  const int nk = dd.K / BK;                 // >= 2 (the dispatcher sends K >= 768 here)
  setup_ptrs(local);
  issue(0);
  issue(1);
  for (;;) {
    refresh_desc();
    {
      float inv_[MI];
      row_scales(tm, z0, a_scale, inv_);
    }
    setup_frag_offsets();
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // K tile 0 landed.  First item: tile 1 (the LPT youngest loads) may stay in flight.  Later items: the epilogue's stores
    // sit behind the two fills in the counter and loads / stores retire out of order with each other: drain.
    wait_vm_lgkm0<0>();
    __builtin_amdgcn_s_barrier();
    u32x4 wfa[NI][NP], wfb[NI][NP];
    f32x4 ar[MI][2];
    read_w(0, wfa);
    read_a(0, ar);
    int stage = 0;
    auto step = [&](int kt, const u32x4 (&wc)[NI][NP], u32x4 (&wn_)[NI][NP]) {
      const bool more = kt + 1 < nk;
#pragma unroll
      for (int i = 0; i < MH; ++i) {
        u32x4 af[NP];
        split8_h2(ar[i][0], ar[i][1], a_scale[i], af[0], af[1]);
        mma(i, wc, af);
      }
      u32x4 af2[MI - MH][NP];
#pragma unroll
      for (int i = MH; i < MI; ++i) split8_h2(ar[i][0], ar[i][1], a_scale[i], af2[i - MH][0], af2[i - MH][1]);
      const int nstage = stage ^ 1;
      __builtin_amdgcn_sched_barrier(0);
      if (more) {
        if (kt + S <= nk) wait_vm_lgkm0<(S - 2) * LPT>();
        else wait_vm_lgkm0<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + S < nk) issue(stage);
        read_w(nstage, wn_);
        read_a(nstage, ar);
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
    // ---- the next output tile's first two K tiles go out BEFORE this tile's epilogue --------------------------------
    const int c_tm = tm, c_tn = tn, c_z0 = z0;
    const int64_t c_cz = cz, c_bz = bz;
    local = next_valid(local + gx);
    if (local >= 0) setup_ptrs(local);
    __builtin_amdgcn_s_barrier();        // every wavefront has its fragments of the last K tile in registers: both stages are free
    if (local >= 0) {
      issue(0);
      issue(1);
    }
    refresh_desc();   // the epilogue's fields: live from here only
    float c_row_inv[MI];
    {
      float sc_[MI];
      row_scales(c_tm, c_z0, sc_, c_row_inv);
    }
    const int elane = opaque_tid() & 63;
    gemm_epilogue<BM, BN, TM, TN, MI, NI, true, 16, 16, EJC>(dd, acc, c_tm, c_tn, wm, wn, elane & 15, elane >> 4, c_cz, c_bz, c_z0, c_row_inv, dd.col_scale,
                                                lds_cols + wave * 3 * TN);
    if (local < 0) break;
    // the per-thread source offsets were dead during the epilogue (its register budget is the one of gemm_split_kernel):
    // rebuild them for the K tiles still to be issued
    setup_ptrs(local);
    advance_k(2);
  }
}

int g_slots = 0;   // workgroup slots of the device: 2 per CU (64 KB + 6 KB of LDS, 252 registers each)

}  // namespace

// the 128 x 128 class of launch_gemm_split (f32h): same descriptor, same results
int launch_gemm_persist(const dzn_gemm_desc& d, hipStream_t s) {
  constexpr int BM = 128, BN = 128, WGM = 4, WGN = 1, NP = 2;
  if ((d.K & 31) || (d.kc & 31) || d.ldw != d.K || d.K < 64 || !d.W2h || !d.col_scale || !d.a_amax || d.w_z0 || d.w_z1)
    return DZN_E_INVALID;
  const int tilesM = (d.M + BM - 1) / BM, tilesN = (d.N + BN - 1) / BN;
  const int nz = d.nz > 0 ? d.nz : 1;
  const int64_t total64 = (int64_t)tilesM * tilesN * nz;
  if (total64 > 0x7fffffff) return DZN_E_INVALID;
  const int total = (int)total64;
  auto kern = gemm_persist_kernel<BM, BN, WGM, WGN, NP, 2, 4>;
  const size_t lds = 2 * (BM * 128 + NP * BN * 64) + WGM * WGN * 3 * (BN / WGN) * sizeof(float);
  static unsigned long long attr_mask = 0;
  if (first_use_on_device(attr_mask)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    g_slots = 2 * cus;
  }
  int G = g_slots > 0 ? g_slots : 512;
  if (total < G) G = (total + 7) & ~7;
  int pid = -1;
  if (prof_enabled()) {
    char cls[64];
    static const bool by_shape = getenv("DZN_PROFILE_SHAPES") != nullptr;
    if (by_shape) snprintf(cls, sizeof(cls), "gemm_f32h_128x128 M%d N%d K%d z%d", d.M, d.N, d.K, d.nz);
    else snprintf(cls, sizeof(cls), "gemm_f32h_128x128");
    const double fl = d.alg_flops > 0 ? d.alg_flops * d.nz : 2.0 * d.M * d.N * d.K * d.nz;
    pid = prof_begin(s, cls, fl, gemm_alg_bytes(d, NP * 2));
  }
  hipLaunchKernelGGL(kern, dim3(G), dim3(WGM * WGN * 64), lds, s, d, tilesN, tilesM * tilesN, total);
  prof_end(pid, s);
  if (hipGetLastError() != hipSuccess) return DZN_E_HIP;
  if (d.stat_partial && d.stat_final)
    return launch_stats_finalize(d.stat_partial, d.M, tilesN * WGN, d.stat_C, d.stat_eps, d.stat_final, s);
  return DZN_OK;
}
