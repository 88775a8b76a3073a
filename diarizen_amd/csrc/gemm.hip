// gemm.hip — the MFMA contraction every dense op of the hot path runs on (gfx950).
//
//   C[m,n] = epilogue( sum_k A(m,k) * W[n,k] )          (see include/dzn_ops.h: dzn_gemm_desc)
//
// Used for: conv1..6 of the WavLM feature extractor (1-D conv as a contraction over
// overlapping channels-last rows, W2V/components.py:119-122), feature projection,
// the grouped positional conv (two-level K addressing, components.py:366-380),
// q/k/v/out projections, feed-forward layers (components.py:805-813), the Conformer
// linears / pointwise convs (conformer.py:136-144, 192-214), the DFT + mel contractions
// of kaldi fbank, all 3x3 / 1x1 convs of ResNet34 (wespeaker/resnet.py:139-144) and seg_1.
//
// Design (CDNA4): 256 threads = 4 wavefronts of 64; each wavefront owns a TM x TN
// sub-tile built from 16x16 MFMA blocks.  Two arithmetic modes share one structure,
// because both read a 16-byte K-run per lane from a 128-byte LDS row:
//   f32 : v_mfma_f32_16x16x4_f32   (exact fp32 == fmaf chain), 32 k per tile
//   bf16: v_mfma_f32_16x16x32_bf16 (fp32 accumulate),           64 k per tile
// The K order inside a 16-float block is permuted (lane group q supplies k = 4q+s at
// step s) identically for both operands, which lets every fragment be one ds_read_b128.
// LDS rows are XOR-swizzled on the 16-B slot ((row>>1)&7) so that the four 16-lane
// groups of ds_read_b128 hit 16 distinct slots (conflict free) and the ds_write_b128
// of the staging pass stays conflict free too.  Global->register->LDS double buffering
// with one barrier per K tile; workgroup ids are remapped so each XCD (private L2)
// walks a contiguous run of tiles.
#include <cstdio>
#include <cstdlib>

#include "common.h"
#include "gemm_epilogue.h"

namespace {

template <bool LOWP>
struct Frag {
  using type = f32x4;
};
template <>
struct Frag<true> {
  using type = bf16x8;
};

__device__ __forceinline__ uint4 pack_bf16x8(const float4& a, const float4& b) {
  bf16x8 v;
  v[0] = (__bf16)a.x; v[1] = (__bf16)a.y; v[2] = (__bf16)a.z; v[3] = (__bf16)a.w;
  v[4] = (__bf16)b.x; v[5] = (__bf16)b.y; v[6] = (__bf16)b.z; v[7] = (__bf16)b.w;
  return *reinterpret_cast<uint4*>(&v);
}

template <int BM, int BN, int WGM, int WGN, bool LOWP>
__global__ __launch_bounds__(256) void gemm_kernel(const dzn_gemm_desc d) {
  static_assert(WGM * WGN == 4, "4 wavefronts per workgroup");
  constexpr int BK = LOWP ? 64 : 32;   // k per tile; an LDS row is 128 B in both modes
  constexpr int KV = LOWP ? 8 : 4;     // k per 16-B LDS chunk
  constexpr int TM = BM / WGM, TN = BN / WGN;
  constexpr int MI = TM / 16, NI = TN / 16;
  constexpr int ACH = BM / 32;         // 16-B LDS chunks per thread, A tile
  constexpr int WCH = BN / 32;         // same, W tile
  constexpr int BUF = (BM + BN) * 128; // bytes per LDS stage
  using frag_t = typename Frag<LOWP>::type;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;

  // ---- tile id, XCD-contiguous remap (bijective for any grid size) ----
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
  const float* __restrict__ A = d.A + z0 * d.a_z0 + z1 * d.a_z1;
  const int64_t wz = z0 * d.w_z0 + z1 * d.w_z1;
  const float* __restrict__ W = d.W + (LOWP ? 0 : wz);
  const u16* __restrict__ W16 = reinterpret_cast<const u16*>(d.W16) + (LOWP ? wz : 0);
  const int64_t cz = z0 * d.c_z0 + z1 * d.c_z1;
  const int64_t bz = z0 * d.b_z0 + z1 * d.b_z1;

  // ---- staging assignment: thread -> (row = tid/8 + 32 i, 16-B chunk c = tid%8) ----
  const int c = tid & 7;
  const int r0 = tid >> 3;
  int64_t abase[ACH];
  bool aval[ACH];
#pragma unroll
  for (int i = 0; i < ACH; ++i) {
    const int m = tm * BM + r0 + 32 * i;
    aval[i] = m < d.M;
    const int mm = aval[i] ? m : 0;
    abase[i] = d.a_rowoff ? (int64_t)d.a_rowoff[mm] : (int64_t)mm * d.lda;
  }
  int64_t wbase[WCH];
  bool wval[WCH];
#pragma unroll
  for (int i = 0; i < WCH; ++i) {
    const int n = tn * BN + r0 + 32 * i;
    wval[i] = n < d.N;
    wbase[i] = (int64_t)(wval[i] ? n : 0) * d.ldw;
  }

  float4 ra[ACH][LOWP ? 2 : 1];
  uint4 rw[WCH];

  auto load_tile = [&](int k0) {
    const int k = k0 + c * KV;
    const bool kval = k < d.K;
    const int ch = k / d.kc;
    const int64_t koff = (int64_t)ch * d.ldk + (k - ch * d.kc);
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
      if (aval[i] && kval) {
        const float4* p = reinterpret_cast<const float4*>(A + abase[i] + koff);
        ra[i][0] = p[0];
        if constexpr (LOWP) ra[i][1] = p[1];
      } else {
        ra[i][0] = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (LOWP) ra[i][1] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
      if (wval[i] && kval) {
        if constexpr (LOWP)
          rw[i] = *reinterpret_cast<const uint4*>(W16 + wbase[i] + k);
        else
          rw[i] = *reinterpret_cast<const uint4*>(W + wbase[i] + k);
      } else {
        rw[i] = make_uint4(0u, 0u, 0u, 0u);
      }
    }
  };

  auto store_tile = [&](int buf) {
    unsigned char* sA = smem + buf * BUF;
    unsigned char* sW = sA + BM * 128;
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
      const int r = r0 + 32 * i;
      const int off = r * 128 + ((c ^ ((r >> 1) & 7)) << 4);
      if constexpr (LOWP)
        *reinterpret_cast<uint4*>(sA + off) = pack_bf16x8(ra[i][0], ra[i][1]);
      else
        *reinterpret_cast<float4*>(sA + off) = ra[i][0];
    }
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
      const int r = r0 + 32 * i;
      const int off = r * 128 + ((c ^ ((r >> 1) & 7)) << 4);
      *reinterpret_cast<uint4*>(sW + off) = rw[i];
    }
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int lr = lane & 15;  // row inside a 16x16 block
  const int lq = lane >> 4;  // lane group -> 16-B slot inside a 64-B K block

  auto compute = [&](int buf) {
    const unsigned char* sA = smem + buf * BUF;
    const unsigned char* sW = sA + BM * 128;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      frag_t af[MI], bf[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int row = wm * TM + i * 16 + lr;
        const int slot = (kb * 4 + lq) ^ ((row >> 1) & 7);
        af[i] = *reinterpret_cast<const frag_t*>(sA + row * 128 + (slot << 4));
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int row = wn * TN + j * 16 + lr;
        const int slot = (kb * 4 + lq) ^ ((row >> 1) & 7);
        bf[j] = *reinterpret_cast<const frag_t*>(sW + row * 128 + (slot << 4));
      }
      // operands swapped (W fragment as the MFMA "A"): the accumulator block is C^T, i.e. lane
      // (lr, lq) holds C[m = lr][n = 4*lq + 0..3] -> 4 consecutive columns per lane, float4 epilogue
      if constexpr (LOWP) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j][s], af[i][s], acc[i][j], 0, 0, 0);
      }
    }
  };

  // ---- main loop: register-staged double buffer, one barrier per K tile ----
  const int nk = (d.K + BK - 1) / BK;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) load_tile((kt + 1) * BK);
    compute(kt & 1);
    if (more) store_tile((kt + 1) & 1);
    __syncthreads();
  }

  gemm_epilogue<BM, BN, TM, TN, MI, NI>(d, acc, tm, tn, wm, wn, lr, lq, cz, bz, z0);
}

// ---------------------------------------------------------------------------------------------
// f32 variant with direct global->LDS loads (global_load_lds_dwordx4, CDNA4 LDS-DMA): no VGPR
// staging and no ds_write pass — the per-K-tile bubble (LDS write at ~79 B/clk + barrier) of the
// register-staged kernel shrinks to the barrier.  The LDS image must be lane-linear per wave
// instruction (1 KiB = 8 rows x 128 B), so the XOR swizzle is applied to the SOURCE chunk index
// and the same XOR on the fragment read (cdna guide rule 21).  Rows beyond M / N are clamped to
// a valid row (their accumulators are never stored).  Requires K % 32 == 0 and kc % 32 == 0.
template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(256) void gemm_glds_kernel(const dzn_gemm_desc d) {
  static_assert(WGM * WGN == 4, "4 wavefronts per workgroup");
  constexpr int BK = 32;
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
  const float* __restrict__ A = d.A + z0 * d.a_z0 + z1 * d.a_z1;
  const float* __restrict__ W = d.W + z0 * d.w_z0 + z1 * d.w_z1;
  const int64_t cz = z0 * d.c_z0 + z1 * d.c_z1;
  const int64_t bz = z0 * d.b_z0 + z1 * d.b_z1;

  // thread -> (row = 8*wave + lane/8 + 32 i, physical 16-B slot p = lane%8); the logical chunk
  // stored at slot p of row r is c = p ^ ((r >> 1) & 7)   (independent of i: 32 i >> 1 = 16 i)
  const int r0 = tid >> 3;
  const int csw = (tid & 7) ^ ((r0 >> 1) & 7);
  int64_t abase[ACH], wbase[WCH];
#pragma unroll
  for (int i = 0; i < ACH; ++i) {
    int m = tm * BM + r0 + 32 * i;
    m = m < d.M ? m : d.M - 1;
    abase[i] = (d.a_rowoff ? (int64_t)d.a_rowoff[m] : (int64_t)m * d.lda) + csw * 4;
  }
#pragma unroll
  for (int i = 0; i < WCH; ++i) {
    int n = tn * BN + r0 + 32 * i;
    n = n < d.N ? n : d.N - 1;
    wbase[i] = (int64_t)n * d.ldw + csw * 4;
  }

  auto issue = [&](int k0, int buf) {
    // k0 is a multiple of 32 and kc % 32 == 0: the whole K tile sits inside one kc chunk
    const int ch = k0 / d.kc;
    const int64_t koff = (int64_t)ch * d.ldk + (k0 - ch * d.kc);
    unsigned char* sA = smem + buf * BUF + wave * 1024;
    unsigned char* sW = smem + buf * BUF + BM * 128 + wave * 1024;
#pragma unroll
    for (int i = 0; i < ACH; ++i)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(A + abase[i] + koff),
          (__attribute__((address_space(3))) void*)(sA + i * 4096), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < WCH; ++i)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(W + wbase[i] + k0),
          (__attribute__((address_space(3))) void*)(sW + i * 4096), 16, 0, 0);
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int lr = lane & 15, lq = lane >> 4;

  // fragments of one 16-float K block (kb = 0 / 1) of the tile in LDS stage `buf`
  auto read_frags = [&](int buf, int kb, f32x4 (&af)[MI], f32x4 (&bf)[NI]) {
    const unsigned char* sA = smem + buf * BUF;
    const unsigned char* sW = sA + BM * 128;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int row = wm * TM + i * 16 + lr;
      const int slot = (kb * 4 + lq) ^ ((row >> 1) & 7);
      af[i] = *reinterpret_cast<const f32x4*>(sA + row * 128 + (slot << 4));
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int row = wn * TN + j * 16 + lr;
      const int slot = (kb * 4 + lq) ^ ((row >> 1) & 7);
      bf[j] = *reinterpret_cast<const f32x4*>(sW + row * 128 + (slot << 4));
    }
  };
  auto mma = [&](const f32x4 (&af)[MI], const f32x4 (&bf)[NI]) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j][s], af[i][s], acc[i][j], 0, 0, 0);
  };

  // Software pipeline: every batch of LDS fragment reads is followed by the 4*MI*NI MFMAs of the
  // OTHER K block, so ds_read latency and the barrier never sit in front of an idle matrix pipe:
  //   [glds tile k+1] [read kb1(k)] [mma kb0(k)] [barrier: tile k+1 landed] [read kb0(k+1)] [mma kb1(k)]
  const int nk = d.K / BK;
  f32x4 a0[MI], b0[NI], a1[MI], b1[NI];
  issue(0, 0);
  __syncthreads();  // drains the LDS-DMA (vmcnt(0)) and publishes tile 0
  read_frags(0, 0, a0, b0);
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    const bool more = kt + 1 < nk;
    if (more) issue((kt + 1) * BK, buf ^ 1);
    read_frags(buf, 1, a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    mma(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();  // tile k+1 resident; all kb1(k) reads retired before stage `buf` is refilled
    if (more) read_frags(buf ^ 1, 0, a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    mma(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
  }
  gemm_epilogue<BM, BN, TM, TN, MI, NI>(d, acc, tm, tn, wm, wn, lr, lq, cz, bz, z0);
}

template <int BM, int BN, int WGM, int WGN, bool LOWP>
int launch_cfg(const dzn_gemm_desc& d, hipStream_t s) {
  const int tilesM = (d.M + BM - 1) / BM, tilesN = (d.N + BN - 1) / BN;
  static const int lds_pad = getenv("DZN_GEMM_LDS_PAD") ? atoi(getenv("DZN_GEMM_LDS_PAD")) : 0;
  const size_t lds = 2 * (BM + BN) * 128 + lds_pad;
  auto kern = gemm_kernel<BM, BN, WGM, WGN, LOWP>;
  static unsigned long long attr_mask = 0;  // one bit per HIP device: function attributes are per device
  if (first_use_on_device(attr_mask)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if constexpr (!LOWP)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_glds_kernel<BM, BN, WGM, WGN>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  static const bool no_glds = getenv("DZN_NO_GLDS") != nullptr;
  const bool use_glds = !LOWP && !no_glds && (d.K % 32 == 0) && (d.kc % 32 == 0);
  dim3 grid(tilesM * tilesN, d.nz > 0 ? d.nz : 1, 1);
  int pid = -1;
  if (prof_enabled()) {
    char cls[64];
    static const bool by_shape = getenv("DZN_PROFILE_SHAPES") != nullptr;
    if (by_shape)
      snprintf(cls, sizeof(cls), "gemm_%s_%dx%d M%d N%d K%d z%d", LOWP ? "bf16" : "f32", BM, BN, d.M, d.N,
               d.K, d.nz);
    else
      snprintf(cls, sizeof(cls), "gemm_%s_%dx%d", LOWP ? "bf16" : "f32", BM, BN);
    const double fl = d.alg_flops > 0 ? d.alg_flops * d.nz : 2.0 * d.M * d.N * d.K * d.nz;
    pid = prof_begin(s, cls, fl, gemm_alg_bytes(d, LOWP ? 2 : 4));
  }
  if constexpr (!LOWP) {
    if (use_glds)
      hipLaunchKernelGGL((gemm_glds_kernel<BM, BN, WGM, WGN>), grid, dim3(256), lds, s, d);
    else
      hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, d);
  } else {
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, d);
  }
  prof_end(pid, s);
  if (hipGetLastError() != hipSuccess) return DZN_E_HIP;
  if (d.stat_partial && d.stat_final)
    return launch_stats_finalize(d.stat_partial, d.M, ((d.N + BN - 1) / BN) * WGN, d.stat_C, d.stat_eps, d.stat_final, s);
  return DZN_OK;
}

template <bool LOWP>
int launch_prec(const dzn_gemm_desc& d, hipStream_t s) {
  static const char* force = getenv("DZN_GEMM_CFG");   // tuning knob: force one tile shape
  if (force) {
    if (!strcmp(force, "128x32")) return launch_cfg<128, 32, 4, 1, LOWP>(d, s);
    if (!strcmp(force, "256x32")) return launch_cfg<256, 32, 4, 1, LOWP>(d, s);
    if (!strcmp(force, "256x64")) return launch_cfg<256, 64, 4, 1, LOWP>(d, s);
    if (!strcmp(force, "128x64")) return launch_cfg<128, 64, 2, 2, LOWP>(d, s);
    if (!strcmp(force, "64x64")) return launch_cfg<64, 64, 2, 2, LOWP>(d, s);
    if (!strcmp(force, "128x128")) return launch_cfg<128, 128, 2, 2, LOWP>(d, s);
  }
  if (d.N <= 32) return launch_cfg<256, 32, 4, 1, LOWP>(d, s);
  if (d.N <= 64) return launch_cfg<128, 64, 2, 2, LOWP>(d, s);
  // short contractions are epilogue-bound: narrower tiles -> 3 resident workgroups per CU hide it
  if (d.K <= 512 && (d.N % 64) == 0) return launch_cfg<128, 64, 2, 2, LOWP>(d, s);
  // pick the column-tile width that wastes the fewest padded columns (irregular pruned widths:
  // 153 -> 160, q/k/v = 192 h, FFN 96..1770); ties go to the wider tile (more reuse per A fragment)
  const int cand[4] = {192, 160, 128, 96};
  int best = 128, best_cols = 1 << 30;
  for (int c : cand) {
    const int cols = (d.N + c - 1) / c * c;
    if (cols < best_cols) { best_cols = cols; best = c; }
  }
  switch (best) {
    case 192: return launch_cfg<128, 192, 2, 2, LOWP>(d, s);
    case 160: return launch_cfg<128, 160, 2, 2, LOWP>(d, s);
    case 96: return launch_cfg<128, 96, 2, 2, LOWP>(d, s);
    default: return launch_cfg<128, 128, 2, 2, LOWP>(d, s);
  }
}

}  // namespace

int launch_gemm(const dzn_gemm_desc& din, hipStream_t s) {
  dzn_gemm_desc d = din;
  if (d.M <= 0 || d.N <= 0) return DZN_OK;
  if (d.K <= 0 || (d.K & 7)) return DZN_E_INVALID;
  if (d.kc <= 0) { d.kc = d.K; d.ldk = 0; }
  if (d.kc & 7) return DZN_E_INVALID;
  if (d.zdiv <= 0) d.zdiv = 1;
  if (d.nz <= 0) d.nz = 1;
  if (d.alpha == 0.f) d.alpha = 1.f;
  if (d.precision == DZN_PREC_BF16) {
    if (!d.W16) return DZN_E_INVALID;
#ifdef DZN_TUNING
    if (d.a_bf16) return launch_gemm_lowp(d, s);
#else
    if (d.a_bf16) return DZN_E_INVALID;    // bf16 activations: the quarantined bf16 engine mode, DZN_TUNING builds only
#endif
    if (d.c_bf16 || d.r_bf16) return DZN_E_INVALID;
    return launch_prec<true>(d, s);
  }
  if (!d.W) return DZN_E_INVALID;
  if (d.kv_planes) {
    // (r6) K / V slots as pre-split planes: written by the epilogue of the 16x16-block split contractions only, whose automatic
    // tiles (128 x 64 / 128 x 128) walk a wavefront tile in 64-column groups when N % 64 == 0; anything else is refused here
    // rather than left to an epilogue that would silently store fp32 where the attention kernel expects planes
    const bool ok = d.kv_scale && d.precision == DZN_PREC_F32_H2 && d.W3 && d.W2h && d.col_scale && d.a_amax && !(d.K & 31) &&
                    !(d.kc & 31) && d.ldw == d.K && !d.a_split3 && d.nz == 1 && !d.c_rowoff && d.kv_col0 >= 0 &&
                    d.kv_col0 < d.N && !(d.kv_col0 & 63) && !(d.N & 63) && d.kv_ld == d.N - d.kv_col0 &&
                    !(d.kv_plane_stride & 3) && d.N > 32 && !getenv("DZN_GEMM_CFG") && !getenv("DZN_NO_H2");
    if (!ok) return DZN_E_INVALID;
  }
  if (d.ln_centered && !(d.precision == DZN_PREC_F16 && d.W3 && !(d.K & 31) && !(d.kc & 31) && d.ldw == d.K && !d.a_split3))
    return DZN_E_INVALID;    // only gemm_split.hip's single-term kernel subtracts the row mean (see launch_gemm_split)
  if (d.a_split3) return prec_is_split(d.precision) ? launch_gemm_split_pre(d, s) : DZN_E_INVALID;
  if (prec_is_split(d.precision) && d.W3 && !(d.K & 31) && !(d.kc & 31) && d.ldw == d.K)
    return launch_gemm_split(d, s);
  return launch_prec<false>(d, s);
}

extern "C" int dzn_op_gemm(const dzn_gemm_desc* d, void* stream) {
  if (!d || !d->A || !d->C) return DZN_E_INVALID;
  return launch_gemm(*d, reinterpret_cast<hipStream_t>(stream));
}
