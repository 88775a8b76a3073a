// attention_planes.hip — (r6) the fused WavLM attention of attention_split.hip (NP = 2: fp16 two-term operands, three MFMA
// products, fp32 accumulate and softmax; reference arithmetic W2V/components.py:453-486, :690-725) on K / V tiles that ARRIVE
// SPLIT: the q/k/v contraction's epilogue (gemm_epilogue.h, dzn_gemm_desc.kv_planes) stores every (row, head) slot of K and V
// as the two fp16 terms of x * 2^e with its own exact power-of-two scale, once, instead of fp32 — same bytes.
//
// Why.  attention_split.hip stages a 64-key tile per 64 queries: global fp32 -> registers -> split (VALU) -> three LDS stores,
// 7 times per (window, head).  Its tile loop issues 491 VALU instructions per wavefront for 48 MFMAs (ISA count, r6): a third
// of them are the K / V splits, and 24 of its 28 global loads are the dword gathers of the transposed V image.  With three
// wavefronts per SIMD that is ~5 900 VALU issue cycles against 2 300 matrix cycles per tile round: the kernel is VALU-bound
// at MfmaUtil 22.6 % (profiles/r5_pmc_f32h_30min_b576.json), which is why removing matrix work (r5) or sharing a staged tile
// between more queries (r4) moved nothing.  Here a tile costs 8 x 16-byte loads + 8 ds_write_b128 per thread and NO split:
//   * K image: [4 d-blocks][64 keys][16 d] fp16 per plane, the 16-byte half of a row swapped by key bit 3 - every A fragment
//     is one conflict-free ds_read_b128;
//   * V image: the same row-major subtiles, unswizzled; the B fragments of P.V (k = key) are gathered by the LDS transpose
//     read `ds_read_b64_tr_b16` (each 16-lane group turns a [4 keys][16 d] block into 4 keys per lane; layout pinned on the
//     device by scripts/ubench/tr_probe.hip) - the HBM layout stays row-major, nothing is transposed in registers;
//   * per-key scales: the scores of key k are multiplied by inv_k[k] (times the query side's window scale), exactly; V's
//     scale goes into the probabilities: with a[k] = floor(log2 |max| of V row k) and A = the running maximum of a[] over the
//     keys seen so far (a function of the keys only, kept by the staging threads and handed over through LDS),
//     p'_k = p_k 2^14 * 2^(a[k] - A) <= 2^14 is what the matrix pipe multiplies with V'_k = V_k 2^(14 - a[k]):
//     O' = 2^(28 - A) sum p_k V_k.  When A grows, O is rescaled by the exact 2^(A_old - A) together with the online-softmax
//     factor.  The row sums stay in true units (sum p_k 2^14), so the final division is O' 2^(A - 14) / l.
// Everything else - work split (grid (ceil(L / 64), kept heads, B), 4 wavefronts x 16 queries), S computed transposed, the
// relative-position bias from the Toeplitz table with the gate, the log2-domain softmax - is attention_split.hip's.
#include "checked.h"
#include "common.h"
#include "split.h"

DZN_CHECKED_TU(attention_planes)

namespace {

constexpr int AP_PLANE = 64 * 128;   // one fp16 plane of a 64-key x 64-d tile (4 subtiles of [64][16])
constexpr int AP_OCC = 3;
typedef short s16x4 __attribute__((ext_vector_type(4)));

// 2^n for an integer-valued n as bits (v_exp_f32 is not promised exact); 0 below the normal range
__device__ __forceinline__ float pow2_int(float n) {
  const int e = (int)n;
  return e < -126 ? 0.f : __uint_as_float((unsigned)((e > 127 ? 127 : e) + 127) << 23);
}

// 4 keys x (16 d across the 16 lanes of a group) -> this lane's 4 keys at its d: ds_read_b64_tr_b16
__device__ __forceinline__ uint2 lds_tr16(const unsigned char* p) {
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(__attribute__((address_space(3))) void*)p);
  return __builtin_bit_cast(uint2, v);
}

// QB = 16-query blocks per wavefront.  QB = 1: 64 queries per workgroup, three workgroups per CU (attention_split.hip's work
// split).  QB = 2: 128 queries per workgroup at two workgroups per CU - every K / V fragment read from LDS and every staged tile
// serves twice the matrix work (per tile a wavefront reads the WHOLE K and V images: 32 KB for 48 MFMAs at QB = 1; twelve
// wavefronts per CU keep the LDS pipe ~50 % busy for a matrix pipe at 25 %).
// PF: the NEXT tile's K / V pieces are requested as soon as this tile's are in LDS and travel while this tile is computed (32
// more live registers: two workgroups per CU instead of three).
template <bool BIAS, int QB, bool PF>
__global__ __launch_bounds__(256, (QB == 1 && !PF) ? AP_OCC : 2) void attn_planes_kernel(
    const float* __restrict__ qkv, const uint16_t* __restrict__ planes, const float* __restrict__ kvs, float* __restrict__ out,
    const float* __restrict__ gate, const float* __restrict__ table, const int32_t* __restrict__ head_idx, int B, int L, int h,
    int Htot, int ldqkv, int ldo, int kv_ld, int64_t plane_stride, float scale, const float* __restrict__ amax, const int abl) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sK = smem;                          // 2 planes
  unsigned char* sV = smem + 2 * AP_PLANE;           // 2 planes
  float* sIK = reinterpret_cast<float*>(smem + 4 * AP_PLANE);        // [64] inverse K scale of the tile's keys x query-side inverse
  float* sW = sIK + 64;                              // [64] 2^(a[k] - A): V's per-key factor of the probabilities
  float* sA = sW + 64;                               // [4]: 2^(A_old - A) of this tile, A
  float* sT = sA + 4;                                // [2L - 1 + 64] bias table of this head, zero tail

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  // XCD-aware work order (1-D grid; workgroup x runs on XCD x % 8): the query tiles of ONE (window, head) take consecutive slots
  // of ONE XCD, so the K / V planes they all re-read (7 x 204 KB at L = 399) are fetched through the fabric once and then hit in
  // that XCD's 4 MB L2; the 8 XCDs work on 8 consecutive (window, head) pairs at a time
  const int nqt = (L + 64 * QB - 1) / (64 * QB);
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int pair = (slot / nqt) * 8 + xcd;           // = b * h + j
  if (pair >= B * h) return;
  const int qt = slot % nqt, j = pair % h, b = pair / h;
  const int64_t rowbase = (int64_t)b * L;
  const float* Qp = qkv + j * 64;
  const int kslot = j, vslot = h + j;                // 64-column slots of the planes: [k heads | v heads]
  const int nslots = kv_ld >> 6;

  int H = 0;
  if constexpr (BIAS) {
    H = head_idx[j];
    for (int i = tid; i < 2 * L - 1 + 64; i += 256) sT[i] = i < 2 * L - 1 ? table[(int64_t)H * (2 * L - 1) + i] : 0.f;
  }
  constexpr float LOG2E = 1.4426950408889634f;
  float op_scale, op_inv;                            // the query side keeps the window's scale (|max| tracker of qkv)
  h2_scale(amax[b], op_scale, op_inv);
  (void)B;
  const float qs = scale * LOG2E;
  DZN_CHECK(qt * 64 * QB < L && j < h && (!BIAS || (H >= 0 && H < Htot)), 0x613, qt);
  const int q_base = qt * 64 * QB + wave * 16 * QB;  // this wavefront's first query
  int q_row[QB];
  bool q_ok[QB];
  u32x4 qf[QB][2][2];
  float g[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    q_row[qb] = q_base + qb * 16 + lr;
    q_ok[qb] = q_row[qb] < L;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      f32x4 u = (f32x4){0.f, 0.f, 0.f, 0.f}, v = u;
      if (q_ok[qb]) {
        const float* p = Qp + (rowbase + q_row[qb]) * ldqkv + half * 32 + lq * 8;
        const float4 a = *reinterpret_cast<const float4*>(p);
        const float4 c = *reinterpret_cast<const float4*>(p + 4);
        u = (f32x4){a.x * qs, a.y * qs, a.z * qs, a.w * qs};
        v = (f32x4){c.x * qs, c.y * qs, c.z * qs, c.w * qs};
      }
      split_np<2>(u, v, op_scale, qf[qb][half]);
    }
    g[qb] = 0.f;
    if constexpr (BIAS) {
      if (q_ok[qb]) g[qb] = gate[(rowbase + q_row[qb]) * Htot + H] * LOG2E;
    }
  }

  float m_run[QB], l_run[QB];
  f32x4 O[QB][4];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    m_run[qb] = -INFINITY;
    l_run[qb] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) O[qb][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  // ---- staging assignment: granule g = tid + 256 i of a plane's tile = one 16-byte piece (8 d) of one key.  The piece and the
  // key come from the bits of g so that EIGHT CONSECUTIVE LANES write eight different 16-byte bank slots (a ds_write_b128 is
  // served 8 lanes = 128 B per cycle; rows are 32 B, subtiles 2 KB = 0 mod 128): half = g & 1, key bits 0-1 = (g >> 1) & 3,
  // d-block = (g >> 3) & 3, key bits 2-5 = g >> 5.  (The first form - key = g / 8, piece = g % 8 - put the 8 lanes of a key
  // on TWO slots: 4-way conflicts on every staging write, SQ_LDS_BANK_CONFLICT 62 M vs 8 M cycles per launch, and the kernel
  // was no faster than the one that splits in-kernel; profiles/r6_attention_pmc.txt.) ----
  const int nkt = (L + 63) / 64;
  const int half_s = tid & 1, dblk_s = (tid >> 3) & 3;
  const int skey = ((tid >> 5) << 2) | ((tid >> 1) & 3);      // + 32 i keys (g >> 5 gains 8 per i)
  const int spc = dblk_s * 2 + half_s;
  const uint16_t* gK = planes + (int64_t)kslot * 64 + spc * 8;
  const uint16_t* gV = planes + (int64_t)vslot * 64 + spc * 8;
  float a_run = -126.f;                              // staging threads (wave 0): running max of a[k] over the keys seen
  u32x4 rk[2][2], rv[2][2];                          // [plane][i]
  float sc_k = 0.f, sc_v = 0.f;                      // wave 0, lane = key: the tile's inverse scales
  auto fetch = [&](int kt) {
    const int64_t trow = rowbase + kt * 64;          // rows past the window / the batch are readable (zero tail) and masked
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int64_t ro = (trow + skey + 32 * i) * kv_ld;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        rk[p][i] = *reinterpret_cast<const u32x4*>(gK + p * plane_stride + ro);
        rv[p][i] = *reinterpret_cast<const u32x4*>(gV + p * plane_stride + ro);
      }
    }
    if (wave == 0) {
      const int64_t r = trow + lane;
      sc_k = kvs[r * nslots + kslot];
      sc_v = kvs[r * nslots + vslot];
    }
  };

  const bool wave_on = __builtin_amdgcn_readfirstlane(q_base) < L;
  if constexpr (PF) fetch(0);
  for (int kt = 0; kt < nkt; ++kt) {
    if constexpr (!PF) {
      if (!(abl & 1) || kt == 0) fetch(kt);         // abl bit 0 (measurement only, wrong results): one tile fetched, re-used
    }
    __syncthreads();  // previous tile fully consumed
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if ((abl & 1) && kt > 0) break;
      const int key = skey + 32 * i;
      const int offk = dblk_s * 2048 + key * 32 + ((half_s ^ ((key >> 3) & 1)) << 4);
      const int offv = dblk_s * 2048 + key * 32 + (half_s << 4);
      DZN_CHECK(offk + 16 <= AP_PLANE && offv + 16 <= AP_PLANE, 0x611, offk);
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        *reinterpret_cast<u32x4*>(sK + p * AP_PLANE + offk) = rk[p][i];
        *reinterpret_cast<u32x4*>(sV + p * AP_PLANE + offv) = rv[p][i];
      }
    }
    if (wave == 0) {
      // a[k] = exponent of V row k's |max| (inverse scale = 2^(a - 14)); keys past the window do not count
      const bool valid = kt * 64 + lane < L;
      float a = valid ? (float)((int)((__float_as_uint(sc_v) >> 23) & 0xff) - 127 + 14) : -126.f;
      float amx = a;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) amx = fmaxf(amx, __shfl_xor(amx, o, 64));
      const float a_new = fmaxf(a_run, amx);
      sIK[lane] = sc_k * op_inv;
      sW[lane] = pow2_int(a - a_new);                       // exact power of two <= 1 (0 for masked keys)
      if (lane == 0) { sA[0] = pow2_int(a_run - a_new); sA[1] = a_new; }
      a_run = a_new;
    }
    if constexpr (PF) {
      if (kt + 1 < nkt) fetch(kt + 1);
    }
    __syncthreads();
    if (!wave_on) continue;
    // ---- S^T = K Q^T : 4 key blocks x 2 halves of d, three products each; a K fragment serves QB query blocks ----
    f32x4 s[QB][4];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) s[qb][kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      u32x4 kf[4][2];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const int key = kb * 16 + lr;
        const int off = (2 * half + (lq >> 1)) * 2048 + key * 32 + (((lq & 1) ^ ((key >> 3) & 1)) << 4);
#pragma unroll
        for (int p = 0; p < 2; ++p) kf[kb][p] = *reinterpret_cast<const u32x4*>(sK + p * AP_PLANE + off);
      }
#pragma unroll
      for (int t = 0; t < SplitTerms<2>::N; ++t)
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
          for (int kb = 0; kb < 4; ++kb)
            s[qb][kb] = mfma_np<2>(kf[kb][SplitTerms<2>::A[t]], qf[qb][half][SplitTerms<2>::B[t]], s[qb][kb]);
    }
    // ---- exact un-scaling per key, bias, mask (last tile only), online softmax; lane owns query q_row, keys kb*16 + lq*4 + rg ----
    const int key0 = kt * 64;
    const float rA = sA[0];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      if (abl & 2) break;                            // abl bit 1: no un-scaling / bias / softmax (the scores go on as probabilities)
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const float4 ik = *reinterpret_cast<const float4*>(sIK + kb * 16 + lq * 4);
        s[qb][kb][0] *= ik.x; s[qb][kb][1] *= ik.y; s[qb][kb][2] *= ik.z; s[qb][kb][3] *= ik.w;
      }
      if constexpr (BIAS) {
        const float* tp = sT + (key0 + lq * 4 - q_row[qb] + L - 1);
        if (q_ok[qb]) {      // keys past L read the table's zero tail; their scores are masked below
#pragma unroll
          for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) s[qb][kb][rg] = fmaf(g[qb], tp[kb * 16 + rg], s[qb][kb][rg]);
        }
      }
      if (key0 + 64 > L) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg)
            if (key0 + kb * 16 + lq * 4 + rg >= L) s[qb][kb][rg] = -INFINITY;
      }
      float mx = -INFINITY;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) mx = fmaxf(mx, s[qb][kb][rg]);
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run[qb], mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_new);
      const float mshift = m_new - 14.0f;              // u = p 2^14
      float psum = 0.f;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const float4 w = *reinterpret_cast<const float4*>(sW + kb * 16 + lq * 4);
        const float u0 = __builtin_amdgcn_exp2f(s[qb][kb][0] - mshift), u1 = __builtin_amdgcn_exp2f(s[qb][kb][1] - mshift);
        const float u2 = __builtin_amdgcn_exp2f(s[qb][kb][2] - mshift), u3 = __builtin_amdgcn_exp2f(s[qb][kb][3] - mshift);
        psum += (u0 + u1) + (u2 + u3);
        s[qb][kb][0] = u0 * w.x; s[qb][kb][1] = u1 * w.y; s[qb][kb][2] = u2 * w.z; s[qb][kb][3] = u3 * w.w;    // p' = u 2^(a[k] - A)
      }
      l_run[qb] = l_run[qb] * alpha + psum;
      m_run[qb] = m_new;
      // O rows are query (lq*4 + rg): that query's alpha from lane (lq*4 + rg), times the exact 2^(A_old - A) of the tile
      const float ar = alpha * rA;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const float a = __shfl(ar, lq * 4 + rg, 64);
#pragma unroll
        for (int dblk = 0; dblk < 4; ++dblk) O[qb][dblk][rg] *= a;
      }
    }
    // ---- O += P' V' : P fragment of MFMA mm = {s[2mm], s[2mm+1]} = keys 32 mm + 16 r + 4 lq .. + 3 (r = 0, 1);
    //      a V fragment (two transpose reads) serves QB query blocks ----
#pragma unroll
    for (int mm = 0; mm < 2; ++mm) {
      if (abl & 4) break;                            // abl bit 2: no P split / V fragments / P.V products
      u32x4 pf[QB][2];
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) split_np<2>(s[qb][2 * mm], s[qb][2 * mm + 1], 1.0f, pf[qb]);
      u32x4 vf[4][2];
#pragma unroll
      for (int dblk = 0; dblk < 4; ++dblk) {
        const int off = dblk * 2048 + (32 * mm + 4 * lq + (lr >> 2)) * 32 + ((lr & 3) << 3);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const uint2 r0 = lds_tr16(sV + p * AP_PLANE + off), r1 = lds_tr16(sV + p * AP_PLANE + off + 16 * 32);
          vf[dblk][p] = (u32x4){r0.x, r0.y, r1.x, r1.y};
        }
      }
#pragma unroll
      for (int t = 0; t < SplitTerms<2>::N; ++t)
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
          for (int dblk = 0; dblk < 4; ++dblk)
            O[qb][dblk] = mfma_np<2>(pf[qb][SplitTerms<2>::B[t]], vf[dblk][SplitTerms<2>::A[t]], O[qb][dblk]);
    }
  }

  // ---- normalise and store: lane holds O[q = lq*4 + rg][d = dblk*16 + lr] = 2^(28 - A) sum p V; l = 2^14 sum p ----
  const float a_fin = nkt > 0 ? sA[1] : 0.f;
  const float cfin = pow2_int(a_fin - 14.0f);
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 16, 64);
    l_tot += __shfl_xor(l_tot, 32, 64);
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const float lt = __shfl(l_tot, lq * 4 + rg, 64);
      const int q = q_base + qb * 16 + lq * 4 + rg;
      if (q < L) {
        const float inv = cfin / lt;
        float* op = out + (rowbase + q) * ldo + j * 64 + lr;
#pragma unroll
        for (int dblk = 0; dblk < 4; ++dblk) op[dblk * 16] = O[qb][dblk][rg] * inv;
      }
    }
  }
}

// A/B switches (environment: whole runs; dzn_op_set_attention_qb / _prefetch: tests).  Measured on one box, 30-min step
// (profiles/r6_attention_planes_ab.txt): in-kernel split 132 TFLOP/s -> planes 148 -> + prefetch 153 -> + XCD order 155;
// two query blocks per wavefront 149 (slower: its softmax section runs at two wavefronts per SIMD).
int g_ap_qb = getenv("DZN_ATT_QB") ? atoi(getenv("DZN_ATT_QB")) : 1;
int g_ap_pf = getenv("DZN_ATT_PF") ? atoi(getenv("DZN_ATT_PF")) : 1;     // 1 = the next tile travels while this one is computed
#ifdef DZN_TUNING
// DZN_ATT_ABL (DZN_TUNING builds, measurement only - WRONG results): bit 0 = one K / V tile fetched and re-used (PF = 0 form),
// bit 1 = no un-scaling / bias / softmax, bit 2 = no P.V.  Record: profiles/r6_attention_ablation.txt
const int g_ap_abl = getenv("DZN_ATT_ABL") ? atoi(getenv("DZN_ATT_ABL")) : 0;
#else
constexpr int g_ap_abl = 0;
#endif

// test helper: what the contraction's epilogue writes for the K / V slots, from a plain fp32 [rows][3 h 64] tensor
__global__ __launch_bounds__(64) void kv_pack_kernel(const float* __restrict__ qkv, int ldqkv, int col0, uint16_t* __restrict__ planes,
                                                     int kv_ld, int64_t plane_stride, float* __restrict__ kvs, int64_t rows) {
  const int64_t r = blockIdx.x;
  const int slot = blockIdx.y, c = threadIdx.x;
  if (r >= rows) return;
  const float x = qkv[r * ldqkv + col0 + slot * 64 + c];
  float am = fabsf(x);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o, 64));
  float s, inv;
  h2_scale(am, s, inv);
  const float xs = x * s;
  const _Float16 hi = (_Float16)xs;
  const _Float16 lo = (_Float16)(xs - (float)hi);
  planes[r * kv_ld + slot * 64 + c] = __builtin_bit_cast(uint16_t, hi);
  planes[plane_stride + r * kv_ld + slot * 64 + c] = __builtin_bit_cast(uint16_t, lo);
  if (c == 0) kvs[r * (kv_ld >> 6) + slot] = inv;
}

}  // namespace

// planes = fp16 [2][rows + 64][kv_ld] written by the q/k/v contraction (dzn_gemm_desc.kv_planes), kvs = f32 [rows + 64][kv_ld / 64];
// rows past B * L must be readable and finite (the engine zero-fills the buffers once)
int launch_attention_planes(const float* qkv, const void* planes, int64_t plane_stride, const float* kvs, int kv_ld, float* out,
                            const float* gate, const float* table, const int32_t* head_idx, int B, int L, int h, int Htot,
                            int ldqkv, int ldo, float scale, hipStream_t s, const float* amax) {
  if (h <= 0 || B <= 0 || L <= 0) return DZN_OK;
  if ((ldqkv & 3) || (reinterpret_cast<uintptr_t>(qkv) & 15) || (reinterpret_cast<uintptr_t>(planes) & 15) || (kv_ld & 63) ||
      kv_ld < 2 * h * 64 || (plane_stride & 7) || !amax || !kvs)
    return DZN_E_INVALID;
  const bool bias = gate && table && head_idx;
  const size_t lds = 4 * AP_PLANE + (64 + 64 + 4) * sizeof(float) + (bias ? (2 * L - 1 + 64) : 0) * sizeof(float);
  if (lds > 160 * 1024) return DZN_E_INVALID;
  static unsigned long long attr_mask = 0;
  if (first_use_on_device(attr_mask)) {
    const void* ks[6] = {reinterpret_cast<const void*>(attn_planes_kernel<true, 1, false>), reinterpret_cast<const void*>(attn_planes_kernel<false, 1, false>),
                         reinterpret_cast<const void*>(attn_planes_kernel<true, 2, false>), reinterpret_cast<const void*>(attn_planes_kernel<false, 2, false>),
                         reinterpret_cast<const void*>(attn_planes_kernel<true, 1, true>), reinterpret_cast<const void*>(attn_planes_kernel<false, 1, true>)};
    for (const void* k : ks) (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const int qb = g_ap_qb == 2 ? 2 : 1;
  const int nqt = (L + 64 * qb - 1) / (64 * qb);
  dim3 grid((unsigned)(8 * nqt * (((int64_t)B * h + 7) / 8)));     // XCD-aware 1-D order, see the kernel
  const int pid = prof_begin(s, bias ? "attention_relpos_f32h" : "attention_f32h", 4.0 * B * h * (double)L * L * 64.0,
                             (double)B * L * h * 64.0 * 4.0 * 4.0 + (gate ? (double)B * L * Htot * 4.0 : 0.0));   // q, k, v in + out, once
  const uint16_t* pl = reinterpret_cast<const uint16_t*>(planes);
#define DZN_AP(BV, QV, PV)                                                                                                      \
  hipLaunchKernelGGL((attn_planes_kernel<BV, QV, PV>), grid, dim3(256), lds, s, qkv, pl, kvs, out, gate, table, head_idx, B, L, h, Htot, \
                     ldqkv, ldo, kv_ld, plane_stride, scale, amax, g_ap_abl)
  if (bias && qb == 2) DZN_AP(true, 2, false);
  else if (bias && g_ap_pf) DZN_AP(true, 1, true);
  else if (bias) DZN_AP(true, 1, false);
  else if (qb == 2) DZN_AP(false, 2, false);
  else if (g_ap_pf) DZN_AP(false, 1, true);
  else DZN_AP(false, 1, false);
#undef DZN_AP
  prof_end(pid, s);
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}

// kernel-level entry point (tests): packs the K / V slots of a plain fp32 qkv = [rows][3 h 64] exactly as the contraction's
// epilogue does, then runs the planes kernel.  planes / kvs: caller's scratch, fp16 [2][B L + 64][2 h 64] and f32 [B L + 64][2 h],
// zero-filled by the caller.
// test / A-B switches: 16-query blocks per wavefront (1 or 2); prefetch of the next tile across the compute (0 / 1)
extern "C" int dzn_op_set_attention_qb(int32_t qb) {
  g_ap_qb = qb == 2 ? 2 : 1;
  return DZN_OK;
}
extern "C" int dzn_op_set_attention_prefetch(int32_t on) {
  g_ap_pf = on != 0;
  return DZN_OK;
}

extern "C" int dzn_op_attention_planes(const float* qkv, float* out, const float* gate, const float* table, const int32_t* head_idx,
                                       int32_t B, int32_t L, int32_t h, int32_t Htot, int32_t ldqkv, int32_t ldo, float scale,
                                       const float* amax, void* planes, float* kvs, void* stream) {
  if (!qkv || !out || !amax || !planes || !kvs || h <= 0) return DZN_E_INVALID;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int kv_ld = 2 * h * 64;
  const int64_t rows = (int64_t)B * L, plane_stride = (rows + 64) * kv_ld;
  hipLaunchKernelGGL(kv_pack_kernel, dim3((unsigned)rows, 2 * h), dim3(64), 0, s, qkv, ldqkv, h * 64, reinterpret_cast<uint16_t*>(planes),
                     kv_ld, plane_stride, kvs, rows);
  return launch_attention_planes(qkv, planes, plane_stride, kvs, kv_ld, out, gate, table, head_idx, B, L, h, Htot, ldqkv, ldo, scale, s,
                                 amax);
}
