// attention_split.hip — the fused attention of attention.hip with both contractions (S = Q K^T and
// O = P V) on the bf16 matrix pipe through the exact 3-way operand split of split.h
// (DZN_PREC_F32_SPLIT): 6 bf16 MFMA products per 16x16x32 block, fp32 accumulate, fp32 softmax.
//
// Same reference arithmetic and work split as attention.hip (W2V/components.py:453-486, :690-725;
// conformer.py:47-71; grid (ceil(L/64), kept heads, B), 4 wavefronts x 16 queries, flash-style
// online softmax, S computed transposed so a lane owns one query column).  What changes:
//   * Q is split once per workgroup (registers); K and V are split ONCE per 64-key tile by the
//     staging pass (each element by exactly one thread) and stored as three bf16 planes in LDS;
//     P (the probabilities, already in registers in MFMA operand order) is split per wavefront.
//   * the k dimension of P.V is the key index, so V is staged TRANSPOSED: thread (d, group)
//     gathers the 8 keys one lane group feeds to one MFMA (coalesced across d) and writes them as
//     one 16-byte run of the [d][key] plane — the P fragment needs no data movement at all.
//   * planes are [64][128 B] images with the 16-B slot XOR ((row >> 1) & 7): every fragment is one
//     conflict-free ds_read_b128.
#include "checked.h"
#include "common.h"
#include "split.h"

DZN_CHECKED_TU(attention_split)

namespace {

constexpr int ATT_PLANE = 64 * 128;  // one bf16 plane of a 64 x 64 tile
constexpr int ATT_OCC = 3;           // wavefronts per SIMD the register budget is held to (LDS allows 3 workgroups/CU)

// NP = 3: bf16 three-term split, six products.  NP = 2 (DZN_PREC_F32_H2): fp16 two-term split, three products:
//   q, k, v are scaled by the exact power of two s from the window's |max| tracker of the qkv tensor (amax[b]);
//   the scores come out of the MFMA times s^2 and are multiplied back (exactly) before bias / softmax;
//   the probabilities are produced as p' = 2^14 p (one exp2 with +14 in the exponent, p' <= 2^14 fits fp16) and the
//   final normalisation divides by s * sum p', so neither scale costs an instruction in the inner loops.
template <bool BIAS, int NP>
__global__ __launch_bounds__(256, ATT_OCC) void attn_split_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                         const float* __restrict__ gate,
                                                         const float* __restrict__ table,
                                                         const int32_t* __restrict__ head_idx, int B, int L,
                                                         int h, int Htot, int ldqkv, int ldo, float scale,
                                                         const float* __restrict__ amax, const int noskip) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sK = smem;                     // NP planes [64 keys][64 d]
  unsigned char* sV = smem + NP * ATT_PLANE;    // NP planes [64 d][64 keys in MFMA order]
  float* sT = reinterpret_cast<float*>(smem + 2 * NP * ATT_PLANE);  // [2L-1] bias table of this head

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int qt = blockIdx.x, j = blockIdx.y, b = blockIdx.z;
  const int64_t rowbase = (int64_t)b * L;
  const float* Qp = qkv + j * 64;
  const float* Kp = qkv + (h + j) * 64;
  const float* Vp = qkv + (2 * h + j) * 64;

  int H = 0;
  if constexpr (BIAS) {
    H = head_idx[j];
    for (int i = tid; i < 2 * L - 1; i += 256) sT[i] = table[(int64_t)H * (2 * L - 1) + i];
  }

  // scores are kept in the log2 domain (q and the bias gate carry a factor log2 e), so the softmax
  // exponentials are bare v_exp_f32: exp(s - m) == exp2(s log2e - m log2e)
  constexpr float LOG2E = 1.4426950408889634f;
  float op_scale = 1.f, op_inv = 1.f;           // NP = 2: power-of-two scale of q / k / v and its inverse
  if constexpr (NP == 2) h2_scale(amax[blockIdx.z], op_scale, op_inv);
  const float qs = scale * LOG2E;
  const float s_inv = op_inv * op_inv;          // scores leave the MFMA scaled by op_scale^2
  constexpr float PSHIFT = NP == 2 ? 14.0f : 0.0f;   // p' = 2^PSHIFT p
  // ---- Q fragments (B operand of S^T = K Q^T): lane (query lr, group lq) holds d = 32 half + 8 lq .. +7 ----
  DZN_CHECK(qt * 64 < L && j < h && (!BIAS || (H >= 0 && H < Htot)), 0x603, qt);              // query tile / head inside the launch
  const int q_row = qt * 64 + wave * 16 + lr;
  const bool q_ok = q_row < L;
  u32x4 qf[2][NP];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    f32x4 u = (f32x4){0.f, 0.f, 0.f, 0.f}, v = u;
    if (q_ok) {
      const float* p = Qp + (rowbase + q_row) * ldqkv + half * 32 + lq * 8;
      const float4 a = *reinterpret_cast<const float4*>(p);
      const float4 c = *reinterpret_cast<const float4*>(p + 4);
      u = (f32x4){a.x * qs, a.y * qs, a.z * qs, a.w * qs};
      v = (f32x4){c.x * qs, c.y * qs, c.z * qs, c.w * qs};
    }
    split_np(u, v, op_scale, qf[half]);
  }
  float g = 0.f;
  if constexpr (BIAS) {
    if (q_ok) g = gate[(rowbase + q_row) * Htot + H] * LOG2E;
  }

  float m_run = -INFINITY, l_run = 0.f;
  f32x4 O[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) O[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- staging assignment ----
  // K: item = tid + 256 i -> (key = tid / 8 + 32 i, 8-d chunk = tid % 8): 32 contiguous bytes per thread
  // V: item = tid + 256 i -> (d = tid % 64, group = wave + 4 i, i.e. MFMA mm = i, lane group lq' = wave):
  //    the 8 keys (2 mm + e/4) * 16 + 4 lq' + e%4 that lane group lq' feeds to P.V MFMA mm.
  // Rows past L are clamped to row L-1: their scores are masked to -inf (p = 0), so any finite data do.
  const int nkt = (L + 63) / 64;
  const int koff = (tid >> 3) * ldqkv + (tid & 7) * 8;   // + (64 kt + 32 i) ldqkv
  const int voff = wave * 4 * ldqkv + (tid & 63);        // + (64 kt + const(i, e)) ldqkv
  f32x4 rk[2][2], rv[2][2];
  auto fetch = [&](int kt) {
    const int64_t trow = rowbase + kt * 64;               // wave-uniform
    const bool full = kt * 64 + 64 <= L;
    const float* kt_base = Kp + trow * ldqkv;
    const float* vt_base = Vp + trow * ldqkv;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int ko = koff + 32 * i * ldqkv;
      if (!full) {
        const int key = (tid >> 3) + 32 * i;
        if (kt * 64 + key >= L) ko += (L - 1 - kt * 64 - key) * ldqkv;
      }
      const float4 a = *reinterpret_cast<const float4*>(kt_base + ko);
      const float4 c = *reinterpret_cast<const float4*>(kt_base + ko + 4);
      rk[i][0] = (f32x4){a.x, a.y, a.z, a.w};
      rk[i][1] = (f32x4){c.x, c.y, c.z, c.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int ce = (2 * i + (e >> 2)) * 16 + (e & 3);   // compile-time part of the key index
        int vo = voff + ce * ldqkv;
        if (!full) {
          const int key = ce + wave * 4;
          if (kt * 64 + key >= L) vo += (L - 1 - kt * 64 - key) * ldqkv;
        }
        const float x = vt_base[vo];
        if (e < 4) rv[i][0][e] = x;
        else rv[i][1][e - 4] = x;
      }
    }
  };

  // no register prefetch across the tile: 3 resident workgroups per CU hide the global-load latency
  // better than 32 more live registers (2 workgroups) do — measured 91 -> 102 TFLOP/s
  // (r4) The K / V tile is staged — fetched, split, stored — once per 64 queries, i.e. 7 times per
  // (window, head), and the r3 PMC pass counts 2.79 GB per launch for 0.70 GB algorithmic.  Sharing a staged tile between
  // 128 or 256 queries (8 / 16 wavefronts per workgroup, every thread staging half / a quarter as much) was built and
  // measured: 132 / 134 TFLOP/s against 138 for this form in the same run (profiles/r4_attention_qw_probe.txt; the
  // templated kernel is in the history at 2ee-series commit "attention: query blocks per workgroup templated"), step
  // unchanged.  The re-reads are L2 / MALL hits and the split is not what the loop waits for; the kernel stays as it was.
  // (r5) L = 399 = 6 x 64 + 15: the seventh query tile has 15 queries and its wavefronts 1-3 own none.  They still stage and take the
  // barriers but skip scores, softmax and P.V (rows that are never stored): same stored bits (`noskip` = the r2-r4 kernel, for the
  // test), 6 registers fewer, 10.7 % of the kernel's matrix work gone — and the SAME time (62.3 ms per step either way,
  // profiles/r5_attention_skip.txt): the tile loop is bound by the latency of its global fetch -> split -> LDS chain, not by the
  // matrix / softmax work.  Skipping the three fully masked key blocks of the last key tile as well needs a second copy of the tile
  // body (compile-time block count: run-time guards spill 61 registers); that form spills 10 and measured 3 % slower — removed.
  // Also probed and removed: requesting the NEXT tile's K / V rows in the middle of the tile (after the scores, when the K
  // fragments are dead) so that the fetch overlaps softmax + P.V — 14-16 spilled registers, 16 % slower (r5_attention_skip.txt).
  const bool wave_on = noskip || __builtin_amdgcn_readfirstlane(qt * 64 + wave * 16) < L;
  for (int kt = 0; kt < nkt; ++kt) {
    fetch(kt);
    __syncthreads();  // previous tile fully consumed
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int item = tid + 256 * i;
      u32x4 pf[NP];
      {
        const int key = item >> 3, slot = item & 7;
        const int off = key * 128 + ((slot ^ ((key >> 1) & 7)) << 4);
        DZN_CHECK(key < 64 && off + 16 <= ATT_PLANE, 0x601, off);                                  // staged K chunk inside its plane
        split_np(rk[i][0], rk[i][1], op_scale, pf);
#pragma unroll
        for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4*>(sK + p * ATT_PLANE + off) = pf[p];
      }
      {
        const int d = item & 63, grp = item >> 6;
        const int off = d * 128 + ((grp ^ ((d >> 1) & 7)) << 4);
        DZN_CHECK(grp < 8 && off + 16 <= ATT_PLANE, 0x602, off);                                   // staged V group inside its plane
        split_np(rv[i][0], rv[i][1], op_scale, pf);
#pragma unroll
        for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4*>(sV + p * ATT_PLANE + off) = pf[p];
      }
    }
    __syncthreads();
    if (!wave_on) continue;
    // ---- S^T = K Q^T : 4 key blocks x 2 halves of d, six products each ----
    f32x4 s[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) s[kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      u32x4 kf[4][NP];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const int key = kb * 16 + lr;
        const int off = key * 128 + (((half * 4 + lq) ^ ((key >> 1) & 7)) << 4);
#pragma unroll
        for (int p = 0; p < NP; ++p) kf[kb][p] = *reinterpret_cast<const u32x4*>(sK + p * ATT_PLANE + off);
      }
      // product-major order (smallest terms first): 4 independent accumulators between dependent MFMAs
#pragma unroll
      for (int t = 0; t < SplitTerms<NP>::N; ++t)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
          s[kb] = mfma_np<NP>(kf[kb][SplitTerms<NP>::A[t]], qf[half][SplitTerms<NP>::B[t]], s[kb]);
    }
    if constexpr (NP == 2) {
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) s[kb] = s[kb] * s_inv;     // exact: s_inv is a power of two
    }

    // ---- bias, mask (last tile only), online softmax; lane owns query q_row, keys kb*16 + lq*4 + rg ----
    const int key0 = kt * 64;
    if constexpr (BIAS) {
      const float* tp = sT + (key0 + lq * 4 - q_row + L - 1);
      if (q_ok) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int key = key0 + kb * 16 + lq * 4 + rg;
            if (key < L) s[kb][rg] = fmaf(g, tp[kb * 16 + rg], s[kb][rg]);
          }
      }
    }
    if (key0 + 64 > L) {
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
          if (key0 + kb * 16 + lq * 4 + rg >= L) s[kb][rg] = -INFINITY;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) mx = fmaxf(mx, s[kb][rg]);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const float p = __builtin_amdgcn_exp2f(s[kb][rg] - m_new + PSHIFT);
        s[kb][rg] = p;
        psum += p;
      }
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;

    // O rows are query (lq*4 + rg): fetch that query's alpha from lane (lq*4 + rg)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const float a = __shfl(alpha, lq * 4 + rg, 64);
#pragma unroll
      for (int dblk = 0; dblk < 4; ++dblk) O[dblk][rg] *= a;
    }

    // ---- O += P V : P fragment of MFMA mm = {s[2mm], s[2mm+1]} (the order V was staged in) ----
#pragma unroll
    for (int mm = 0; mm < 2; ++mm) {
      u32x4 pf[NP];
      split_np(s[2 * mm], s[2 * mm + 1], 1.0f, pf);
      u32x4 vf[4][NP];
#pragma unroll
      for (int dblk = 0; dblk < 4; ++dblk) {
        const int d = dblk * 16 + lr;
        const int off = d * 128 + (((mm * 4 + lq) ^ ((d >> 1) & 7)) << 4);
#pragma unroll
        for (int p = 0; p < NP; ++p) vf[dblk][p] = *reinterpret_cast<const u32x4*>(sV + p * ATT_PLANE + off);
      }
#pragma unroll
      for (int t = 0; t < SplitTerms<NP>::N; ++t)
#pragma unroll
        for (int dblk = 0; dblk < 4; ++dblk)
          O[dblk] = mfma_np<NP>(pf[SplitTerms<NP>::B[t]], vf[dblk][SplitTerms<NP>::A[t]], O[dblk]);
    }
  }

  // ---- normalise and store: lane holds O[q = lq*4 + rg][d = dblk*16 + lr] ----
  float l_tot = l_run + __shfl_xor(l_run, 16, 64);
  l_tot += __shfl_xor(l_tot, 32, 64);
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) {
    const float lt = __shfl(l_tot, lq * 4 + rg, 64);
    const int q = qt * 64 + wave * 16 + lq * 4 + rg;
    if (q < L) {
      const float inv = op_inv / lt;      // NP = 2: lt = 2^14 sum p and O carries 2^14 op_scale
      float* op = out + (rowbase + q) * ldo + j * 64 + lr;
#pragma unroll
      for (int dblk = 0; dblk < 4; ++dblk) op[dblk * 16] = O[dblk][rg] * inv;
    }
  }
}

int g_att_noskip = getenv("DZN_ATT_NOSKIP") != nullptr;     // environment: A/B of whole runs; dzn_op_set_attention_noskip: tests
}  // namespace

// test switch: 1 = the query-less wavefronts of a partial last query tile compute as r2-r4 did (same stored bits)
extern "C" int dzn_op_set_attention_noskip(int32_t on) {
  g_att_noskip = on != 0;
  return DZN_OK;
}

int launch_attention_split(const float* qkv, float* out, const float* gate, const float* table,
                           const int32_t* head_idx, int B, int L, int h, int Htot, int ldqkv, int ldo,
                           float scale, hipStream_t s, const float* amax) {
  if (h <= 0 || B <= 0 || L <= 0) return DZN_OK;
  const int noskip = g_att_noskip;
  if ((ldqkv & 3) || (reinterpret_cast<uintptr_t>(qkv) & 15)) return DZN_E_INVALID;
  const bool bias = gate && table && head_idx;
  const int np = amax ? 2 : 3;
  const size_t lds = 2 * np * ATT_PLANE + (bias ? (2 * L - 1) : 0) * sizeof(float);
  if (lds > 160 * 1024) return DZN_E_INVALID;
  static unsigned long long attr_mask = 0;  // one bit per HIP device: function attributes are per device
  if (first_use_on_device(attr_mask)) {
    const void* ks[4] = {reinterpret_cast<const void*>(attn_split_kernel<true, 3>), reinterpret_cast<const void*>(attn_split_kernel<false, 3>),
                         reinterpret_cast<const void*>(attn_split_kernel<true, 2>), reinterpret_cast<const void*>(attn_split_kernel<false, 2>)};
    for (const void* k : ks) (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  dim3 grid((L + 63) / 64, h, B);
  const int pid = prof_begin(s, bias ? (amax ? "attention_relpos_f32h" : "attention_relpos_f32s") : (amax ? "attention_f32h" : "attention_f32s"),
                             4.0 * B * h * (double)L * L * 64.0,
                             (double)B * L * h * 64.0 * 4.0 * 4.0 + (gate ? (double)B * L * Htot * 4.0 : 0.0));   // q, k, v in + out, once
#define DZN_ATT(BV, NPV)                                                                                            \
  hipLaunchKernelGGL((attn_split_kernel<BV, NPV>), grid, dim3(256), lds, s, qkv, out, gate, table, head_idx, B, L, h, \
                     Htot, ldqkv, ldo, scale, amax, noskip)
  if (bias && amax) DZN_ATT(true, 2);
  else if (bias) DZN_ATT(true, 3);
  else if (amax) DZN_ATT(false, 2);
  else DZN_ATT(false, 3);
#undef DZN_ATT
  prof_end(pid, s);
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}

// kernel-level entry point of the fp16 two-term variant (tests): amax = f32 [B], per-window |max| of qkv
extern "C" int dzn_op_attention_h2(const float* qkv, float* out, const float* gate, const float* table,
                                   const int32_t* head_idx, int32_t B, int32_t L, int32_t h, int32_t Htot, int32_t ldqkv,
                                   int32_t ldo, float scale, const float* amax, void* stream) {
  if (!qkv || !out || !amax) return DZN_E_INVALID;
  return launch_attention_split(qkv, out, gate, table, head_idx, B, L, h, Htot, ldqkv, ldo, scale,
                                reinterpret_cast<hipStream_t>(stream), amax);
}
