// gemm_epilogue.h — the fused epilogue shared by the fp32 contraction kernels (gemm.hip,
// gemm_split*.hip): bias, activation, alpha, residual, post-ReLU, store, layer-weighted-sum
// accumulate.  The accumulator block of lane (lr, lq) is C[m = ..+lr][n0 .. n0+3] (operands are
// fed to the MFMA swapped, so a lane owns 4 consecutive columns -> float4 traffic).
#pragma once
#include <type_traits>

#include "common.h"
#include "split.h"

namespace {

// General form: every feature test and every column guard sits inside the (i, j) block loop.  Correct for any N / stride,
// but each block then is its own chain of load -> s_waitcnt vmcnt(0) -> use (col_scale, ln_colsum, bias, R, WS ...):
// 16 blocks x 3-5 dependent round trips.  Kept for unaligned shapes only; gemm_epilogue() below is the one that runs.
// RS / CS (r4): lane -> element map of the accumulator blocks.  16 / 16 = the 16x16 MFMA forms (lane (lr = l & 15, lq = l >> 4) of
// block (i, j) holds row 16 i + lr, columns 16 j + 4 lq .. + 3); 32 / 8 = the 32x32x16 form viewed as 8-column blocks
// (lane (lr = l & 31, lq = l >> 5) of block (i, j) holds row 32 i + lr, columns 8 j + 4 lq .. + 3).
template <int BM, int BN, int TM, int TN, int MI, int NI, int RS = 16, int CS = 16>
__device__ __forceinline__ void gemm_epilogue_general(const dzn_gemm_desc& d, f32x4 (&acc)[MI][NI], int tm, int tn,
                                              int wm, int wn, int lr, int lq, int64_t cz, int64_t bz, int z0 = 0,
                                              const float* row_inv = nullptr, const float* col_scale = nullptr) {
  // ---- epilogue: lane (lr, lq) of block (i, j) holds row m = ..+lr, columns n0..n0+3 ----
  const float* __restrict__ bias = d.bias ? d.bias + bz : nullptr;
  const bool vec = (((int64_t)d.N | d.ldc | d.ldws | cz | bz) & 3) == 0;
  // running |max| of what this lane stores, per unit (d.c_amax[unit]: the scale of the consumer's fp16 split);
  // unit = m / amax_unit (rows of one window) or the z batch index
  float amax = 0.f, amax_hi = 0.f;
  // a wavefront's TM consecutive rows span at most two units when amax_unit >= TM: two running maxima (rows below /
  // at or above the first unit boundary inside the wave tile) and one tracker update each, instead of one per row
  const int wrow0 = tm * BM + wm * TM;
  const bool two_unit = d.c_amax && d.amax_unit >= TM;
  const int unit0 = two_unit ? (wrow0 < d.M ? wrow0 : d.M - 1) / d.amax_unit : 0;
  const int boundary = two_unit ? (unit0 + 1) * d.amax_unit : 0;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = tm * BM + wm * TM + i * RS + lr;
    if (m >= d.M) continue;
    const int64_t crow = cz + (d.c_rowoff ? (int64_t)d.c_rowoff[m] : (int64_t)m * d.ldc);
    // LayerNorm folded into the weights: finish it with the row statistics (dzn_ops.h)
    const float acc_scale = row_inv ? row_inv[i] : 1.f;
    float amax_row = 0.f;
    float st_s = 0.f, st_q = 0.f;   // partial (sum, sum of squares) of this row over the wave tile's columns (d.stat_partial)
    float ln_mu = 0.f, ln_rs = 1.f;
    if (d.ln_stats) {
      const float2 st = *reinterpret_cast<const float2*>(d.ln_stats + 2 * (int64_t)m);
      ln_mu = d.ln_centered ? 0.f : st.x;      // centered: the kernel already multiplied (x - mean)
      ln_rs = st.y;
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n0 = tn * BN + wn * TN + j * CS + lq * 4;
      if (n0 >= d.N) continue;
      f32x4 v = acc[i][j];
      if (vec && n0 + 3 < d.N) {
        if (col_scale) {   // fp16 two-term operands were scaled by exact powers of two: undo (exact)
          const float4 c4 = *reinterpret_cast<const float4*>(col_scale + n0);
          v[0] *= acc_scale * c4.x; v[1] *= acc_scale * c4.y; v[2] *= acc_scale * c4.z; v[3] *= acc_scale * c4.w;
        }
        if (d.ln_stats) {
          const float4 s4 = *reinterpret_cast<const float4*>(d.ln_colsum + n0);
          v[0] = ln_rs * (v[0] - ln_mu * s4.x); v[1] = ln_rs * (v[1] - ln_mu * s4.y);
          v[2] = ln_rs * (v[2] - ln_mu * s4.z); v[3] = ln_rs * (v[3] - ln_mu * s4.w);
        }
        if (bias) {
          const float4 b4 = *reinterpret_cast<const float4*>(bias + n0);
          v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], d.act) * d.alpha;
        if (d.R) {
          const float4 r4 = *reinterpret_cast<const float4*>(d.R + crow + n0);
          v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
        }
        if (d.post_relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        *reinterpret_cast<float4*>(d.C + crow + n0) = make_float4(v[0], v[1], v[2], v[3]);
        amax_row = fmaxf(fmaxf(amax_row, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        st_s += (v[0] + v[1]) + (v[2] + v[3]);
        st_q = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], fmaf(v[3], v[3], st_q))));
        if (d.WS) {
          float4* w = reinterpret_cast<float4*>(d.WS + (int64_t)m * d.ldws + n0);
          float4 a = d.ws_init ? make_float4(0.f, 0.f, 0.f, 0.f) : *w;
          a.x += d.ws_w * v[0]; a.y += d.ws_w * v[1]; a.z += d.ws_w * v[2]; a.w += d.ws_w * v[3];
          *w = a;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int n = n0 + e;
          if (n >= d.N) continue;
          float x = v[e];
          if (col_scale) x *= acc_scale * col_scale[n];
          if (d.ln_stats) x = ln_rs * (x - ln_mu * d.ln_colsum[n]);
          if (bias) x += bias[n];
          x = apply_act(x, d.act) * d.alpha;
          if (d.R) x += d.R[crow + n];
          if (d.post_relu) x = fmaxf(x, 0.f);
          d.C[crow + n] = x;
          amax_row = fmaxf(amax_row, fabsf(x));
          st_s += x;
          st_q = fmaf(x, x, st_q);
          if (d.WS) {
            float* w = d.WS + (int64_t)m * d.ldws + n;
            *w = d.ws_init ? d.ws_w * x : (*w + d.ws_w * x);
          }
        }
      }
    }
    if (d.stat_partial) {
      // LayerNorm statistics of the rows this contraction WRITES, for the next (folded) LayerNorm: every wavefront
      // leaves (sum, sum of squares) of its TN columns; stats_finalize_kernel adds the tilesN * (BN / TN) partials of a
      // row in a fixed order (deterministic) and turns them into (mean, rstd) — the separate row_stats pass over the
      // tensor disappears.  The row lives in lanes lr, lr + 16, lr + 32, lr + 48.
      float s1 = st_s, q1 = st_q;
      if constexpr (RS == 16) {
        s1 += __shfl_xor(s1, 16, 64);
        q1 += __shfl_xor(q1, 16, 64);
      }
      s1 += __shfl_xor(s1, 32, 64);
      q1 += __shfl_xor(q1, 32, 64);
      if (lq == 0) {
        const int P = ((d.N + BN - 1) / BN) * (BN / TN);
        reinterpret_cast<float2*>(d.stat_partial)[(int64_t)m * P + tn * (BN / TN) + wn] = make_float2(s1, q1);
      }
    }
    if (d.c_amax) {
      if (two_unit) {
        if (m < boundary) amax = fmaxf(amax, amax_row);
        else amax_hi = fmaxf(amax_hi, amax_row);
      } else if (d.amax_unit > 0) {
        // the row lives in lanes lr, lr + 16, lr + 32, lr + 48 (all of them took this branch: m depends on lr only)
        float r = amax_row;
        if constexpr (RS == 16) r = fmaxf(r, __shfl_xor(r, 16, 64));
        r = fmaxf(r, __shfl_xor(r, 32, 64));
        if (lq == 0) track_amax_lane(d.c_amax + m / d.amax_unit, r);
      } else {
        amax = fmaxf(amax, amax_row);
      }
    }
  }
  if (two_unit) {
    track_amax(d.c_amax + unit0, amax);
    if (wrow0 + TM > boundary && boundary < d.M) track_amax(d.c_amax + unit0 + 1, amax_hi);   // wave-uniform
  } else if (d.c_amax && d.amax_unit <= 0) {
    track_amax(d.c_amax + z0, amax);
  }
}

// largest divisor of ni that is <= cap: column blocks per epilogue batch
constexpr int epi_chunk(int ni, int cap) {
  int best = 1;
  for (int c = 1; c <= cap; ++c)
    if (ni % c == 0) best = c;
  return best;
}

// (r4) Write-through (`sc1`) stores for C and the layer-weighted sum — so that the outputs, which nothing re-reads before
// they have left the 4 MB L2, stop evicting the operand tiles — were built and measured: step 1063-1065 ms against
// 1053-1055 ms with plain stores (gpurun r4j, DZN_GEMM_WT probe; profiles/r4_gemm_wt_probe.txt).  Plain stores stay.
// Likewise NON-TEMPORAL loads of the residual / layer-weighted sum and non-temporal stores of C / WS (`__builtin_nontemporal_*`,
// the `nt` cache policy): step 1076 ms against 1053-1057, both contraction classes 3-4 % slower (profiles/r4_epilogue_nt_probe.txt)
// — the next kernel re-reads C, and what it finds in the Infinity Cache today it then fetches from HBM.
template <int ACT>
__device__ __forceinline__ float apply_act_c(float v) {
  if constexpr (ACT == DZN_ACT_GELU) return gelu_erf(v);
  else if constexpr (ACT == DZN_ACT_SWISH) return swishf_(v);
  else if constexpr (ACT == DZN_ACT_RELU) return fmaxf(v, 0.0f);
  else return v;
}

// The epilogue that runs (round 3).  r2's form (gemm_epilogue_general) cost a fixed ~400 us per launch on the M = 149 k,
// N = 1024 contractions whatever K was — 1.2 GB of C + R moved at 3 TB/s, a third of the K = 1024 launch — because
// every (i, j) block waited for its own loads one after the other.  Here:
//   * the column vectors (col_scale, ln_colsum, bias) are fetched ONCE per wavefront — into registers, or, when the
//     caller hands over a wavefront-private LDS scratch (`lds_cols`, 3 x TN floats; wide tiles), into LDS, from where
//     the element loop reads them back as broadcast ds_read_b128s;
//   * the tile is walked in batches of one 16-row block x JC column blocks; the residual float4s of batch b+1 are
//     issued BEFORE batch b is computed and stored (two-deep ring), so one global round trip is exposed per wavefront,
//     not one per block;
//   * the layer-weighted-sum read-modify-write is a second pass with the same two-deep ring;
//   * out-of-range rows / columns load from clamped (valid) addresses and only their STORES are predicated: no control
//     flow between the loads; the activation is a template argument (one switch per wavefront).
// Needs 4-element alignment of N and every stride (else the general form).
// (r5) `rpre` != nullptr: the residual float4s of the whole wavefront tile were fetched at the START of the kernel
// (gemm_prefetch_residual below) and the epilogue only consumes them — no load sits between the last MFMA and the first store.
template <int BM, int BN, int TM, int TN, int MI, int NI, bool LDSCOLS = false, int RS = 16, int CS = 16>
__device__ __forceinline__ void gemm_epilogue(const dzn_gemm_desc& d, f32x4 (&acc)[MI][NI], int tm, int tn,
                                              int wm, int wn, int lr, int lq, int64_t cz, int64_t bz, int z0 = 0,
                                              const float* row_inv = nullptr, const float* col_scale = nullptr,
                                              float* lds_cols = nullptr, const f32x4 (*rpre)[NI] = nullptr) {
  const bool vec = (((int64_t)d.N | d.ldc | d.ldws | cz | bz) & 3) == 0 && d.N >= 4;
  if (!vec) {
    gemm_epilogue_general<BM, BN, TM, TN, MI, NI, RS, CS>(d, acc, tm, tn, wm, wn, lr, lq, cz, bz, z0, row_inv, col_scale);
    return;
  }
  const bool ring = d.R && !rpre;                       // residual through the two-deep ring (fetched here)
  // column blocks per batch: 4 keeps acc + two residual batches + the hoisted column-vector reads inside 256 registers
  constexpr int JCMAX = MI * NI > 16 ? 4 : 8;          // wide tiles: 128 accumulator registers leave room for 2 x 4 float4s
  constexpr int JC = epi_chunk(NI, JCMAX);
  static_assert(NI % JC == 0, "column chunks tile the wavefront tile");
  constexpr int NJC = NI / JC, NB = MI * NJC;            // batch b = (row block b / NJC, column chunk b % NJC)
  constexpr bool REGCOLS = !LDSCOLS && NI <= 4;          // register column vectors need 12 NI registers
  const float* __restrict__ bias = d.bias ? d.bias + bz : nullptr;
  const int ncol0 = tn * BN + wn * TN + lq * 4;          // column of block j: ncol0 + CS j
  // recomputed at every use rather than kept in 2 NI registers: N % 4 == 0, so a float4 is inside or outside as a whole
  auto nok_ = [&](const int j) { return ncol0 + j * CS < d.N; };
  auto nc_ = [&](const int j) {                          // ... clamped for the loads
    const int n0 = ncol0 + j * CS;
    return n0 < d.N ? n0 : d.N - 4;
  };
  int64_t crow[MI];
  bool mok[MI];
  int mcl[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = tm * BM + wm * TM + i * RS + lr;
    mok[i] = m < d.M;
    mcl[i] = mok[i] ? m : d.M - 1;
    crow[i] = cz + (d.c_rowoff ? (int64_t)d.c_rowoff[mcl[i]] : (int64_t)mcl[i] * d.ldc);
  }
  // ---- residual ring: batch 0 goes out first, the column vectors and row statistics travel beside it ----
  float4 r4[2][JC];
  auto issue_r = [&](const int b, const int slot) {
    const int i = b / NJC, jc = b % NJC;
#pragma unroll
    for (int jj = 0; jj < JC; ++jj) r4[slot][jj] = *reinterpret_cast<const float4*>(d.R + crow[i] + nc_(jc * JC + jj));
  };
  if (ring) issue_r(0, 0);
  float ln_mu[MI], ln_rs[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    ln_mu[i] = 0.f;
    ln_rs[i] = 1.f;
  }
  if (d.ln_stats) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const float2 st = *reinterpret_cast<const float2*>(d.ln_stats + 2 * (int64_t)mcl[i]);
      ln_mu[i] = d.ln_centered ? 0.f : st.x;   // centered (DZN_PREC_F16): the kernel already multiplied (x - mean)
      ln_rs[i] = st.y;
    }
  }
  float4 cs4[REGCOLS ? NI : 1], lc4[REGCOLS ? NI : 1], b4[REGCOLS ? NI : 1];
  if constexpr (REGCOLS) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      cs4[j] = make_float4(1.f, 1.f, 1.f, 1.f);
      lc4[j] = b4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (col_scale) {
#pragma unroll
      for (int j = 0; j < NI; ++j) cs4[j] = *reinterpret_cast<const float4*>(col_scale + nc_(j));
    }
    if (d.ln_stats) {
#pragma unroll
      for (int j = 0; j < NI; ++j) lc4[j] = *reinterpret_cast<const float4*>(d.ln_colsum + nc_(j));
    }
    if (bias) {
#pragma unroll
      for (int j = 0; j < NI; ++j) b4[j] = *reinterpret_cast<const float4*>(bias + nc_(j));
    }
  } else if constexpr (LDSCOLS) {
    // lane l < TN / 4 owns columns 4 l .. 4 l + 3 of the wavefront tile
    const int lane = lq * RS + lr;
    if (lane < TN / 4) {
      int n = tn * BN + wn * TN + 4 * lane;
      n = n < d.N ? n : d.N - 4;
      float4 c = make_float4(1.f, 1.f, 1.f, 1.f), l = make_float4(0.f, 0.f, 0.f, 0.f), bb = l;
      if (col_scale) c = *reinterpret_cast<const float4*>(col_scale + n);
      if (d.ln_stats) l = *reinterpret_cast<const float4*>(d.ln_colsum + n);
      if (bias) bb = *reinterpret_cast<const float4*>(bias + n);
      *reinterpret_cast<float4*>(lds_cols + 4 * lane) = c;
      *reinterpret_cast<float4*>(lds_cols + TN + 4 * lane) = l;
      *reinterpret_cast<float4*>(lds_cols + 2 * TN + 4 * lane) = bb;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // wave-private: order the ds_writes before the ds_reads
  }
  auto cols = [&](const int j, float4& c, float4& l, float4& bb) {
    if constexpr (REGCOLS) {
      c = cs4[j]; l = lc4[j]; bb = b4[j];
    } else if constexpr (LDSCOLS) {
      const float* p = lds_cols + j * CS + lq * 4;
      c = *reinterpret_cast<const float4*>(p);
      l = *reinterpret_cast<const float4*>(p + TN);
      bb = *reinterpret_cast<const float4*>(p + 2 * TN);
    } else {    // wide tile without scratch: straight from global (L2 hits), still no control flow
      c = col_scale ? *reinterpret_cast<const float4*>(col_scale + nc_(j)) : make_float4(1.f, 1.f, 1.f, 1.f);
      l = d.ln_stats ? *reinterpret_cast<const float4*>(d.ln_colsum + nc_(j)) : make_float4(0.f, 0.f, 0.f, 0.f);
      bb = bias ? *reinterpret_cast<const float4*>(bias + nc_(j)) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  float amax_row[MI], st_s[MI], st_q[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) amax_row[i] = st_s[i] = st_q[i] = 0.f;

  auto run = [&](auto actc) {
    constexpr int ACT = decltype(actc)::value;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int i = b / NJC, jc = b % NJC;
      if (ring && b + 1 < NB) issue_r(b + 1, (b + 1) & 1);
      const float acc_scale = (col_scale && row_inv) ? row_inv[i] : 1.f;
#pragma unroll
      for (int jj = 0; jj < JC; ++jj) {
        const int j = jc * JC + jj;
        float4 c4, l4, bb4;
        cols(j, c4, l4, bb4);
        f32x4 v = acc[i][j];
        if (col_scale) {   // fp16 two-term operands were scaled by exact powers of two: undo (exact)
          v[0] *= acc_scale * c4.x; v[1] *= acc_scale * c4.y; v[2] *= acc_scale * c4.z; v[3] *= acc_scale * c4.w;
        }
        if (d.ln_stats) {
          v[0] = ln_rs[i] * (v[0] - ln_mu[i] * l4.x); v[1] = ln_rs[i] * (v[1] - ln_mu[i] * l4.y);
          v[2] = ln_rs[i] * (v[2] - ln_mu[i] * l4.z); v[3] = ln_rs[i] * (v[3] - ln_mu[i] * l4.w);
        }
        if (bias) { v[0] += bb4.x; v[1] += bb4.y; v[2] += bb4.z; v[3] += bb4.w; }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act_c<ACT>(v[e]) * d.alpha;
        if (rpre) {
          const f32x4 r = rpre[i][j];
          v[0] += r[0]; v[1] += r[1]; v[2] += r[2]; v[3] += r[3];
        } else if (d.R) {
          const float4 r = r4[b & 1][jj];
          v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
        }
        if (d.post_relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        acc[i][j] = v;
        if (nok_(j)) {      // statistics / |max| over the columns that exist
          amax_row[i] = fmaxf(fmaxf(amax_row[i], fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
          st_s[i] += (v[0] + v[1]) + (v[2] + v[3]);
          st_q[i] = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], fmaf(v[3], v[3], st_q[i]))));
        }
        // keep the ALU work of one column block together (memory instructions may still move across): interleaving the
        // erf chains of 16 elements costs ~100 live registers and spills
        __builtin_amdgcn_sched_barrier(0x00E0 | 0x0200);
      }
      // (r6) K / V head slots leave as pre-split fp16 planes with their own per-(row, slot) power-of-two scale (dzn_ops.h:
      // kv_planes): the slot's 64 columns of a row sit in 4 column blocks x the 4 lanes lr, lr + 16, lr + 32, lr + 48
      if constexpr (JC % 4 == 0 && RS == 16 && CS == 16) {
        if (d.kv_planes) {
#pragma unroll
          for (int g4 = 0; g4 < JC / 4; ++g4) {
            const int jg = jc * JC + g4 * 4;
            const int nslot = tn * BN + wn * TN + jg * CS;                  // wave-uniform first column of the slot
            if (nslot >= d.kv_col0 && nslot < d.N) {
              float am = 0.f;
#pragma unroll
              for (int q = 0; q < 4; ++q)
                am = fmaxf(fmaxf(am, fmaxf(fabsf(acc[i][jg + q][0]), fabsf(acc[i][jg + q][1]))),
                           fmaxf(fabsf(acc[i][jg + q][2]), fabsf(acc[i][jg + q][3])));
              am = fmaxf(am, __shfl_xor(am, 16, 64));
              am = fmaxf(am, __shfl_xor(am, 32, 64));
              float ks, kinv;
              h2_scale(am, ks, kinv);
              uint16_t* p0 = reinterpret_cast<uint16_t*>(d.kv_planes) + (int64_t)mcl[i] * d.kv_ld + (nslot - d.kv_col0) + lq * 4;
              uint16_t* p1 = p0 + d.kv_plane_stride;
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                f32x2 x0, x1;
                x0[0] = acc[i][jg + q][0] * ks; x0[1] = acc[i][jg + q][1] * ks;
                x1[0] = acc[i][jg + q][2] * ks; x1[1] = acc[i][jg + q][3] * ks;
                const f16x2 h0 = __builtin_convertvector(x0, f16x2), h1 = __builtin_convertvector(x1, f16x2);
                const f32x2 f0 = __builtin_convertvector(h0, f32x2), f1 = __builtin_convertvector(h1, f32x2);
                f32x2 r0, r1;
                r0[0] = x0[0] - f0[0]; r0[1] = x0[1] - f0[1]; r1[0] = x1[0] - f1[0]; r1[1] = x1[1] - f1[1];   // exact in fp32
                const f16x2 l0 = __builtin_convertvector(r0, f16x2), l1 = __builtin_convertvector(r1, f16x2);
                if (mok[i]) {
                  *reinterpret_cast<uint2*>(p0 + q * CS) = make_uint2(__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1));
                  *reinterpret_cast<uint2*>(p1 + q * CS) = make_uint2(__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1));
                }
              }
              if (lq == 0 && mok[i]) d.kv_scale[(int64_t)mcl[i] * (d.kv_ld >> 6) + ((nslot - d.kv_col0) >> 6)] = kinv;
            }
          }
        }
      }
#pragma unroll
      for (int jj = 0; jj < JC; ++jj) {
        const int j = jc * JC + jj;
        bool plain = true;
        if constexpr (JC % 4 == 0 && RS == 16 && CS == 16)
          plain = !(d.kv_planes && tn * BN + wn * TN + (j & ~3) * CS >= d.kv_col0);      // the slot went out as planes
        if (plain && mok[i] && nok_(j))
          *reinterpret_cast<float4*>(d.C + crow[i] + nc_(j)) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      }
    }
  };
  switch (d.act) {
    case DZN_ACT_GELU: run(std::integral_constant<int, DZN_ACT_GELU>{}); break;
    case DZN_ACT_SWISH: run(std::integral_constant<int, DZN_ACT_SWISH>{}); break;
    case DZN_ACT_RELU: run(std::integral_constant<int, DZN_ACT_RELU>{}); break;
    default: run(std::integral_constant<int, DZN_ACT_NONE>{}); break;
  }
  // ---- layer-weighted sum: WS (+)= ws_w * v, second pass, same two-deep ring ----
  if (d.WS) {
    float4 w4[2][JC];
    auto issue_w = [&](const int b, const int slot) {
      const int i = b / NJC, jc = b % NJC;
      const float* wrow = d.WS + (int64_t)mcl[i] * d.ldws;
#pragma unroll
      for (int jj = 0; jj < JC; ++jj) w4[slot][jj] = *reinterpret_cast<const float4*>(wrow + nc_(jc * JC + jj));
    };
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int jj = 0; jj < JC; ++jj) w4[s][jj] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!d.ws_init) issue_w(0, 0);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int i = b / NJC, jc = b % NJC;
      if (!d.ws_init && b + 1 < NB) issue_w(b + 1, (b + 1) & 1);
      float* wrow = d.WS + (int64_t)mcl[i] * d.ldws;
#pragma unroll
      for (int jj = 0; jj < JC; ++jj) {
        const int j = jc * JC + jj;
        const float4 w = w4[b & 1][jj];
        if (mok[i] && nok_(j))
          *reinterpret_cast<float4*>(wrow + nc_(j)) =
              make_float4(w.x + d.ws_w * acc[i][j][0], w.y + d.ws_w * acc[i][j][1], w.z + d.ws_w * acc[i][j][2],
                          w.w + d.ws_w * acc[i][j][3]);
      }
    }
  }
  // ---- row statistics for a following folded LayerNorm, |max| trackers ----
  float amax = 0.f, amax_hi = 0.f;
  const int wrow0 = tm * BM + wm * TM;
  const bool two_unit = d.c_amax && d.amax_unit >= TM;
  const int unit0 = two_unit ? (wrow0 < d.M ? wrow0 : d.M - 1) / d.amax_unit : 0;
  const int boundary = two_unit ? (unit0 + 1) * d.amax_unit : 0;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = tm * BM + wm * TM + i * RS + lr;
    if (!mok[i]) continue;
    if (d.stat_partial) {
      // every wavefront leaves (sum, sum of squares) of its TN columns; stats_finalize_kernel adds the tilesN * (BN / TN)
      // partials of a row in a fixed order (deterministic).  The row lives in lanes lr, lr + 16, lr + 32, lr + 48.
      float s1 = st_s[i], q1 = st_q[i];
      if constexpr (RS == 16) {
        s1 += __shfl_xor(s1, 16, 64);
        q1 += __shfl_xor(q1, 16, 64);
      }
      s1 += __shfl_xor(s1, 32, 64);
      q1 += __shfl_xor(q1, 32, 64);
      if (lq == 0) {
        const int P = ((d.N + BN - 1) / BN) * (BN / TN);
        reinterpret_cast<float2*>(d.stat_partial)[(int64_t)m * P + tn * (BN / TN) + wn] = make_float2(s1, q1);
      }
    }
    if (d.c_amax) {
      if (two_unit) {
        if (m < boundary) amax = fmaxf(amax, amax_row[i]);
        else amax_hi = fmaxf(amax_hi, amax_row[i]);
      } else if (d.amax_unit > 0) {
        float r = amax_row[i];
        if constexpr (RS == 16) r = fmaxf(r, __shfl_xor(r, 16, 64));
        r = fmaxf(r, __shfl_xor(r, 32, 64));
        if (lq == 0) track_amax_lane(d.c_amax + m / d.amax_unit, r);
      } else {
        amax = fmaxf(amax, amax_row[i]);
      }
    }
  }
  if (two_unit) {
    track_amax(d.c_amax + unit0, amax);
    if (wrow0 + TM > boundary && boundary < d.M) track_amax(d.c_amax + unit0 + 1, amax_hi);   // wave-uniform
  } else if (d.c_amax && d.amax_unit <= 0) {
    track_amax(d.c_amax + z0, amax);
  }
}

// (r5) residual prefetch for short-K launches (out_proj / FFN-output: K <= 512, N = 1024), whose time is their epilogue: the
// float4s the epilogue will add are requested BEFORE the first operand tile, so they travel beside the prologue's LDS-DMA and
// the epilogue is stores only.  Same addresses, same clamping as gemm_epilogue (valid: N % 4 == 0 and aligned strides — the
// caller checks `gemm_epilogue_vec`); costs MI x NI x 4 registers for the life of the K loop.
__device__ __forceinline__ bool gemm_epilogue_vec(const dzn_gemm_desc& d, int64_t cz, int64_t bz) {
  return (((int64_t)d.N | d.ldc | d.ldws | cz | bz) & 3) == 0 && d.N >= 4;
}
template <int BM, int BN, int TM, int TN, int MI, int NI, int RS = 16, int CS = 16>
__device__ __forceinline__ void gemm_prefetch_residual(const dzn_gemm_desc& d, f32x4 (&rpre)[MI][NI], int tm, int tn, int wm, int wn,
                                                       int lr, int lq, int64_t cz) {
  const int ncol0 = tn * BN + wn * TN + lq * 4;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    int m = tm * BM + wm * TM + i * RS + lr;
    m = m < d.M ? m : d.M - 1;
    const int64_t crow = cz + (d.c_rowoff ? (int64_t)d.c_rowoff[m] : (int64_t)m * d.ldc);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      int n0 = ncol0 + j * CS;
      n0 = n0 < d.N ? n0 : d.N - 4;
      rpre[i][j] = *reinterpret_cast<const f32x4*>(d.R + crow + n0);
    }
  }
}

}  // namespace
