// gemm_epilogue.h — the fused epilogue shared by the fp32 contraction kernels (gemm.hip,
// gemm_split*.hip): bias, activation, alpha, residual, post-ReLU, store, layer-weighted-sum
// accumulate.  The accumulator block of lane (lr, lq) is C[m = ..+lr][n0 .. n0+3] (operands are
// fed to the MFMA swapped, so a lane owns 4 consecutive columns -> float4 traffic).
#pragma once
#include "common.h"

namespace {

template <int BM, int BN, int TM, int TN, int MI, int NI>
__device__ __forceinline__ void gemm_epilogue(const dzn_gemm_desc& d, f32x4 (&acc)[MI][NI], int tm, int tn,
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
    const int m = tm * BM + wm * TM + i * 16 + lr;
    if (m >= d.M) continue;
    const int64_t crow = cz + (d.c_rowoff ? (int64_t)d.c_rowoff[m] : (int64_t)m * d.ldc);
    // LayerNorm folded into the weights: finish it with the row statistics (dzn_ops.h)
    const float acc_scale = row_inv ? row_inv[i] : 1.f;
    float amax_row = 0.f;
    float st_s = 0.f, st_q = 0.f;   // partial (sum, sum of squares) of this row over the wave tile's columns (d.stat_partial)
    float ln_mu = 0.f, ln_rs = 1.f;
    if (d.ln_stats) {
      const float2 st = *reinterpret_cast<const float2*>(d.ln_stats + 2 * (int64_t)m);
      ln_mu = st.x;
      ln_rs = st.y;
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n0 = tn * BN + wn * TN + j * 16 + lq * 4;
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
      float s1 = st_s + __shfl_xor(st_s, 16, 64), q1 = st_q + __shfl_xor(st_q, 16, 64);
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
        float r = fmaxf(amax_row, __shfl_xor(amax_row, 16, 64));
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

}  // namespace
