// attention.hip — fused multi-head self-attention for the WavLM encoder (with the gated
// relative-position bias) and the Conformer head (no bias), on fp32 MFMA (gfx950).
//
// Reference arithmetic (W2V/components.py:453-486 SelfAttention.forward, :690-725
// WavLMSelfAttention.forward; diarizen/models/module/conformer.py:47-71):
//     S = (q * d^-0.5) k^T + gate[q] * P[head][key - query]   ;  A = softmax(S) ;  o = A v
// The reference materialises P as [B*H, L, L] and S as [B, h, L, L]; here the bias is a
// per-head Toeplitz table of 2L-1 floats held in LDS, S never leaves registers
// (flash-style online softmax) and only kept heads are computed.
//
// Work split: grid (ceil(L/64), kept heads, B); 4 wavefronts x 16 query rows.  The
// score block is computed TRANSPOSED (S^T = K Q^T) so that every lane owns one query
// column: row max / sum are in-lane reductions plus two cross-lane-group shuffles, and
// the C-layout of S^T is exactly the A-operand layout of P for the P.V MFMA (k index
// = lane group * 4 + step on both operands), so P never round-trips through LDS.
// Output columns of P.V are permuted (lane lr owns dd = 4 lr + block) so each V operand
// fetch is one ds_read_b128 and the final store is a contiguous float4 per lane.
#include "common.h"

namespace {

template <bool BIAS, typename TO>
__global__ __launch_bounds__(256) void attn_kernel(const float* __restrict__ qkv,
                                                   TO* __restrict__ out,
                                                   const float* __restrict__ gate,
                                                   const float* __restrict__ table,
                                                   const int32_t* __restrict__ head_idx, int B,
                                                   int L, int h, int Htot, int ldqkv, int ldo,
                                                   float scale) {
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  float* sK = smem_f;            // [64 keys][64 d], 16-B slot XOR (key & 15)
  float* sV = smem_f + 64 * 64;  // [64 keys][64 d], linear
  float* sT = smem_f + 2 * 64 * 64;  // [2L-1] bias table of this head

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

  // ---- Q fragment (B operand of S^T = K Q^T): lane owns query row q_row ----
  const int q_row = qt * 64 + wave * 16 + lr;
  const bool q_ok = q_row < L;
  f32x4 qf[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) {
    if (q_ok) {
      const float4 v =
          *reinterpret_cast<const float4*>(Qp + (rowbase + q_row) * ldqkv + db * 16 + lq * 4);
      qf[db] = (f32x4){v.x * scale, v.y * scale, v.z * scale, v.w * scale};
    } else {
      qf[db] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
  float g = 0.f;
  if constexpr (BIAS) {
    if (q_ok) g = gate[(rowbase + q_row) * Htot + H];
  }

  float m_run = -INFINITY, l_run = 0.f;
  f32x4 O[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) O[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nkt = (L + 63) / 64;
  float4 rk[4], rv[4];
  auto prefetch = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i;
      const int row = idx >> 4, slot = idx & 15;
      const int key = kt * 64 + row;
      if (key < L) {
        rk[i] = *reinterpret_cast<const float4*>(Kp + (rowbase + key) * ldqkv + slot * 4);
        rv[i] = *reinterpret_cast<const float4*>(Vp + (rowbase + key) * ldqkv + slot * 4);
      } else {
        rk[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        rv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };

  prefetch(0);
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();  // previous tile fully consumed
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i;
      const int row = idx >> 4, slot = idx & 15;
      *reinterpret_cast<float4*>(sK + row * 64 + ((slot ^ (row & 15)) << 2)) = rk[i];
      *reinterpret_cast<float4*>(sV + row * 64 + (slot << 2)) = rv[i];
    }
    __syncthreads();
    if (kt + 1 < nkt) prefetch(kt + 1);

    // ---- S^T = K Q^T : 4 key blocks x (4 d blocks x 4 steps) ----
    f32x4 s[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      s[kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const f32x4 kf = *reinterpret_cast<const f32x4*>(
            sK + (kb * 16 + lr) * 64 + (((db * 4 + lq) ^ lr) << 2));
#pragma unroll
        for (int st = 0; st < 4; ++st)
          s[kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[st], qf[db][st], s[kb], 0, 0, 0);
      }
    }

    // ---- bias, mask, online softmax (lane owns query q_row; keys kb*16 + lq*4 + rg) ----
    const int key0 = kt * 64;
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int key = key0 + kb * 16 + lq * 4 + rg;
        float v = s[kb][rg];
        if constexpr (BIAS) {
          if (q_ok && key < L) v += g * sT[key - q_row + L - 1];
        }
        if (key >= L) v = -INFINITY;
        s[kb][rg] = v;
        mx = fmaxf(mx, v);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = expf(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const float p = expf(s[kb][rg] - m_new);
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

    // ---- O += P V ----
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        const f32x4 vf =
            *reinterpret_cast<const f32x4*>(sV + (kb * 16 + lq * 4 + st) * 64 + (lr << 2));
#pragma unroll
        for (int dblk = 0; dblk < 4; ++dblk)
          O[dblk] = __builtin_amdgcn_mfma_f32_16x16x4f32(s[kb][st], vf[dblk], O[dblk], 0, 0, 0);
      }
    }
  }

  // ---- normalise and store: lane holds O[q = lq*4+rg][dd = lr*4 + dblk] ----
  float l_tot = l_run + __shfl_xor(l_run, 16, 64);
  l_tot += __shfl_xor(l_tot, 32, 64);
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) {
    const float lt = __shfl(l_tot, lq * 4 + rg, 64);
    const int q = qt * 64 + wave * 16 + lq * 4 + rg;
    if (q < L) {
      const float inv = 1.0f / lt;
      TO* op = out + (rowbase + q) * ldo + j * 64 + lr * 4;
#pragma unroll
      for (int dblk = 0; dblk < 4; ++dblk) st_act(op, dblk, O[dblk][rg] * inv);
    }
  }
}

// gate_a_1[row, H] of the gated relative position bias, W2V/components.py:702-710:
//   t = Linear(64->8)(y[row, H*64:(H+1)*64]) ; (a, b) = sigmoid(sum t[0:4]), sigmoid(sum t[4:8])
//   gate = a * (b * const[H] - 1) + 2
// One wavefront per row; lane (H = lane/4, sub = lane%4) computes outputs 2*sub, 2*sub+1.
template <typename TI>
__global__ __launch_bounds__(256) void gate_kernel(const TI* __restrict__ y, int64_t ldy,
                                                   const float* __restrict__ Wg,  // [8,64]
                                                   const float* __restrict__ bg,  // [8]
                                                   const float* __restrict__ cst,  // [Htot]
                                                   float* __restrict__ gate, int64_t rows,
                                                   int Htot) {
  __shared__ float sW[8 * 64 + 8];
  for (int i = threadIdx.x; i < 8 * 64; i += 256) sW[i] = Wg[i];
  if (threadIdx.x < 8) sW[512 + threadIdx.x] = bg[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const int sub = lane & 3;
  for (int H0 = 0; H0 < Htot; H0 += 16) {
    const int H = H0 + (lane >> 2);
    float t0 = 0.f, t1 = 0.f;
    if (H < Htot) {
      const TI* yp = y + row * ldy + H * 64;
      const float* w0 = sW + (2 * sub) * 64;
      const float* w1 = w0 + 64;
#pragma unroll 8
      for (int d = 0; d < 64; ++d) {
        const float v = ld_act(yp, d);
        t0 = fmaf(v, w0[d], t0);
        t1 = fmaf(v, w1[d], t1);
      }
      t0 += sW[512 + 2 * sub];
      t1 += sW[512 + 2 * sub + 1];
    }
    // outputs 0..3 live in sub 0,1 ; outputs 4..7 in sub 2,3  (sum in the reference's order)
    float pair = t0 + t1;
    float other = __shfl_xor(pair, 1, 64);
    float sum4 = (sub & 1) ? (other + pair) : (pair + other);
    // lane sub==0 -> sum of t[0..3]; sub==2 -> sum of t[4..7]
    const float sa = __shfl(sum4, (lane & ~3), 64);
    const float sb = __shfl(sum4, (lane & ~3) + 2, 64);
    if (H < Htot && sub == 0) {
      const float ga = 1.0f / (1.0f + expf(-sa));
      const float gb = 1.0f / (1.0f + expf(-sb));
      gate[row * Htot + H] = ga * (gb * cst[H] - 1.0f) + 2.0f;
    }
  }
}

// Pre-norm layers (W2V/components.py:920-925 with layer_norm_first): ONE pass over the residual row x that
//   * computes the LayerNorm statistics (mean, rstd) consumed by the q/k/v contraction, whose weights carry
//     gamma / beta (dzn_gemm_desc.ln_stats) — the normalised copy y is never written;
//   * applies that LayerNorm in registers and evaluates the gate of the relative position bias on it
//     (components.py:702-710): t = Linear(64->8)(y_h); a = sigmoid(t[0:4].sum()), b = sigmoid(t[4:8].sum()).
//     The two 4-sums are linear in y_h, so the weights are pre-summed into wa = sum Wg[0:4], wb = sum Wg[4:8].
// One wavefront per row; element lane + 64 i of the row is channel `lane` of head i, staged through a padded
// LDS row so that lane (H = lane/4, sub = lane%4) reads 16 consecutive channels of head H without bank conflicts.
template <int MAXI>
__global__ __launch_bounds__(256) void gate_stats_kernel(const float* __restrict__ x, int64_t ldx,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ Wg, const float* __restrict__ bg,
                                                         const float* __restrict__ cst, float* __restrict__ gate,
                                                         float* __restrict__ stats, int64_t rows, int Htot, float eps) {
  constexpr int ROW = MAXI * 64 + MAXI * 4;          // one pad float per 16 channels
  __shared__ float sy[4][ROW];
  __shared__ float sw[2 * 64 + 2];                   // wa[64], wb[64], ba, bb
  if (threadIdx.x < 128) {
    const int o0 = threadIdx.x < 64 ? 0 : 4, dch = threadIdx.x & 63;
    sw[threadIdx.x] = (Wg[(o0 + 0) * 64 + dch] + Wg[(o0 + 1) * 64 + dch]) + (Wg[(o0 + 2) * 64 + dch] + Wg[(o0 + 3) * 64 + dch]);
  } else if (threadIdx.x < 130) {
    const int o0 = threadIdx.x == 128 ? 0 : 4;
    sw[threadIdx.x] = (bg[o0] + bg[o0 + 1]) + (bg[o0 + 2] + bg[o0 + 3]);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane & 3;
  float wa[16], wb[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    wa[j] = sw[sub * 16 + j];
    wb[j] = sw[64 + sub * 16 + j];
  }
  const float ba = sw[128], bb = sw[129];
  const int C = Htot * 64;
  // gamma / beta of this lane's channels stay in registers across the rows of the wave
  float gm[MAXI], bt[MAXI];
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    gm[i] = i < Htot ? gamma[lane + 64 * i] : 0.f;
    bt[i] = i < Htot ? beta[lane + 64 * i] : 0.f;
  }
  float* sr = sy[wave];
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    const float* xp = x + row * ldx;
    float v[MAXI];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
      v[i] = i < Htot ? xp[lane + 64 * i] : 0.f;
      sum += v[i];
    }
    const float mean = wave_sum(sum) / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
      const float dv = i < Htot ? v[i] - mean : 0.f;
      sq += dv * dv;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)C + eps);
    if (lane == 0) *reinterpret_cast<float2*>(stats + 2 * row) = make_float2(mean, rstd);
    __builtin_amdgcn_wave_barrier();      // the previous row's LDS reads are done (in-order LDS queue of the wave)
#pragma unroll
    for (int i = 0; i < MAXI; ++i)
      if (i < Htot) {
        const int e = lane + 64 * i;
        sr[e + (e >> 4)] = (v[i] - mean) * rstd * gm[i] + bt[i];
      }
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): LDS writes of this wave have landed
    __builtin_amdgcn_wave_barrier();      // and the compiler keeps the cross-lane reads below them
    for (int H0 = 0; H0 < Htot; H0 += 16) {
      const int H = H0 + (lane >> 2);
      float ta = 0.f, tb = 0.f;
      if (H < Htot) {
        const int e0 = H * 64 + sub * 16;
        const float* yp = sr + e0 + (e0 >> 4);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          ta = fmaf(yp[j], wa[j], ta);
          tb = fmaf(yp[j], wb[j], tb);
        }
      }
      ta = dpp_add<0xB1, 0xF>(ta);   // quad_perm [1,0,3,2]
      ta = dpp_add<0x4E, 0xF>(ta);   // quad_perm [2,3,0,1] -> every lane of the quad holds the head's sum
      tb = dpp_add<0xB1, 0xF>(tb);
      tb = dpp_add<0x4E, 0xF>(tb);
      if (H < Htot && sub == 0) {
        const float ga = 1.0f / (1.0f + expf(-(ta + ba)));
        const float gb = 1.0f / (1.0f + expf(-(tb + bb)));
        gate[row * Htot + H] = ga * (gb * cst[H] - 1.0f) + 2.0f;
      }
    }
  }
}

}  // namespace

int launch_gate_stats(const float* x, int64_t ldx, const float* gamma, const float* beta, const float* Wg,
                      const float* bg, const float* cst, float* gate, float* stats, int64_t rows, int Htot, float eps,
                      hipStream_t s) {
  if (rows <= 0) return DZN_OK;
  if (Htot < 1 || Htot > 16) return DZN_E_INVALID;
  int pid = prof_enabled() ? prof_begin(s, "gate_ln_stats", 0.0, (double)rows * Htot * 64 * 4.0) : -1;
  int64_t grid = cdiv64(rows, 4);
  grid = grid > 256 * 8 ? 256 * 8 : grid;   // waves walk rows grid-stride: per-wave constants are loaded once
  hipLaunchKernelGGL(gate_stats_kernel<16>, dim3((unsigned)grid), dim3(256), 0, s, x, ldx, gamma, beta, Wg,
                     bg, cst, gate, stats, rows, Htot, eps);
  prof_end(pid, s);
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}

extern "C" int dzn_op_gate_stats(const float* x, int64_t ldx, const float* gamma, const float* beta, const float* Wg,
                                 const float* bg, const float* cst, float* gate, float* stats, int64_t rows,
                                 int32_t Htot, float eps, void* stream) {
  if (!x || !gamma || !beta || !Wg || !bg || !cst || !gate || !stats) return DZN_E_INVALID;
  return launch_gate_stats(x, ldx, gamma, beta, Wg, bg, cst, gate, stats, rows, Htot, eps,
                           reinterpret_cast<hipStream_t>(stream));
}

int launch_attention_t(const float* qkv, void* out, int out_bf16, const float* gate, const float* table,
                       const int32_t* head_idx, int B, int L, int h, int Htot, int ldqkv, int ldo,
                       float scale, hipStream_t s) {
  if (h <= 0 || B <= 0 || L <= 0) return DZN_OK;
  const bool bias = gate && table && head_idx;
  const size_t lds = (2 * 64 * 64 + (bias ? (2 * L - 1) : 0)) * sizeof(float);
  if (lds > 160 * 1024) return DZN_E_INVALID;
  dim3 grid((L + 63) / 64, h, B);
  // algorithmic flops: QK^T and PV, 2*L*L*64 each per (batch, head)
  const int pid = prof_begin(s, bias ? "attention_relpos_f32" : "attention_f32",
                             4.0 * B * h * (double)L * L * 64.0,
                             (double)B * L * h * 64.0 * 4.0 * 4.0 + (gate ? (double)B * L * Htot * 4.0 : 0.0));
#define DZN_ATT(BIASV, T)                                                                          \
  hipLaunchKernelGGL((attn_kernel<BIASV, T>), grid, dim3(256), lds, s, qkv, static_cast<T*>(out), gate, \
                     table, head_idx, B, L, h, Htot, ldqkv, ldo, scale)
  if (bias) {
    if (out_bf16) DZN_ATT(true, u16); else DZN_ATT(true, float);
  } else {
    if (out_bf16) DZN_ATT(false, u16); else DZN_ATT(false, float);
  }
#undef DZN_ATT
  prof_end(pid, s);
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}

int launch_attention(const float* qkv, float* out, const float* gate, const float* table,
                     const int32_t* head_idx, int B, int L, int h, int Htot, int ldqkv, int ldo,
                     float scale, int precision, hipStream_t s) {
  if (prec_is_split(precision))
    return launch_attention_split(qkv, out, gate, table, head_idx, B, L, h, Htot, ldqkv, ldo, scale, s);
  return launch_attention_t(qkv, out, 0, gate, table, head_idx, B, L, h, Htot, ldqkv, ldo, scale, s);
}

int launch_gate_t(const void* y, int y_bf16, int64_t ldy, const float* Wg, const float* bg,
                  const float* cst, float* gate, int64_t rows, int Htot, hipStream_t s) {
  ProfScope prof_scope_(s, "gate", 0.0, (double)rows * Htot * (64.0 + 1.0) * 4.0);
  if (rows <= 0) return DZN_OK;
  if (y_bf16)
    hipLaunchKernelGGL(gate_kernel<u16>, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, s,
                       static_cast<const u16*>(y), ldy, Wg, bg, cst, gate, rows, Htot);
  else
    hipLaunchKernelGGL(gate_kernel<float>, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, s,
                       static_cast<const float*>(y), ldy, Wg, bg, cst, gate, rows, Htot);
  return hipGetLastError() == hipSuccess ? DZN_OK : DZN_E_HIP;
}

int launch_gate(const float* y, int64_t ldy, const float* Wg, const float* bg, const float* cst,
                float* gate, int64_t rows, int Htot, hipStream_t s) {
  return launch_gate_t(y, 0, ldy, Wg, bg, cst, gate, rows, Htot, s);
}

extern "C" int dzn_op_attention(const float* qkv, float* out, const float* gate,
                                const float* table, const int32_t* head_idx, int32_t B, int32_t L,
                                int32_t h, int32_t Htot, int32_t ldqkv, int32_t ldo, float scale,
                                int32_t precision, void* stream) {
  if (!qkv || !out) return DZN_E_INVALID;
  return launch_attention(qkv, out, gate, table, head_idx, B, L, h, Htot, ldqkv, ldo, scale,
                          precision, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int dzn_op_gate(const float* y, int64_t ldy, const float* Wg, const float* bg,
                           const float* cst, float* gate, int64_t rows, int32_t Htot,
                           void* stream) {
  return launch_gate(y, ldy, Wg, bg, cst, gate, rows, Htot, reinterpret_cast<hipStream_t>(stream));
}
