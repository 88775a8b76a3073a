// split.h — exact 3-way bf16 split of fp32 values (DZN_PREC_F32_SPLIT kernels).
//
// x = hi + mid + lo with hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid), each
// conversion round-to-nearest-even; both subtractions are exact in fp32 and the last remainder
// has at most 8 significant bits, so the three terms reproduce x bit for bit (inf / nan excepted).
#pragma once
#include "common.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// exact remainder a - b as a scalar v_sub_f32: inline asm keeps the SLP vectoriser from fusing
// two of them into one v_pk_add_f32, which is several issue slots dearer next to MFMAs (measured:
// scripts/ubench/mfma_rate_bf16.hip, +6 % MFMA rate with the split running beside it)
__device__ __forceinline__ float exact_sub(float a, float b) {
  float r;
  asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// 8 fp32 values (two float4) -> three bf16x8 MFMA fragments.  Per pair: v_cvt_pk_bf16_f32,
// v_lshlrev + v_and (the two hi values back as fp32), 2 v_sub_f32, v_cvt_pk, v_lshlrev + v_and,
// 2 v_sub_f32, v_cvt_pk.
__device__ __forceinline__ void split8(const f32x4& u, const f32x4& v, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
  u32x4 H, M, L;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    f32x2 x;
    x[0] = p < 2 ? u[2 * p] : v[2 * p - 4];
    x[1] = p < 2 ? u[2 * p + 1] : v[2 * p - 3];
    const unsigned hp = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
    f32x2 r1;
    r1[0] = exact_sub(x[0], __uint_as_float(hp << 16));
    r1[1] = exact_sub(x[1], __uint_as_float(hp & 0xffff0000u));
    const unsigned mp = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2));
    f32x2 r2;
    r2[0] = exact_sub(r1[0], __uint_as_float(mp << 16));
    r2[1] = exact_sub(r1[1], __uint_as_float(mp & 0xffff0000u));
    H[p] = hp;
    M[p] = mp;
    L[p] = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
  }
  hi = __builtin_bit_cast(bf16x8, H);
  mid = __builtin_bit_cast(bf16x8, M);
  lo = __builtin_bit_cast(bf16x8, L);
}

// acc += a * b for 16x16x32 blocks given as split fragments (a = MFMA "A" operand): the six
// products that matter, smallest first
__device__ __forceinline__ f32x4 mfma_split6(const bf16x8& ah, const bf16x8& am, const bf16x8& al,
                                             const bf16x8& bh, const bf16x8& bm, const bf16x8& bl, f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
  return acc;
}

// ---------------------------------------------------------------------------------------------------------
// Two-term fp16 split (DZN_PREC_F32_H2, "3xFP16"): see the header of gemm_split.hip.  NP = number of planes /
// terms per operand: 3 = bf16 (six products), 2 = fp16 (three products).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// a 128-bit MFMA operand fragment of either element type
template <int NP>
__device__ __forceinline__ f32x4 mfma_np(const u32x4& a, const u32x4& b, f32x4 c) {
  if constexpr (NP == 3)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// 8 fp32 values (two float4), pre-scaled by the exact power of two s -> two fp16x8 fragments (hi, lo)
// (r6) DZN_SPLIT_MIX: the same arithmetic on the mixed-precision FMA instructions — hi = f16(fma(x, s, 0)) and
// lo = f16(fma(x, s, -hi)) with hi read as an fp16 operand (v_fma_mixlo_f16 / v_fma_mixhi_f16: fp32 FMA, result rounded to
// nearest even into the low / high half of the destination) — 4 instructions per pair of values instead of the 6 hipcc picks for
// the natural form (v_pk_mul_f32, v_cvt_pk_f16_f32, 2 v_cvt_f32_f16, v_pk_fma_f32, v_cvt_pk_f16_f32), no conversions back, no
// packing.  x s is an exact scaling by a power of two and x s - hi is exact in fp32, so both roundings are the same single
// roundings as before: bit-identical fragments.  (A form on plain v_mul / v_cvt / v_sub / v_pack — 12 instructions per pair,
// none of them in the matrix pipe's way, which packed fp32 / packed converts are: profiles/r6_mfma_valu_coissue.txt — measured
// 3.5 % SLOWER on the 128 x 128 class, this form the same time as the default (1.349 vs 1.342 ms): the loop is not bound by its
// operand split; profiles/r6_split_forms_ab.txt.  Compile with -DDZN_SPLIT_MIX to select it.)
#ifdef DZN_SPLIT_MIX
__device__ __forceinline__ void split8_h2(const f32x4& u, const f32x4& v, float s, u32x4& hi, u32x4& lo) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float a0 = p < 2 ? u[2 * p] : v[2 * p - 4], a1 = p < 2 ? u[2 * p + 1] : v[2 * p - 3];
    unsigned hp, lp;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hp) : "v"(a0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hp) : "v"(a1), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(lp) : "v"(a0), "v"(s), "v"(hp));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lp) : "v"(a1), "v"(s), "v"(hp));
    hi[p] = hp;
    lo[p] = lp;
  }
}
#else
__device__ __forceinline__ void split8_h2(const f32x4& u, const f32x4& v, float s, u32x4& hi, u32x4& lo) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    f32x2 x;
    x[0] = (p < 2 ? u[2 * p] : v[2 * p - 4]) * s;
    x[1] = (p < 2 ? u[2 * p + 1] : v[2 * p - 3]) * s;
    const f16x2 h = __builtin_convertvector(x, f16x2);        // v_cvt_pk_f16_f32 (round to nearest even)
    const f32x2 hf = __builtin_convertvector(h, f32x2);
    f32x2 r;
    r[0] = x[0] - hf[0];                                      // exact in fp32
    r[1] = x[1] - hf[1];
    hi[p] = __builtin_bit_cast(unsigned, h);
    lo[p] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
  }
}
#endif

// DZN_PREC_F16: 8 fp32 values, pre-scaled by the power of two s -> ONE fp16x8 fragment (the leading term only)
__device__ __forceinline__ void cvt8_h1(const f32x4& u, const f32x4& v, float s, u32x4& hi) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    f32x2 x;
    x[0] = (p < 2 ? u[2 * p] : v[2 * p - 4]) * s;
    x[1] = (p < 2 ? u[2 * p + 1] : v[2 * p - 3]) * s;
    hi[p] = __builtin_bit_cast(unsigned, __builtin_convertvector(x, f16x2));
  }
}

// exact power-of-two scale that puts `amax` into [2^14, 2^15) (fp16 max is 65504), and its inverse
__device__ __forceinline__ void h2_scale(float amax, float& s, float& inv) {
  int e = (int)((__float_as_uint(amax) >> 23) & 0xff) - 127;   // floor(log2 amax) for normal amax
  if (!(amax > 0.f) || e > 100) e = 14;                         // empty / non-finite tracker: scale 1
  e = e < -100 ? -100 : e;
  s = __uint_as_float((unsigned)(14 - e + 127) << 23);
  inv = __uint_as_float((unsigned)(e - 14 + 127) << 23);
}


// the products that are summed, smallest first, as (term of operand A, term of operand B) with 0 = hi:
//   NP = 3: lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi        NP = 2: lo*hi, hi*lo, hi*hi
template <int NP> struct SplitTerms;
template <> struct SplitTerms<3> {
  static constexpr int N = 6;
  static constexpr int A[6] = {2, 0, 1, 1, 0, 0};
  static constexpr int B[6] = {0, 2, 1, 0, 1, 0};
};
template <> struct SplitTerms<2> {
  static constexpr int N = 3;
  static constexpr int A[3] = {1, 0, 0};
  static constexpr int B[3] = {0, 1, 0};
};
template <> struct SplitTerms<1> {     // leading fp16 term only (DZN_PREC_F16 single-term sites)
  static constexpr int N = 1;
  static constexpr int A[1] = {0};
  static constexpr int B[1] = {0};
};

// 8 fp32 values -> NP fragments (hi, [mid,] lo); `scale` (exact power of two) applies to NP = 2 only
template <int NP>
__device__ __forceinline__ void split_np(const f32x4& u, const f32x4& v, float scale, u32x4 (&f)[NP]) {
  if constexpr (NP == 3) {
    bf16x8 h_, m_, l_;
    split8(u, v, h_, m_, l_);
    f[0] = __builtin_bit_cast(u32x4, h_);
    f[1] = __builtin_bit_cast(u32x4, m_);
    f[2] = __builtin_bit_cast(u32x4, l_);
  } else if constexpr (NP == 2) {
    split8_h2(u, v, scale, f[0], f[1]);
  } else {
    cvt8_h1(u, v, scale, f[0]);
  }
}
