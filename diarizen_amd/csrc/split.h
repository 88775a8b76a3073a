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
