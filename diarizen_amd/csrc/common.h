// common.h — shared device/host helpers for the gfx950 kernels of libdzn_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <string.h>
#include "../../include/dzn.h"
#include "../../include/dzn_ops.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

#define DZN_WAVE 64

// ---- wave-level reductions (64-wide wavefront) ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- activations (match torch CPU fp32 semantics) ----
__device__ __forceinline__ float gelu_erf(float x) {
  // torch.nn.functional.gelu(approximate='none'): 0.5 * x * (1 + erf(x / sqrt(2)))
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float swishf_(float x) { return x * sigmoidf_(x); }

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case DZN_ACT_GELU: return gelu_erf(v);
    case DZN_ACT_SWISH: return swishf_(v);
    case DZN_ACT_RELU: return fmaxf(v, 0.0f);
    default: return v;
  }
}

// host-side float -> bf16 (round to nearest even), used when packing weights
static inline u16 f32_to_bf16_host(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u16)((u >> 16) | 0x40);  // NaN
  uint32_t r = 0x7fffu + ((u >> 16) & 1u);
  return (u16)((u + r) >> 16);
}

// ---- kernel launchers (implemented in the .hip files) ----
int launch_gemm(const dzn_gemm_desc& d, hipStream_t s);
int launch_layernorm(const float* x, int64_t ldx, float* y, int64_t ldy, const float* g,
                     const float* b, int64_t rows, int C, int Cpad, float eps, int gelu,
                     hipStream_t s);
int launch_wave_stats(const float* w, int B, int N, float eps, float* stats, hipStream_t s);
int launch_gate(const float* y, int64_t ldy, const float* Wg, const float* bg, const float* cst,
                float* gate, int64_t rows, int Htot, hipStream_t s);
int launch_attention(const float* qkv, float* out, const float* gate, const float* table,
                     const int32_t* head_idx, int B, int L, int h, int Htot, int ldqkv, int ldo,
                     float scale, int precision, hipStream_t s);

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }
