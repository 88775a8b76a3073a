// bf16 MFMA (16x16x32) issue rate, alone and with the fp32 -> 3 x bf16 operand split next to it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split_rne(const f32x4& u, const f32x4& v, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float x = e < 4 ? u[e] : v[e - 4];
    const __bf16 h = (__bf16)x;
    const float r1 = x - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    hi[e] = h; mid[e] = m; lo[e] = (__bf16)r2;
  }
}
// truncation split: hi = top 16 bits, exact remainders; packing by v_perm_b32
__device__ __forceinline__ void split_trunc(const f32x4& u, const f32x4& v, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
  u32x4 H, M, L;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float x0 = p < 2 ? u[2 * p] : v[2 * p - 4], x1 = p < 2 ? u[2 * p + 1] : v[2 * p - 3];
    const unsigned h0 = __float_as_uint(x0) & 0xffff0000u, h1 = __float_as_uint(x1) & 0xffff0000u;
    const float r0 = x0 - __uint_as_float(h0), r1 = x1 - __uint_as_float(h1);
    const unsigned m0 = __float_as_uint(r0) & 0xffff0000u, m1 = __float_as_uint(r1) & 0xffff0000u;
    const float s0 = r0 - __uint_as_float(m0), s1 = r1 - __uint_as_float(m1);
    H[p] = __builtin_amdgcn_perm(h1, h0, 0x07060302u);
    M[p] = __builtin_amdgcn_perm(m1, m0, 0x07060302u);
    L[p] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
  }
  hi = __builtin_bit_cast(bf16x8, H); mid = __builtin_bit_cast(bf16x8, M); lo = __builtin_bit_cast(bf16x8, L);
}

// RNE split on pairs, the two exact subtractions as scalar v_sub_f32 (inline asm keeps the SLP
// vectoriser from fusing them into v_pk_add_f32, which is expensive next to MFMAs)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float vsub(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
template <bool ASM>
__device__ __forceinline__ void split_pairs(const f32x4& u, const f32x4& v, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
  u32x4 H, M, L;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    f32x2 x; x[0] = p < 2 ? u[2 * p] : v[2 * p - 4]; x[1] = p < 2 ? u[2 * p + 1] : v[2 * p - 3];
    const unsigned hp = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
    f32x2 r1;
    if (ASM) { r1[0] = vsub(x[0], __uint_as_float(hp << 16)); r1[1] = vsub(x[1], __uint_as_float(hp & 0xffff0000u)); }
    else { f32x2 hf; hf[0] = __uint_as_float(hp << 16); hf[1] = __uint_as_float(hp & 0xffff0000u); r1 = x - hf; }
    const unsigned mp = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2));
    f32x2 r2;
    if (ASM) { r2[0] = vsub(r1[0], __uint_as_float(mp << 16)); r2[1] = vsub(r1[1], __uint_as_float(mp & 0xffff0000u)); }
    else { f32x2 mf; mf[0] = __uint_as_float(mp << 16); mf[1] = __uint_as_float(mp & 0xffff0000u); r2 = r1 - mf; }
    H[p] = hp; M[p] = mp; L[p] = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
  }
  hi = __builtin_bit_cast(bf16x8, H); mid = __builtin_bit_cast(bf16x8, M); lo = __builtin_bit_cast(bf16x8, L);
}

// MODE 0: MFMA only (96 per iteration, 16 accumulators); 1: + RNE split of 4 A fragments; 2: + truncation split
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters) {
  const int tid = threadIdx.x;
  bf16x8 w[4][3];
  f32x4 au[4], av[4];
  for (int j = 0; j < 4; ++j) {
    for (int p = 0; p < 3; ++p)
      for (int e = 0; e < 8; ++e) w[j][p][e] = (__bf16)in[(tid * 7 + j * 24 + p * 8 + e) & 16383];
    for (int e = 0; e < 4; ++e) { au[j][e] = in[(tid * 3 + j * 8 + e) & 16383]; av[j][e] = in[(tid * 5 + j * 8 + e + 4) & 16383]; }
  }
  f32x4 acc[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
  bf16x8 ah[4], am[4], al[4];
  for (int i = 0; i < 4; ++i) split_rne(au[i], av[i], ah[i], am[i], al[i]);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (MODE == 1) { split_rne(au[i], av[i], ah[i], am[i], al[i]); }
      if constexpr (MODE == 2) { split_trunc(au[i], av[i], ah[i], am[i], al[i]); }
      if constexpr (MODE == 3) { split_pairs<false>(au[i], av[i], ah[i], am[i], al[i]); }
      if constexpr (MODE == 4) { split_pairs<true>(au[i], av[i], ah[i], am[i], al[i]); }
      if constexpr (MODE >= 1) {   // keep the inputs live / changing so the split is not hoisted
        au[i][0] = __uint_as_float(__float_as_uint(au[i][0]) ^ (unsigned)it);
        asm volatile("" : "+v"(au[i]), "+v"(av[i]));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j][2], ah[i], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j][0], al[i], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j][1], am[i], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j][1], ah[i], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j][0], am[i], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j][0], ah[i], acc[i][j], 0, 0, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, float* out, float* in, int blocks) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, in, 100);
  (void)hipDeviceSynchronize();
  const int iters = 4000;
  (void)hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, in, iters);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  const double mf = 96.0 * iters * (double)blocks * 4;          // MFMA instructions
  const double flops = mf * 2 * 16 * 16 * 32;
  // cycles per MFMA per SIMD at 2.4 GHz: each CU has 4 SIMDs; blocks/256 workgroups per CU, 4 waves each
  const double cyc = ms * 1e-3 * 2.4e9 / (96.0 * iters * (blocks / 256.0));
  printf("%-34s blocks=%d %.3f ms  %.0f TF/s executed (%.0f TF/s fp32-equivalent)  %.1f cycles/MFMA/SIMD\n", name, blocks, ms,
         flops / ms / 1e9, flops / 6 / ms / 1e9, cyc);
}
int main(int argc, char** argv) {
  const bool rnd = argc > 1;
  float *out, *in;
  (void)hipMalloc(&out, 4096 * 256 * 4); (void)hipMalloc(&in, 16384 * 4);
  std::vector<float> h(16384);
  for (auto& v : h) v = rnd ? (float)rand() / RAND_MAX * 2.f - 1.f : 0.f;
  (void)hipMemcpy(in, h.data(), 16384 * 4, hipMemcpyHostToDevice);
  printf("data: %s\n", rnd ? "random" : "zeros");
  for (int blocks : {256, 512}) {
    run<0>("mfma only", out, in, blocks);
    run<1>("+ RNE split (cvt_pk_bf16)", out, in, blocks);
    run<2>("+ truncation split (and/perm)", out, in, blocks);
    run<3>("+ RNE pairs (v_pk_add_f32)", out, in, blocks);
    run<4>("+ RNE pairs (v_sub_f32 asm)", out, in, blocks);
  }
  return 0;
}
