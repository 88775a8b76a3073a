// MFMA issue-rate microbenchmark #2: operand patterns of the GEMM inner loop (1 wave / SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// variant 0: 4x4 blocks, a[i], b[j] distinct registers, order s,i,j (as in the kernel)
// variant 1: same but order s,j,i
// variant 2: 32x32x2 MFMA, 2x2 blocks
template <int V>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters) {
  f32x4 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = *reinterpret_cast<const f32x4*>(in + threadIdx.x * 4 + i * 1024);
    b[i] = *reinterpret_cast<const f32x4*>(in + threadIdx.x * 4 + i * 1024 + 4096);
  }
  float s = 0;
  if constexpr (V < 2) {
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        if constexpr (V == 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j][st], a[i][st], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j][st], a[i][st], acc[i][j], 0, 0, 0);
        }
      }
      asm volatile("" ::: "memory");
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
  } else {
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int h = 0; h < 2; ++h)   // two k-pairs per 16x16x4-equivalent step
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j + 2 * h][st], a[i + 2 * h][st], acc[i][j], 0, 0, 0);
      asm volatile("" ::: "memory");
    }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][15];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int V>
void run(const char* name, float* out, float* in, int blocks) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, out, in, 100);
  (void)hipDeviceSynchronize();
  const int iters = 4000;
  (void)hipEventRecord(a);
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, out, in, iters);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  double flops = 64.0 * 2 * 16 * 16 * 4 * iters * (double)blocks * 4;   // 64 16x16x4-equivalents / iter / wave
  printf("%-34s blocks=%d %.3f ms %.1f TF/s\n", name, blocks, ms, flops / ms / 1e9);
}
int main() {
  float *out, *in;
  (void)hipMalloc(&out, 4096 * 256 * 4); (void)hipMalloc(&in, 65536 * 4);
  (void)hipMemset(in, 0, 65536 * 4);
  for (int blocks : {256, 512}) {
    run<0>("16x16x4 4x4 distinct regs s,i,j", out, in, blocks);
    run<1>("16x16x4 4x4 distinct regs s,j,i", out, in, blocks);
    run<2>("32x32x2 2x2 distinct regs", out, in, blocks);
  }
  return 0;
}
