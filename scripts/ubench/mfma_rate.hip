// MFMA issue-rate microbenchmark (f32 16x16x4 vs 32x32x2), 1 or 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a0, float b0) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F>
void run(const char* name, F launch, double flop_per_thread_iter_wave, int blocks) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  launch(100);
  hipDeviceSynchronize();
  hipEventRecord(a); launch(4000); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double flops = flop_per_thread_iter_wave * 4000.0 * blocks * 4;
  printf("%-28s blocks=%d  %.3f ms  %.1f TF/s\n", name, blocks, ms, flops / ms / 1e9);
}
int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  for (int blocks : {256, 512}) {
    run("16x16x4 nacc=4", [&](int it) { hipLaunchKernelGGL(k16<4>, dim3(blocks), dim3(256), 0, 0, out, it, 1.f, 2.f); }, 4 * 2.0 * 16 * 16 * 4, blocks);
    run("16x16x4 nacc=16", [&](int it) { hipLaunchKernelGGL(k16<16>, dim3(blocks), dim3(256), 0, 0, out, it, 1.f, 2.f); }, 16 * 2.0 * 16 * 16 * 4, blocks);
    run("32x32x2 nacc=2", [&](int it) { hipLaunchKernelGGL(k32<2>, dim3(blocks), dim3(256), 0, 0, out, it, 1.f, 2.f); }, 2 * 2.0 * 32 * 32 * 2, blocks);
    run("32x32x2 nacc=4", [&](int it) { hipLaunchKernelGGL(k32<4>, dim3(blocks), dim3(256), 0, 0, out, it, 1.f, 2.f); }, 4 * 2.0 * 32 * 32 * 2, blocks);
  }
  return 0;
}
