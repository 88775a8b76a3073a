// bf16 MFMA shape comparison under real (random) operands: 16x16x32 vs 32x32x16, register operands only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters) {
  const int tid = threadIdx.x;
  bf16x8 a[4], b[4];
  for (int j = 0; j < 4; ++j)
    for (int e = 0; e < 8; ++e) {
      a[j][e] = (__bf16)in[(tid * 7 + j * 8 + e) & 16383];
      b[j][e] = (__bf16)in[(tid * 5 + j * 8 + e + 64) & 16383];
    }
  float s = 0;
  if constexpr (SHAPE == 16) {
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
  } else {
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i + 2 * (r & 1)], b[j + 2 * ((r >> 1) & 1)], acc[i][j], 0, 0, 0);
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][15];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int SHAPE>
void run(const char* name, float* out, float* in, int blocks) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 0, 0, out, in, 100);
  (void)hipDeviceSynchronize();
  const int iters = 6000;
  (void)hipEventRecord(a);
  hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 0, 0, out, in, iters);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  // both variants execute 64 x 16384 = 32 x 32768 flops per iteration per wave
  const double flops = 64.0 * 16384.0 * iters * (double)blocks * 4;
  printf("%-12s blocks=%d %.3f ms  %.0f TF/s\n", name, blocks, ms, flops / ms / 1e9);
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
    run<16>("16x16x32", out, in, blocks);
    run<32>("32x32x16", out, in, blocks);
  }
  return 0;
}
