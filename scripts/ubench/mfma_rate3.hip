// MFMA issue-rate microbenchmark #3: MFMA stream + LDS fragment reads (+ random data), 1-2 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: registers only; 1: re-read fragments from LDS every 64 MFMAs; 2: as 1 + __syncthreads per 128
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[16384];
  for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = in[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = *reinterpret_cast<const f32x4*>(lds + ((lane & 15) * 32 + (lane >> 4) * 4 + i * 512 + wave * 2048));
    b[i] = *reinterpret_cast<const f32x4*>(lds + ((lane & 15) * 32 + (lane >> 4) * 4 + i * 512 + 8192 + wave * 2048));
  }
  f32x4 acc[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if constexpr (MODE >= 1) {
        const int off = ((it * 2 + kb) & 3) * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          a[i] = *reinterpret_cast<const f32x4*>(lds + ((lane & 15) * 32 + (lane >> 4) * 4 + i * 512 + wave * 2048 + off) % 16384);
          b[i] = *reinterpret_cast<const f32x4*>(lds + ((lane & 15) * 32 + (lane >> 4) * 4 + i * 512 + 8192 + wave * 2048 + off) % 16384);
        }
      }
#pragma unroll
      for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j][st], a[i][st], acc[i][j], 0, 0, 0);
    }
    if constexpr (MODE == 2) __syncthreads();
    asm volatile("" ::: "memory");
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
  const int iters = 3000;
  (void)hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, in, iters);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  double flops = 128.0 * 2 * 16 * 16 * 4 * iters * (double)blocks * 4;
  printf("%-40s blocks=%d %.3f ms %.1f TF/s\n", name, blocks, ms, flops / ms / 1e9);
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
    run<0>("regs only", out, in, blocks);
    run<1>("+ 8 ds_read_b128 per 64 mfma", out, in, blocks);
    run<2>("+ ds_reads + barrier per 128 mfma", out, in, blocks);
  }
  return 0;
}
