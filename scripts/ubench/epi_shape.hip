// What does the ACCESS SHAPE of the contraction epilogue cost?  (r6 probe for the short-K class: its launches are an HBM-rate
// epilogue - residual in + C out, 1.83 GB at 4.1 TB/s - plus an MFMA-rate loop)
// The MFMA accumulator layout makes one wavefront instruction touch 16 ROWS x 64 BYTES (lane (lr, lq): row lr, 4 floats at
// column 16 j + 4 lq); the same bytes could leave as 4 rows x 256 B after a transpose through LDS.  This microbenchmark moves
// C[m][n] = R[m][n] + 1 over [M][1024] fp32 with both lane maps, same tiles (128 rows x 64 columns per workgroup, 4 wavefronts of
// 32 rows), same bytes, nothing else:  hipcc --offload-arch=gfx950 -O3 epi_shape.hip -o epi_shape && ./epi_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int SHAPE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ R, float* __restrict__ C, int M, int N) {
  const int tilesN = N / 64;
  const int tm = blockIdx.x / tilesN, tn = blockIdx.x % tilesN;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row0 = tm * 128 + wave * 32, col0 = tn * 64;
  float4 v[8];
  if constexpr (SHAPE == 0) {            // accumulator shape: block (i, j): row 16 i + lr, columns 16 j + 4 lq
    const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = row0 + 16 * i + lr;
        v[i * 4 + j] = m < M ? *reinterpret_cast<const float4*>(R + (size_t)m * N + col0 + 16 * j + 4 * lq) : make_float4(0, 0, 0, 0);
      }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = row0 + 16 * i + lr;
        float4 x = v[i * 4 + j];
        x.x += 1.f; x.y += 1.f; x.z += 1.f; x.w += 1.f;
        if (m < M) *reinterpret_cast<float4*>(C + (size_t)m * N + col0 + 16 * j + 4 * lq) = x;
      }
  } else if constexpr (SHAPE == 2) {     // full rows in HBM, accumulator shape in registers, transposed through wave-private LDS
    __shared__ __attribute__((aligned(16))) float scr[4][16 * 68];
    float* my = scr[wave];
    const int lr = lane & 15, lq = lane >> 4, fr = lane >> 4, fc = (lane & 15) * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {        // one 16-row block at a time
      float4 r[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int m = row0 + 16 * i + 4 * t + fr;
        r[t] = m < M ? *reinterpret_cast<const float4*>(R + (size_t)m * N + col0 + fc) : make_float4(0, 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) *reinterpret_cast<float4*>(my + (4 * t + fr) * 68 + fc) = r[t];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      float4 a[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = *reinterpret_cast<const float4*>(my + lr * 68 + 16 * j + 4 * lq);
#pragma unroll
      for (int j = 0; j < 4; ++j) { a[j].x += 1.f; a[j].y += 1.f; a[j].z += 1.f; a[j].w += 1.f; }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(my + lr * 68 + 16 * j + 4 * lq) = a[j];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int m = row0 + 16 * i + 4 * t + fr;
        const float4 x = *reinterpret_cast<const float4*>(my + (4 * t + fr) * 68 + fc);
        if (m < M) *reinterpret_cast<float4*>(C + (size_t)m * N + col0 + fc) = x;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
  } else {                               // full rows: instruction t: rows 4 t + lane / 16, columns 4 (lane % 16): 4 rows x 256 B
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int m = row0 + 4 * t + (lane >> 4);
      v[t] = m < M ? *reinterpret_cast<const float4*>(R + (size_t)m * N + col0 + 4 * (lane & 15)) : make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int m = row0 + 4 * t + (lane >> 4);
      float4 x = v[t];
      x.x += 1.f; x.y += 1.f; x.z += 1.f; x.w += 1.f;
      if (m < M) *reinterpret_cast<float4*>(C + (size_t)m * N + col0 + 4 * (lane & 15)) = x;
    }
  }
}
int main() {
  const int M = 223839, N = 1024;
  float *R, *C;
  hipMalloc(&R, (size_t)M * N * 4); hipMalloc(&C, (size_t)M * N * 4);
  hipMemset(R, 0, (size_t)M * N * 4);
  const int grid = ((M + 127) / 128) * (N / 64);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int shape = 0; shape < 3; ++shape)
    for (int rep = 0; rep < 2; ++rep) {
      auto go = [&] {
        if (shape == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, R, C, M, N);
        else if (shape == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, R, C, M, N);
        else hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, R, C, M, N);
      };
      for (int w = 0; w < 3; ++w) go();
      hipEventRecord(a);
      for (int w = 0; w < 20; ++w) go();
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b); ms /= 20;
      printf("%s: %.1f us per pass, %.2f TB/s (read + write %.2f GB)\n", shape == 0 ? "accumulator shape (16 rows x 64 B per instruction)" : shape == 1 ? "full rows (4 rows x 256 B per instruction)      " : "full rows in HBM, accumulator shape via LDS     ",
             ms * 1e3, 2.0 * M * N * 4 / (ms * 1e-3) / 1e12, 2.0 * M * N * 4 / 1e9);
    }
  return 0;
}
