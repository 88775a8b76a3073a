// lds_fill_rate.hip — how fast can a CU fill LDS from global memory with LDS-DMA (global_load_lds_dwordx4), with NO other work?
//
// DESIGN.md 4.5 claims the dominant contraction (gemm_split_kernel<128,128,...>) is bound by its global -> LDS fill: 32 KB per
// K tile and workgroup, 9.3 TB/s chip-wide at 290 TFLOP/s.  This program issues the SAME fill pattern — 256-thread workgroups,
// 2 per CU (grid 512), 32 KB per "K tile" as 8 x 1-KiB wave instructions (128-byte rows like the A tile + 64-byte rows like
// the W planes), two LDS stages, s_waitcnt vmcnt(8) + s_barrier per tile as the kernel's pipeline does — and nothing else:
// no fragment reads, no MFMAs, no epilogue.  Sources: (a) an L2-resident region (every workgroup streams the same 2 MB, like
// the weight planes), (b) a per-workgroup streaming region of a 1 GiB buffer (like the A rows).  Prints TB/s per variant.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <bool SHARED_SRC>
__global__ __launch_bounds__(256, 2) void fill_kernel(const float* __restrict__ src, size_t wg_stride_floats, int tiles, float* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 stages x 32 KB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float* base = src + (SHARED_SRC ? 0 : (size_t)blockIdx.x * wg_stride_floats);
  // per tile: 8 instructions per thread, each wave instruction = 1 KiB contiguous in LDS; global side: 16 B per lane,
  // 8 lanes per 128-B row (rows 4 KiB apart for the first four = "A rows", contiguous for the last four = "W planes")
  for (int t = 0; t < tiles; ++t) {
    unsigned char* st = smem + (t & 1) * 32768 + wave * 1024;
    const float* g = base + (size_t)t * 8192;       // 32 KB of floats per tile
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float* p = g + (size_t)(i * 4 + wave) * 256 + lane * 4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                       (__attribute__((address_space(3))) void*)(st + i * 4096), 16, 0, 0);
    }
    if (t > 0) {
      __builtin_amdgcn_s_waitcnt((8 & 0xF) | ((8 >> 4) << 14) | (0x7 << 4) | (0xF << 8));   // vmcnt(8): the previous tile has landed
      __builtin_amdgcn_s_barrier();
    }
  }
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (tid == 0 && out) out[blockIdx.x] = reinterpret_cast<float*>(smem)[lane];
}

int main() {
  const size_t bytes = (size_t)1 << 30;
  float *buf, *out;
  CHECK(hipMalloc(&buf, bytes));
  CHECK(hipMalloc(&out, 4096 * 4));
  CHECK(hipMemset(buf, 0x3c, bytes));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fill_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fill_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  const int grid = 512;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int variant = 0; variant < 2; ++variant) {
    const int tiles = 64;                                    // 2 MB per workgroup
    const size_t stride = bytes / 4 / grid;                  // 2 MB apart: 512 workgroups x 2 MB = the whole 1 GiB buffer
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipEventRecord(e0));
      for (int it = 0; it < 20; ++it) {
        if (variant == 0) hipLaunchKernelGGL(fill_kernel<true>, dim3(grid), dim3(256), 65536, 0, buf, stride, tiles, out);
        else hipLaunchKernelGGL(fill_kernel<false>, dim3(grid), dim3(256), 65536, 0, buf, stride, tiles, out);
      }
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms = 0.f;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      const double tb = 20.0 * grid * tiles * 32768.0 / (ms * 1e-3) / 1e12;
      printf("%s  rep %d: %.2f TB/s of LDS-DMA fill = %.1f GB/s per CU = %.1f B/clk per CU at 2.1 GHz\n",
             variant == 0 ? "shared 2 MB source (L2-resident, like weight planes)" : "per-workgroup streaming source (1 GiB, like A rows) ",
             rep, tb, tb * 1e3 / 256.0, tb * 1e12 / 256.0 / 2.1e9);
    }
  }
  return 0;
}
