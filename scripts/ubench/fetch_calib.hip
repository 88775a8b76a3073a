// fetch_calib.hip — what does rocprofv3's FETCH_SIZE report for a KNOWN byte count, per access pattern?  
//
// The guide (MI355X_MICROARCH.md, HBM) says FETCH_SIZE on gfx950 reports exactly half the bytes of a wide coalesced
// streaming read (16 B per lane: 128-B requests tallied at 64 B) and leaves other widths uncalibrated.  r3 applied the
// x2 to every kernel, which put conv3x3_c32_split_kernel at 11.9 GB per launch for 7.6 GB algorithmic (1.56x), while its
// RAW counter equalled the algorithmic reads.  The stage-1 conv reads its strip as two float4 per thread, 64 B apart,
// four lanes per 128-B pixel (conv_split.hip: fetch) — i.e. 64-byte segments per instruction, not 1 KiB per wavefront.
// This program reads the same 2 GiB buffer (8 x the 256 MiB Infinity Cache) once per kernel with
//   calib_wide     : 16 B per lane, 64 consecutive lanes = 1 KiB contiguous per wave instruction
//   calib_conv_a/b : the conv pattern — lane l reads 16 B at pixel (l / 4) * 128 B + (l % 4) * 16 B (instruction A) and the
//                    same + 64 B (instruction B); _a issues only A (half the bytes), _ab issues both (all bytes)
//   calib_dword    : 4 B per lane
// and prints the byte counts; run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and divide.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void calib_wide(const float4* __restrict__ p, size_t n16, float* out) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
    const float4 v = p[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 123.456f) out[0] = acc;
}

template <bool BOTH>
__global__ __launch_bounds__(256) void calib_conv(const float* __restrict__ p, size_t npix, float* out) {
  float acc = 0.f;
  // thread -> (pixel = g / 4, chunk = g % 4): channels 4c..4c+3 and 16+4c..16+4c+3 of a 32-channel fp32 pixel
  for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < npix * 4; g += (size_t)gridDim.x * 256) {
    const float* src = p + (g >> 2) * 32 + 4 * (g & 3);
    const float4 u = *reinterpret_cast<const float4*>(src);
    acc += u.x + u.y + u.z + u.w;
    if (BOTH) {
      const float4 v = *reinterpret_cast<const float4*>(src + 16);
      acc += v.x + v.y + v.z + v.w;
    }
  }
  if (acc == 123.456f) out[0] = acc;
}

__global__ __launch_bounds__(256) void calib_dword(const float* __restrict__ p, size_t n, float* out) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += p[i];
  if (acc == 123.456f) out[0] = acc;
}

int main() {
  const size_t bytes = (size_t)2 << 30;
  float *buf, *out;
  CHECK(hipMalloc(&buf, bytes));
  CHECK(hipMalloc(&out, 4));
  CHECK(hipMemset(buf, 0x3c, bytes));
  CHECK(hipDeviceSynchronize());
  const int grid = 256 * 8;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(calib_wide, dim3(grid), dim3(256), 0, 0, reinterpret_cast<const float4*>(buf), bytes / 16, out);
    hipLaunchKernelGGL(calib_conv<true>, dim3(grid), dim3(256), 0, 0, buf, bytes / 128, out);
    hipLaunchKernelGGL(calib_conv<false>, dim3(grid), dim3(256), 0, 0, buf, bytes / 128, out);
    hipLaunchKernelGGL(calib_dword, dim3(grid), dim3(256), 0, 0, buf, bytes / 4, out);
    CHECK(hipDeviceSynchronize());
  }
  printf("known bytes per launch: calib_wide %zu, calib_conv<true> %zu, calib_conv<false> %zu (touches every line, uses half), calib_dword %zu\n",
         bytes, bytes, bytes / 2, bytes);
  return 0;
}
