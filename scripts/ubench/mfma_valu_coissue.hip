// Do an MFMA wavefront and a VALU wavefront of the SAME SIMD overlap?  512-thread workgroups, one per CU (wavefronts 0-3:
// role A, 4-7: role B, i.e. one of each per SIMD).  Role A issues independent v_mfma_f32_16x16x32_f16 back to back, role B
// independent v_pk_fma_f32 (or v_fma_f32 / v_exp_f32) back to back.  Timed: A alone, B alone, both.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_valu_coissue.hip -o /tmp/coissue && /tmp/coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <int VKIND>
__global__ __launch_bounds__(512, 1) void k(float* out, int iters, int mode) {      // mode bit 0: role A works, bit 1: role B works
  const int wave = threadIdx.x >> 6;
  const bool roleA = wave < 4;
  if (roleA) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f32x4 acc[8];
    for (int j = 0; j < 8; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (mode & 1)
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j], 0, 0, 0);
      }
    float s = 0.f;
    for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  } else {
    v2f x[8];
    for (int j = 0; j < 8; ++j) x[j] = (v2f){threadIdx.x * 0.001f + j, 1.0f + j};
    const v2f m = {1.0001f, 0.9999f}, c = {0.001f, -0.001f};
    if (mode & 2)
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if constexpr (VKIND == 0) x[j] = __builtin_elementwise_fma(x[j], m, c);                 // v_pk_fma_f32
            else if constexpr (VKIND == 1) { x[j][0] = __builtin_fmaf(x[j][0], m[0], c[0]); }      // v_fma_f32
            else if constexpr (VKIND == 2) { x[j][0] = __builtin_amdgcn_exp2f(x[j][0]); }               // v_exp_f32 (transcendental)
            else if constexpr (VKIND == 3) { x[j] = x[j] * m; }                                          // v_pk_mul_f32
            else if constexpr (VKIND == 4) { x[j] = x[j] + c; }                                          // v_pk_add_f32
            else if constexpr (VKIND == 5) {                                                             // v_cvt_pk_f16_f32 (+ v_cvt_f32_f16 back)
              typedef _Float16 h2 __attribute__((ext_vector_type(2)));
              const h2 h = __builtin_convertvector(x[j], h2);
              x[j][0] = (float)h[0] + x[j][1];
            }
            else if constexpr (VKIND == 6) { x[j][0] = x[j][0] * m[0]; }                                 // v_mul_f32
            else if constexpr (VKIND == 7) {                                                             // v_fma_mixlo_f16 (fp32 FMA -> f16 half)
              unsigned h = __float_as_uint(x[j][1]);
              asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x[j][0]), "v"(m[0]));
              x[j][1] = __uint_as_float(h);
            }
            else if constexpr (VKIND == 8) {                                                             // v_cvt_f16_f32 (plain convert)
              unsigned h;
              asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(h) : "v"(x[j][0]));
              x[j][1] = __uint_as_float(h);
            }
            else {                                                                                       // v_pack_b32_f16
              unsigned h;
              asm volatile("v_pack_b32_f16 %0, %1, %2" : "=v"(h) : "v"(x[j][0]), "v"(x[j][1]));
              x[j][1] = __uint_as_float(h);
            }
          }
      }
    float s = 0.f;
    for (int j = 0; j < 8; ++j) s += x[j][0] + x[j][1];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  }
}

template <int VKIND>
void run(const char* name, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  printf("%s: per iteration a role-A wavefront issues 8 MFMAs (128 matrix-pipe cycles), a role-B wavefront 16 VALU ops\n", name);
  for (int mode = 1; mode <= 3; ++mode) {
    hipLaunchKernelGGL(k<VKIND>, dim3(256), dim3(512), 0, 0, out, 100, mode);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<VKIND>, dim3(256), dim3(512), 0, 0, out, iters, mode);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("  %-10s %8.3f ms  = %7.1f ns per iteration\n", mode == 1 ? "MFMA only" : mode == 2 ? "VALU only" : "both", ms, ms * 1e6 / iters);
  }
}

int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  run<0>("v_pk_fma_f32", out);
  run<1>("v_fma_f32", out);
  run<2>("v_exp_f32", out);
  run<3>("v_pk_mul_f32", out);
  run<4>("v_pk_add_f32", out);
  run<5>("v_cvt_pk_f16_f32 + v_cvt_f32_f16 + v_add_f32", out);
  run<6>("v_mul_f32", out);
  run<7>("v_fma_mixlo_f16", out);
  run<8>("v_cvt_f16_f32", out);
  run<9>("v_pack_b32_f16", out);
  return 0;
}
