// mx_probe.hip — pins, on the device, what csrc/gemm_mx.hip assumes about gfx950's block-scaled fp8 matrix instruction and
// the fp8 conversion (nothing in the offline guides states the A / B operand layout of v_mfma_scale_f32_32x32x64_f8f6f4):
//   1. v_cvt_pk_fp8_f32 = OCP e4m3fn, round to nearest even, byte order (a -> byte 0, b -> byte 1 of the selected word);
//   2. operand layout: lane l of the first operand = row (l & 31) of the 32 x 32 result's M index, its 32 bytes = 32 k values of
//      k half (l >> 5); likewise the second operand for the N index; byte b of k half h of BOTH operands is the same k
//      (established by a full one-hot scan: 64 x 64 (half, byte) pairs);
//   3. C / D layout = the 32x32 map of the f16 forms (col = lane & 31 -> N index, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5));
//   4. block scale = 2^(byte0 - 127) per operand, from a VGPR; whether op_sel picks another byte; what a literal does;
//   5. rate of the scaled fp8 form against v_mfma_f32_32x32x16_f16.
// Build: hipcc -O2 --offload-arch=gfx950 scripts/ubench/mx_probe.hip -o scripts/ubench/mx_probe     Run on the GPU box.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

static float e4m3_decode(uint8_t v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float x;
  if (e == 15 && m == 7) return NAN;
  if (e == 0) x = ldexpf((float)m / 8.f, -6);
  else x = ldexpf(1.f + (float)m / 8.f, e - 7);
  return s ? -x : x;
}
// nearest-even onto the e4m3fn grid, saturating at 448 (host model of the conversion)
static uint8_t e4m3_encode(float x) {
  uint8_t best = 0;
  float bd = INFINITY;
  for (int v = 0; v < 256; ++v) {
    const float y = e4m3_decode((uint8_t)v);
    if (std::isnan(y)) continue;
    const float d = fabsf(y - x);
    if (d < bd || (d == bd && !(v & 1) && (best & 1))) { bd = d; best = (uint8_t)v; }
  }
  return best;
}

__global__ void cvt_kernel(const float* x, int n, uint32_t* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n / 4) return;
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(x[4 * i], x[4 * i + 1], w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(x[4 * i + 2], x[4 * i + 3], w, true);
  out[i] = (uint32_t)w;
}

typedef short s16x2 __attribute__((ext_vector_type(2)));
// 6. v_cvt_scalef32_pk_fp8_f32: does the conversion divide or multiply by its scale operand?
__global__ void cvt_scale_kernel(const float* x, float scale, uint32_t* out) {
  s16x2 r = {0, 0};
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, x[0], x[1], scale, false);
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, x[2], x[3], scale, true);
  out[0] = __builtin_bit_cast(uint32_t, r);
}

// one wavefront: D = mfma(A, B) with per-lane operand images a[64][8], b[64][8]; d[64][16]
__global__ void mx32_kernel(const int* a, const int* b, float* d, int sa, int sb, int mode) {
  const int l = threadIdx.x;
  i32x8 A, B;
  for (int i = 0; i < 8; ++i) { A[i] = a[l * 8 + i]; B[i] = b[l * 8 + i]; }
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  if (mode == 0) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc, 0, 0, 0, sa, 0, sb);
  else if (mode == 1) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc, 0, 0, 1, sa, 1, sb);   // op_sel = byte 1
  else if (mode == 2) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc, 0, 0, 0, 127, 0, 127); // literal scales
  else acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc, 0, 0, 0, 0, 0, 0);                    // literal zero
  for (int i = 0; i < 16; ++i) d[l * 16 + i] = acc[i];
}
__global__ void mx16_kernel(const int* a, const int* b, float* d, int sa, int sb) {
  const int l = threadIdx.x;
  i32x8 A, B;
  for (int i = 0; i < 8; ++i) { A[i] = a[l * 8 + i]; B[i] = b[l * 8 + i]; }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, acc, 0, 0, 0, sa, 0, sb);
  for (int i = 0; i < 4; ++i) d[l * 4 + i] = acc[i];
}

// rate: NIT dependent-free MFMAs on 4 accumulators per wavefront
template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, int sa) {
  i32x8 A, B;
  for (int i = 0; i < 8; ++i) { A[i] = 0x38383838 + threadIdx.x + i; B[i] = 0x3c3c3838 ^ (threadIdx.x * 7 + i); }
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  f16x8 ha, hb;
  for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(0.5f + threadIdx.x * 0.01f); hb[i] = (_Float16)(1.5f - i * 0.1f); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (KIND == 0) acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc[j], 0, 0, 0, sa, 0, sa);
      else if constexpr (KIND == 1) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[j], 0, 0, 0);
      else {   // the mix of csrc/gemm_mx.hip per 64 k of a 32 x 32 block: 4 fp16 products + 2 scaled fp8 products
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hb, ha, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, ha, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hb, hb, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc[j], 0, 0, 0, sa, 0, sa);
        acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B, A, acc[j], 0, 0, 0, sa, 0, sa);
      }
    }
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 16; ++i) s += acc[j][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  int bad = 0;
  // ---- 1. conversion ----
  {
    std::vector<float> x;
    for (int v = 0; v < 256; ++v) {
      const float y = e4m3_decode((uint8_t)v);
      if (!std::isnan(y)) x.push_back(y);
    }
    const float extra[] = {0.f, 1e-4f, 0.0009765625f, 0.00146484375f, 0.002f, 0.0029296875f, 0.017f, 0.9f, 1.0625f, 1.1875f, 3.3f, 17.f, 100.f, 239.9f, 247.9f,
                           248.f, 250.f, 255.9f, 300.f, 447.f, 448.f, 460.f, 464.f, 480.f, 1000.f, -0.3f, -5.5f, -100.f, -500.f, 2.5f, 3.5f, 0.03f};
    for (float e : extra) x.push_back(e);
    srand(1);
    while (x.size() % 4 || x.size() < 1024) x.push_back(ldexpf((float)(rand() % 65536) / 65536.f * 2.f - 1.f, rand() % 16 - 8));
    float* dx; uint32_t* dout;
    CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&dout, x.size()));
    CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(cvt_kernel, dim3((x.size() / 4 + 63) / 64), dim3(64), 0, 0, dx, (int)x.size(), dout);
    std::vector<uint8_t> out(x.size());
    CK(hipMemcpy(out.data(), dout, x.size(), hipMemcpyDeviceToHost));
    int mism = 0;
    for (size_t i = 0; i < x.size(); ++i) {
      const uint8_t want = e4m3_encode(x[i]);
      if (e4m3_decode(out[i]) != e4m3_decode(want) && !(fabsf(x[i]) > 448.f)) {
        if (mism < 12) printf("  cvt: x = %.9g -> 0x%02x (%.6g), host model 0x%02x (%.6g)\n", x[i], out[i], e4m3_decode(out[i]), want, e4m3_decode(want));
        ++mism;
      }
      if (fabsf(x[i]) > 448.f) printf("  cvt beyond range: x = %g -> 0x%02x (%g)\n", x[i], out[i], e4m3_decode(out[i]));
    }
    printf("1. v_cvt_pk_fp8_f32 vs OCP e4m3fn nearest-even, byte order a,b -> bytes 0,1 of the word: %d mismatches of %zu\n", mism, x.size());
    bad += mism != 0;
  }
  int *da, *db; float* dd;
  CK(hipMalloc(&da, 64 * 32)); CK(hipMalloc(&db, 64 * 32)); CK(hipMalloc(&dd, 64 * 16 * 4));
  std::vector<uint8_t> a(2048), b(2048);
  std::vector<float> d(1024);
  auto run32 = [&](int sa, int sb, int mode) {
    CK(hipMemcpy(da, a.data(), 2048, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, b.data(), 2048, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(mx32_kernel, dim3(1), dim3(64), 0, 0, da, db, dd, sa, sb, mode);
    CK(hipMemcpy(d.data(), dd, 4096, hipMemcpyDeviceToHost));
  };
  // D as a matrix under the assumed C layout: D[mrow][ncol], lane l reg r: ncol = l & 31, mrow = (r & 3) + 8 (r >> 2) + 4 (l >> 5)
  auto Dm = [&](int mrow, int ncol) {
    for (int hl = 0; hl < 2; ++hl)
      for (int r = 0; r < 16; ++r)
        if ((r & 3) + 8 * (r >> 2) + 4 * hl == mrow) return d[(hl * 32 + ncol) * 16 + r];
    return NAN;
  };
  const uint8_t ONE = 0x38;   // 1.0 in e4m3 (bias 7)
  // ---- 2/3. layout: random small-integer operands against the host product under the assumed maps ----
  {
    srand(7);
    const uint8_t vals[] = {0x00, 0x38, 0xb8, 0x40, 0xc0, 0x44, 0x30, 0xb0, 0x48};   // 0, 1, -1, 2, -2, 3, 0.5, -0.5, 4
    for (int i = 0; i < 2048; ++i) { a[i] = vals[rand() % 9]; b[i] = vals[rand() % 9]; }
    run32(127, 127, 0);
    double maxerr = 0;
    for (int m = 0; m < 32; ++m)
      for (int n = 0; n < 32; ++n) {
        double s = 0;
        for (int h = 0; h < 2; ++h)
          for (int bb = 0; bb < 32; ++bb) s += (double)e4m3_decode(a[(h * 32 + m) * 32 + bb]) * e4m3_decode(b[(h * 32 + n) * 32 + bb]);
        maxerr = fmax(maxerr, fabs(s - Dm(m, n)));
      }
    printf("2. 32x32x64 fp8, operands lane = (k half, row), 32 bytes = 32 k; C = 32x32 map of the f16 forms: max |err| %.3g %s\n", maxerr,
           maxerr == 0 ? "(exact: layout as assumed)" : "(LAYOUT DIFFERS)");
    bad += maxerr != 0;
  }
  // one-hot scan: which (half, byte) of B meets (half, byte) of A
  {
    int offdiag = 0, missing = 0;
    for (int ha = 0; ha < 2; ++ha)
      for (int ba = 0; ba < 32; ++ba) {
        memset(a.data(), 0, 2048);
        a[(ha * 32 + 5) * 32 + ba] = ONE;      // row 5
        int hits = 0, hit_h = -1, hit_b = -1;
        for (int hb = 0; hb < 2; ++hb)
          for (int bb = 0; bb < 32; ++bb) {
            memset(b.data(), 0, 2048);
            for (int n = 0; n < 32; ++n) b[(hb * 32 + n) * 32 + bb] = ONE;
            run32(127, 127, 0);
            if (Dm(5, 9) != 0.f) { ++hits; hit_h = hb; hit_b = bb; }
          }
        if (hits != 1) { ++missing; printf("  one-hot A (half %d, byte %d): %d matching B positions\n", ha, ba, hits); }
        else if (hit_h != ha || hit_b != ba) { if (offdiag < 16) printf("  A (half %d, byte %d) meets B (half %d, byte %d)\n", ha, ba, hit_h, hit_b); ++offdiag; }
      }
    printf("   one-hot scan of the 64 (half, byte) positions: %d off-diagonal, %d without a unique partner %s\n", offdiag, missing,
           offdiag + missing == 0 ? "(byte b of half h is the same k in both operands)" : "(K ORDER DIFFERS BETWEEN THE OPERANDS)");
    bad += offdiag + missing != 0;
  }
  // ---- 4. scales ----
  {
    for (int i = 0; i < 2048; ++i) a[i] = b[i] = ONE;
    run32(127, 127, 0);
    printf("4. ones x ones (K = 64), scales 127 / 127 from registers: D = %g (expect 64)\n", d[0]);
    bad += d[0] != 64.f;
    run32(127 + 3, 127, 0);
    printf("   scale_a = 130: D = %g (expect 512)\n", d[0]);
    bad += d[0] != 512.f;
    run32(127, 127 - 11, 0);
    printf("   scale_b = 116: D = %g (expect %g)\n", d[0], 64.0 / 2048);
    bad += d[0] != 64.f / 2048;
    run32(127 | (130 << 8), 127 | (116 << 8), 1);
    printf("   op_sel = 1 with bytes (127, 130) / (127, 116): D = %g (byte 1 honoured: %g; byte 0 taken: 64)\n", d[0], 512.0 / 2048);
    run32(0, 0, 2);
    printf("   literal 127 as both scale arguments: D = %g\n", d[0]);
    run32(0, 0, 3);
    printf("   literal 0 as both scale arguments: D = %g\n", d[0]);
  }
  // ---- 16x16x128 (for the record) ----
  {
    srand(9);
    const uint8_t vals[] = {0x00, 0x38, 0xb8, 0x40, 0xc0, 0x44, 0x30, 0xb0, 0x48};
    for (int i = 0; i < 2048; ++i) { a[i] = vals[rand() % 9]; b[i] = vals[rand() % 9]; }
    CK(hipMemcpy(da, a.data(), 2048, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, b.data(), 2048, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(mx16_kernel, dim3(1), dim3(64), 0, 0, da, db, dd, 127, 127);
    CK(hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost));
    double maxerr = 0;
    for (int m = 0; m < 16; ++m)
      for (int n = 0; n < 16; ++n) {
        double s = 0;
        for (int q = 0; q < 4; ++q)
          for (int bb = 0; bb < 32; ++bb) s += (double)e4m3_decode(a[(q * 16 + m) * 32 + bb]) * e4m3_decode(b[(q * 16 + n) * 32 + bb]);
        // C: col = lane & 15 (N), row = 4 (lane >> 4) + reg (M)
        maxerr = fmax(maxerr, fabs(s - d[((m >> 2) * 16 + n) * 4 + (m & 3)]));
      }
    printf("   16x16x128 fp8 under the analogous maps (lane = (k quarter, row)): max |err| %.3g\n", maxerr);
  }
  // ---- 5. rates ----
  {
    float* dout;
    const int blocks = 256 * 8, iters = 2000;
    CK(hipMalloc(&dout, blocks * 256 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](auto kern, const char* name, double flop_per_iter) {
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, dout, 10, 127);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, dout, iters, 127);
      CK(hipEventRecord(e1));
      CK(hipDeviceSynchronize());
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double fl = (double)blocks * 4 * iters * flop_per_iter;
      printf("5. %-44s %8.3f ms  %8.1f TFLOP/s\n", name, ms, fl / ms * 1e-9);
    };
    time(rate_kernel<0>, "v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 x fp8)", 4 * 2.0 * 32 * 32 * 64);
    time(rate_kernel<1>, "v_mfma_f32_32x32x16_f16", 4 * 2.0 * 32 * 32 * 16);
    time(rate_kernel<2>, "4 x f16 + 2 x scaled fp8 (algorithmic 64 k)", 4 * 2.0 * 32 * 32 * 64);
  }
  // ---- 6. scaled conversion ----
  {
    const float xs[4] = {1.0f, -3.0f, 0.5f, 20.0f};
    float* dx; uint32_t* dout;
    CK(hipMalloc(&dx, 16)); CK(hipMalloc(&dout, 4));
    CK(hipMemcpy(dx, xs, 16, hipMemcpyHostToDevice));
    for (float sc : {1.0f, 4.0f, 0.25f}) {
      hipLaunchKernelGGL(cvt_scale_kernel, dim3(1), dim3(1), 0, 0, dx, sc, dout);
      uint32_t w;
      CK(hipMemcpy(&w, dout, 4, hipMemcpyDeviceToHost));
      printf("6. v_cvt_scalef32_pk_fp8_f32(1, -3, 0.5, 20; scale %g) -> %g %g %g %g  (divides by the scale if x / scale, multiplies if x * scale)\n", sc,
             e4m3_decode(w & 0xff), e4m3_decode((w >> 8) & 0xff), e4m3_decode((w >> 16) & 0xff), e4m3_decode(w >> 24));
    }
  }
  printf(bad ? "PROBE: %d assumption(s) FAILED\n" : "PROBE: all assumptions hold\n", bad);
  return bad ? 1 : 0;
}
