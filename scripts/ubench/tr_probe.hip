#include <hip/hip_runtime.h>
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const _Float16* in, _Float16* out) {
  __shared__ __attribute__((aligned(16))) _Float16 s[64 * 16];
  for (int i = threadIdx.x; i < 1024; i += 64) s[i] = in[i];
  __syncthreads();
  const int lane = threadIdx.x, lr = lane & 15, g = lane >> 4;
  const _Float16* p = s + (4 * g + (lr >> 2)) * 16 + (lr & 3) * 4;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  f16x4 f = __builtin_bit_cast(f16x4, v);
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = f[j];
}
int main() {
  _Float16 h[1024], o[256];
  for (int i = 0; i < 1024; ++i) h[i] = (_Float16)(float)i;   // value = row*16 + col
  _Float16 *di, *dout;
  hipMalloc(&di, sizeof(h)); hipMalloc(&dout, sizeof(o));
  hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
  hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane)
    for (int j = 0; j < 4; ++j) {
      int want = (4 * (lane >> 4) + j) * 16 + (lane & 15);
      if ((int)(float)o[lane * 4 + j] != want) { if (bad < 8) printf("lane %d j %d got %d want %d\n", lane, j, (int)(float)o[lane*4+j], want); ++bad; }
    }
  printf("tr_probe mismatches: %d\n", bad);
  return 0;
}
