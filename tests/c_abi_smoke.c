/*
 * c_abi_smoke.c — drives libdzn_hip.so through include/dzn.h from plain C (gcc -std=c11, no Python, no torch, no C++):
 *
 *     dzn_create -> dzn_load_tensor (every state_dict entry) -> dzn_finalize_weights -> dzn_num_frames
 *       -> dzn_segment_forward on a caller-owned device buffer -> compare with the expected log-probs
 *       -> error paths (bad batch size must fail with a message) -> dzn_destroy
 *
 * This is the binding a reference-side maintainer would write in any FFI: the boundary is a C ABI,
 * there are no torch types in it.  Input: a blob written by tests/test_properties_gpu.py::test_c_abi_without_python
 * (little endian):
 *     "DZNBLOB1"  dzn_config (sizeof as in this header)
 *     int32 n_tensors, then per tensor: int32 key_len, key bytes, int32 dtype, int32 ndim, int64 shape[ndim], data
 *     int32 B, int32 N, float wave[B*N], int32 L, int32 n_classes, float expected_logp[B*L*n_classes], float tol
 * Exit code 0 = parity within tol; prints max |d logp|.   Built by diarizen_amd/build.py (build()).
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dzn.h"

#define CHECK(x, what)                                                              \
  do {                                                                              \
    int rc_ = (x);                                                                  \
    if (rc_ != 0) {                                                                 \
      fprintf(stderr, "%s failed: %d (%s)\n", what, rc_, dzn_last_error(h));        \
      return 2;                                                                     \
    }                                                                               \
  } while (0)

static int rd(FILE* f, void* p, size_t n) { return fread(p, 1, n, f) == n ? 0 : -1; }

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s blob\n", argv[0]);
    return 64;
  }
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 65;
  char magic[8];
  dzn_config cfg;
  dzn_handle* h = NULL;
  if (rd(f, magic, 8) || memcmp(magic, "DZNBLOB1", 8) || rd(f, &cfg, sizeof cfg)) return 66;
  if (cfg.struct_size != (int32_t)sizeof cfg) {
    fprintf(stderr, "dzn_config size mismatch: blob %d, header %zu\n", cfg.struct_size, sizeof cfg);
    return 67;
  }
  CHECK(dzn_create(&cfg, &h), "dzn_create");
  int32_t nt = 0;
  if (rd(f, &nt, 4)) return 66;
  for (int32_t t = 0; t < nt; ++t) {
    int32_t klen, dtype, ndim;
    char key[512];
    int64_t shape[8], numel = 1;
    if (rd(f, &klen, 4) || klen >= (int32_t)sizeof key || rd(f, key, (size_t)klen)) return 66;
    key[klen] = 0;
    if (rd(f, &dtype, 4) || rd(f, &ndim, 4) || ndim > 8 || rd(f, shape, 8 * (size_t)ndim)) return 66;
    for (int i = 0; i < ndim; ++i) numel *= shape[i];
    const size_t esz = dtype == DZN_F32 ? 4 : 8;
    void* buf = malloc((size_t)numel * esz + 8);
    if (!buf || rd(f, buf, (size_t)numel * esz)) return 66;
    CHECK(dzn_load_tensor(h, key, buf, shape, ndim, dtype), key);
    free(buf);
  }
  CHECK(dzn_finalize_weights(h), "dzn_finalize_weights");
  int32_t B, N, L, NC;
  float tol;
  if (rd(f, &B, 4) || rd(f, &N, 4)) return 66;
  float* wave = (float*)malloc(sizeof(float) * (size_t)B * N);
  if (rd(f, wave, sizeof(float) * (size_t)B * N) || rd(f, &L, 4) || rd(f, &NC, 4)) return 66;
  float* expect = (float*)malloc(sizeof(float) * (size_t)B * L * NC);
  if (rd(f, expect, sizeof(float) * (size_t)B * L * NC) || rd(f, &tol, 4)) return 66;
  fclose(f);
  if (dzn_num_frames(h, N) != L) {
    fprintf(stderr, "dzn_num_frames(%d) = %d, expected %d\n", N, dzn_num_frames(h, N), L);
    return 3;
  }
  /* caller-owned device buffers and stream (the ownership rule of dzn.h) */
  float *d_wave = NULL, *d_logp = NULL;
  uint8_t* d_ml = NULL;
  hipStream_t st;
  if (hipStreamCreate(&st) != hipSuccess || hipMalloc((void**)&d_wave, sizeof(float) * (size_t)B * N) != hipSuccess ||
      hipMalloc((void**)&d_logp, sizeof(float) * (size_t)B * L * NC) != hipSuccess ||
      hipMalloc((void**)&d_ml, (size_t)B * L * cfg.max_speakers_per_chunk) != hipSuccess)
    return 4;
  hipMemcpyAsync(d_wave, wave, sizeof(float) * (size_t)B * N, hipMemcpyHostToDevice, st);
  CHECK(dzn_segment_forward(h, d_wave, B, N, d_logp, d_ml, st), "dzn_segment_forward");
  float* got = (float*)malloc(sizeof(float) * (size_t)B * L * NC);
  uint8_t* ml = (uint8_t*)malloc((size_t)B * L * cfg.max_speakers_per_chunk);
  hipMemcpyAsync(got, d_logp, sizeof(float) * (size_t)B * L * NC, hipMemcpyDeviceToHost, st);
  hipMemcpyAsync(ml, d_ml, (size_t)B * L * cfg.max_speakers_per_chunk, hipMemcpyDeviceToHost, st);
  if (hipStreamSynchronize(st) != hipSuccess) return 5;
  double err = 0.0;
  long flips = 0;
  for (long i = 0; i < (long)B * L; ++i) {
    int a0 = 0, a1 = 0;
    for (int c = 0; c < NC; ++c) {
      const double d = fabs((double)got[i * NC + c] - (double)expect[i * NC + c]);
      if (d > err) err = d;
      if (got[i * NC + c] > got[i * NC + a0]) a0 = c;
      if (expect[i * NC + c] > expect[i * NC + a1]) a1 = c;
    }
    flips += a0 != a1;
  }
  printf("c_abi_smoke: %s  B=%d N=%d L=%d  max|dlogp|=%.3e (tol %.1e)  argmax flips=%ld  workspace=%lld bytes\n",
         dzn_version(), B, N, L, err, (double)tol, flips, (long long)dzn_workspace_bytes(h));
  /* error behaviour: B > max_batch is refused with a message, the handle stays usable */
  if (dzn_segment_forward(h, d_wave, cfg.max_batch + 1, N, d_logp, d_ml, st) == 0 || !dzn_last_error(h)[0]) return 6;
  if (dzn_load_tensor(h, "late", wave, NULL, 0, DZN_F32) != DZN_E_STATE) return 7;
  hipFree(d_wave);
  hipFree(d_logp);
  hipFree(d_ml);
  hipStreamDestroy(st);
  CHECK(dzn_destroy(h), "dzn_destroy");
  return (err <= tol && flips == 0) ? 0 : 1;
}
