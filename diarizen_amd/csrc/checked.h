// checked.h — the CHECKED build (python -m diarizen_amd.build --checked  ->  lib/libdzn_hip_checked.so, -DDZN_CHECKED).
//
// ~60 hand-scheduled kernels with raw s_waitcnt / s_barrier pipelines, LDS-DMA fills and XOR-swizzled LDS images: an index that
// leaves its LDS stage, a tracker read past its array (r4 found one by accident: a B = 1 conv launch indexed its |max| tracker by
// m / L) or a tile that walks off the descriptor does not fault on this hardware — it reads whatever lies there.  In a checked
// build the kernels carry DZN_CHECK(cond, id, detail) at those index computations.  A failed check does NOT trap (a trap kills
// the context and every later test with it): it counts itself in a per-translation-unit device word — the first failure also
// records its id, workgroup and a detail value — and the run goes on.  dzn_checked_status() (engine.cpp) sums the words of all
// translation units; tests/conftest.py fails the session when it is non-zero (scripts/run_checked.sh ->
// profiles/r5_checked_build.log).  Release builds compile the macro to nothing.
//
// Device symbols are per translation unit (the library is not built with relocatable device code): every .hip file that uses
// DZN_CHECK instantiates its word with DZN_CHECKED_TU(name), which also defines the host-side collector that engine.cpp calls.
#pragma once
#include <hip/hip_runtime.h>

#ifdef DZN_CHECKED
#define DZN_CHECKED_TU(name)                                                                                     \
  namespace { __device__ unsigned int dzn_check_word_[4] = {0u, 0u, 0u, 0u}; }                                    \
  extern "C" int dzn_checked_collect_##name(unsigned int* out4, int reset) {                                     \
    unsigned int z[4] = {0u, 0u, 0u, 0u};                                                                         \
    if (hipDeviceSynchronize() != hipSuccess) return -3;                                                          \
    if (hipMemcpyFromSymbol(out4, HIP_SYMBOL(dzn_check_word_), sizeof(z)) != hipSuccess) return -3;               \
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(dzn_check_word_), z, sizeof(z)) != hipSuccess) return -3;           \
    return 0;                                                                                                     \
  }
#define DZN_CHECK(cond, id, detail)                                                                              \
  do {                                                                                                            \
    if (!(cond)) {                                                                                                \
      if (atomicAdd(&dzn_check_word_[0], 1u) == 0u) {                                                             \
        dzn_check_word_[1] = (unsigned)(id);                                                                      \
        dzn_check_word_[2] = blockIdx.x;                                                                          \
        dzn_check_word_[3] = (unsigned)(detail);                                                                  \
      }                                                                                                           \
    }                                                                                                             \
  } while (0)
#else
#define DZN_CHECKED_TU(name)
#define DZN_CHECK(cond, id, detail) do { } while (0)
#endif

// check ids: 0x1xx gemm_split, 0x2xx gemm_mx, 0x3xx gemm_split_pre, 0x4xx resblock_fused, 0x5xx resblock_ws,
//            0x60x attention_split, 0x61x attention_planes, 0x7xx frontend_fused
