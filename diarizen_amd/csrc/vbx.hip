// vbx.hip — the variational-Bayes mixture of VBx on the device (row f1 of SURVEY.md §8f).
//
// Replaces the numpy loop of `VBx` (diarizen/clustering/VBx.py:27-125, the loopProb = 0 branch :99-107 that
// VBxClustering.__call__ exercises, PA/pipelines/clustering.py:677-688): E embeddings in PLDA space (D = 128), K initial
// speakers from the AHC, <= 20 iterations of
//     invL  = 1 / (1 + Fa/Fb * Ns[k] * Phi[d])                      Ns[k]   = sum_e gamma[e,k]
//     alpha = Fa/Fb * invL * Fs[k,d]                                Fs[k,d] = sum_e gamma[e,k] rho[e,d]
//     log_p = Fa * (rho . alpha^T - 0.5 * sum_d (invL + alpha^2) Phi + G[e])
//     gamma = softmax_k(log_p + log pi) ; pi = normalised column sums ; ELBO for the stopping rule
// On the host that is 1.9 s for 60 k embeddings (4 h of audio) in numpy; here the two E-sized passes per iteration are
// kernels over float64 data resident in HBM, the K x D statistics (a few KB) go through the host, which keeps the
// reference's own expressions for invL / alpha / ELBO / the stopping rule (diarizen_amd/clustering.py:vb_gmm drives it).
//   vb_setup_kernel   rho = X * sqrt(Phi), G = -0.5 (|x|^2 + D log 2 pi)                     once
//   vb_accum_kernel   per 256-row chunk: partial Ns / Fs   (then vb_reduce_kernel sums the chunks IN ORDER:
//                     deterministic, no atomics)                                                per iteration
//   vb_estep_kernel   one wavefront per embedding: K dot products of length D, logsumexp as scipy computes it
//                     (max-shifted), gamma written back, per-chunk partial sum of log_px          per iteration
// float64 throughout (the reference is float64); sums over E are chunked, so results agree with numpy's BLAS to
// rounding (tests/test_ops_gpu.py: |d gamma| <= 1e-9, same iteration count, same hard decisions).
#include <math.h>

#include "common.h"

namespace {

constexpr int VB_CHUNK = 256;   // rows per partial-sum chunk
constexpr int VB_KT = 8;        // speakers per register tile of the accumulation kernel

__global__ __launch_bounds__(256) void vb_setup_kernel(const double* __restrict__ X, const double* __restrict__ Phi,
                                                       int E, int D, double* __restrict__ rho, double* __restrict__ G) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int e = blockIdx.x * 4 + wave;
  if (e >= E) return;
  double s = 0.0;
  for (int d = lane; d < D; d += 64) {
    const double x = X[(int64_t)e * D + d];
    rho[(int64_t)e * D + d] = x * sqrt(Phi[d]);
    s += x * x;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) G[e] = -0.5 * (s + D * 1.8378770664093453 /* log(2 pi) */);
}

// partial[c][k][0..D) = sum over the chunk's rows of gamma[e,k] rho[e,d]; partial[c][k][D] = sum gamma[e,k]
__global__ __launch_bounds__(256) void vb_accum_kernel(const double* __restrict__ gamma, const double* __restrict__ rho,
                                                       int E, int D, int K, double* __restrict__ partial) {
  const int c = blockIdx.x;
  const int e0 = c * VB_CHUNK, e1 = min(E, e0 + VB_CHUNK);
  double* out = partial + (int64_t)c * K * (D + 1);
  for (int k0 = 0; k0 < K; k0 += VB_KT) {
    for (int d = threadIdx.x; d <= D; d += 256) {          // column D = the plain sum of gamma
      double acc[VB_KT];
#pragma unroll
      for (int j = 0; j < VB_KT; ++j) acc[j] = 0.0;
      for (int e = e0; e < e1; ++e) {
        const double r = d < D ? rho[(int64_t)e * D + d] : 1.0;
        const double* g = gamma + (int64_t)e * K + k0;
#pragma unroll
        for (int j = 0; j < VB_KT; ++j)
          if (k0 + j < K) acc[j] += g[j] * r;
      }
#pragma unroll
      for (int j = 0; j < VB_KT; ++j)
        if (k0 + j < K) out[(int64_t)(k0 + j) * (D + 1) + d] = acc[j];
    }
  }
}

// stats[i] = sum over chunks (ascending) of partial[c][i]
__global__ __launch_bounds__(256) void vb_reduce_kernel(const double* __restrict__ partial, int nchunk, int n,
                                                        double* __restrict__ stats) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double s = 0.0;
  for (int c = 0; c < nchunk; ++c) s += partial[(int64_t)c * n + i];
  stats[i] = s;
}

// one wavefront per embedding.  ck[k] = 0.5 * sum_d (invL + alpha^2) Phi ; lpi[k] = log(pi + 1e-8)
__global__ __launch_bounds__(256) void vb_estep_kernel(const double* __restrict__ rho, const double* __restrict__ G,
                                                       const double* __restrict__ alpha, const double* __restrict__ ck,
                                                       const double* __restrict__ lpi, double Fa, int E, int D, int K,
                                                       double* __restrict__ gamma, double* __restrict__ lpx) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int e = blockIdx.x * 4 + wave;
  if (e >= E) return;
  const double* r = rho + (int64_t)e * D;
  double* g = gamma + (int64_t)e * K;
  const double ge = G[e];
  // the K values a[k] = log_p + lpi are recomputed in each pass (every lane holds the reduced dot product, so
  // nothing is exchanged through memory between lanes): max, then the sum of exponentials, then the write-back
  auto a_of = [&](int k) {
    const double* a = alpha + (int64_t)k * D;
    double s = 0.0;
    for (int d = lane; d < D; d += 64) s += r[d] * a[d];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    return Fa * (s - ck[k] + ge) + lpi[k];
  };
  double amax = -__builtin_huge_val();
  for (int k = 0; k < K; ++k) {
    const double v = a_of(k);
    amax = v > amax ? v : amax;
  }
  // scipy.special.logsumexp: log(sum exp(a - max)) + max ; gamma = exp(a - log_px)
  double se = 0.0;
  for (int k = 0; k < K; ++k) se += exp(a_of(k) - amax);
  const double l = log(se) + amax;
  for (int k = 0; k < K; ++k) {
    const double v = exp(a_of(k) - l);
    if (lane == 0) g[k] = v;
  }
  if (lane == 0) lpx[e] = l;
}

#define VCHK(call)                                           \
  do {                                                       \
    if ((call) != hipSuccess) { rc = DZN_E_HIP; goto done; } \
  } while (0)

struct VbState {
  int E = 0, D = 0, K = 0, nchunk = 0, device = -1;
  double *rho = nullptr, *G = nullptr, *gamma = nullptr, *partial = nullptr, *stats = nullptr, *alpha = nullptr,
         *ck = nullptr, *lpi = nullptr, *lpx = nullptr, *lpx_part = nullptr;
  // (r5) every array above is carved from ONE block: the host stage's state arena (linkage.hip: no hipMalloc / hipFree per
  // recording, and hipFree is a device-wide synchronisation that a host stage running beside the engine must not take), or, when
  // another state holds the arena, a block of its own.  All work runs on the host stage's stream.
  hipStream_t stream = nullptr;
  char* own_block = nullptr;
  bool leased = false;
};

void vb_free(VbState* s) {
  if (!s) return;
  if (s->stream) (void)hipStreamSynchronize(s->stream);
  if (s->leased) host_state_release(s->device);
  if (s->own_block) (void)hipFree(s->own_block);
  delete s;
}

template <typename T>
T* vb_take(char* base, size_t& off, size_t count) {
  T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
  off += (count * sizeof(T) + 255) & ~(size_t)255;
  return p;
}

}  // namespace

// ---- C ABI (include/dzn.h): an opaque state with the E-sized arrays resident on the device ----
extern "C" int dzn_vbx_create(const double* h_X, const double* h_Phi, const double* h_gamma0, int32_t E, int32_t D,
                              int32_t K, int32_t device, void** out_state) {
  if (!h_X || !h_Phi || !h_gamma0 || !out_state || E < 1 || D < 1 || K < 1) return DZN_E_INVALID;
  int rc = DZN_OK;
  DeviceGuard dg(device);
  if (!dg.ok) return DZN_E_HIP;      // before the state exists: nothing to release
  VbState* s = new VbState();
  double *X = nullptr, *Phi = nullptr;
  s->E = E; s->D = D; s->K = K; s->device = device;
  if (device < 0 && hipGetDevice(&s->device) != hipSuccess) { delete s; *out_state = nullptr; return DZN_E_HIP; }
  s->nchunk = (E + VB_CHUNK - 1) / VB_CHUNK;
  auto carve = [&](char* base) {
    size_t off = 0;
    s->rho = vb_take<double>(base, off, (size_t)E * D);
    s->G = vb_take<double>(base, off, E);
    s->gamma = vb_take<double>(base, off, (size_t)E * K);
    s->partial = vb_take<double>(base, off, (size_t)s->nchunk * K * (D + 1));
    s->stats = vb_take<double>(base, off, (size_t)K * (D + 1));
    s->alpha = vb_take<double>(base, off, (size_t)K * D);
    s->ck = vb_take<double>(base, off, K);
    s->lpi = vb_take<double>(base, off, K);
    s->lpx = vb_take<double>(base, off, E);
    s->lpx_part = vb_take<double>(base, off, s->nchunk);
    X = vb_take<double>(base, off, (size_t)E * D);        // setup only
    Phi = vb_take<double>(base, off, D);
    return off;
  };
  const size_t bytes = carve(nullptr);
  char* base = nullptr;
  rc = host_state_lease(s->device, bytes, &s->stream, &base);
  if (rc != DZN_OK) { delete s; *out_state = nullptr; return rc; }
  if (base) {
    s->leased = true;
  } else {                                                 // the arena belongs to another live state
    if (hipMalloc(&s->own_block, bytes) != hipSuccess) { (void)hipGetLastError(); rc = DZN_E_NOMEM; goto done; }
    base = s->own_block;
  }
  carve(base);
  VCHK(hipMemcpyAsync(X, h_X, (size_t)E * D * 8, hipMemcpyHostToDevice, s->stream));
  VCHK(hipMemcpyAsync(Phi, h_Phi, (size_t)D * 8, hipMemcpyHostToDevice, s->stream));
  VCHK(hipMemcpyAsync(s->gamma, h_gamma0, (size_t)E * K * 8, hipMemcpyHostToDevice, s->stream));
  hipLaunchKernelGGL(vb_setup_kernel, dim3((E + 3) / 4), dim3(256), 0, s->stream, X, Phi, E, D, s->rho, s->G);
  VCHK(hipGetLastError());
  VCHK(hipStreamSynchronize(s->stream));
done:
  if (rc != DZN_OK) { vb_free(s); s = nullptr; }
  *out_state = s;
  return rc;
}

// h_stats [K][D+1] <- (Fs[k, 0..D), Ns[k]) of the CURRENT responsibilities (gamma^T rho and the column sums of gamma)
extern "C" int dzn_vbx_stats(void* state, double* h_stats) {
  VbState* s = static_cast<VbState*>(state);
  if (!s || !h_stats) return DZN_E_INVALID;
  int rc = DZN_OK;
  const int n = s->K * (s->D + 1);
  DeviceGuard dg(s->device);
  if (!dg.ok) return DZN_E_HIP;
  hipLaunchKernelGGL(vb_accum_kernel, dim3(s->nchunk), dim3(256), 0, s->stream, s->gamma, s->rho, s->E, s->D, s->K, s->partial);
  hipLaunchKernelGGL(vb_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, s->stream, s->partial, s->nchunk, n, s->stats);
  VCHK(hipGetLastError());
  VCHK(hipMemcpyAsync(h_stats, s->stats, (size_t)n * 8, hipMemcpyDeviceToHost, s->stream));
  VCHK(hipStreamSynchronize(s->stream));
done:
  return rc;
}

// one E-step with the host-computed alpha [K][D], ck [K] = 0.5 sum_d (invL + alpha^2) Phi, lpi [K] = log(pi + 1e-8):
// gamma <- softmax_k(Fa (rho.alpha^T - ck + G) + lpi) ; *h_total = sum_e logsumexp (the data term of the ELBO)
extern "C" int dzn_vbx_estep(void* state, const double* h_alpha, const double* h_ck, const double* h_lpi, double Fa,
                             double* h_total) {
  VbState* s = static_cast<VbState*>(state);
  if (!s || !h_alpha || !h_ck || !h_lpi || !h_total) return DZN_E_INVALID;
  int rc = DZN_OK;
  double total = 0.0;
  DeviceGuard dg(s->device);
  if (!dg.ok) return DZN_E_HIP;
  VCHK(hipMemcpyAsync(s->alpha, h_alpha, (size_t)s->K * s->D * 8, hipMemcpyHostToDevice, s->stream));
  VCHK(hipMemcpyAsync(s->ck, h_ck, (size_t)s->K * 8, hipMemcpyHostToDevice, s->stream));
  VCHK(hipMemcpyAsync(s->lpi, h_lpi, (size_t)s->K * 8, hipMemcpyHostToDevice, s->stream));
  hipLaunchKernelGGL(vb_estep_kernel, dim3((s->E + 3) / 4), dim3(256), 0, s->stream, s->rho, s->G, s->alpha, s->ck, s->lpi, Fa,
                     s->E, s->D, s->K, s->gamma, s->lpx);
  // sum of log_px: chunk partials (vb_accum with K = 1, D = 0 reads "gamma" = lpx and sums it), then in-order reduce
  hipLaunchKernelGGL(vb_accum_kernel, dim3(s->nchunk), dim3(256), 0, s->stream, s->lpx, s->rho, s->E, 0, 1, s->lpx_part);
  hipLaunchKernelGGL(vb_reduce_kernel, dim3(1), dim3(256), 0, s->stream, s->lpx_part, s->nchunk, 1, s->stats);
  VCHK(hipGetLastError());
  VCHK(hipMemcpyAsync(&total, s->stats, 8, hipMemcpyDeviceToHost, s->stream));
  VCHK(hipStreamSynchronize(s->stream));
  *h_total = total;
done:
  return rc;
}

extern "C" int dzn_vbx_gamma(void* state, double* h_gamma) {
  VbState* s = static_cast<VbState*>(state);
  if (!s || !h_gamma) return DZN_E_INVALID;
  DeviceGuard dg(s->device);
  if (!dg.ok) return DZN_E_HIP;
  if (hipMemcpyAsync(h_gamma, s->gamma, (size_t)s->E * s->K * 8, hipMemcpyDeviceToHost, s->stream) != hipSuccess) return DZN_E_HIP;
  return hipStreamSynchronize(s->stream) == hipSuccess ? DZN_OK : DZN_E_HIP;
}

extern "C" int dzn_vbx_destroy(void* state) {
  if (!state) return DZN_OK;
  DeviceGuard dg(static_cast<VbState*>(state)->device);
  vb_free(static_cast<VbState*>(state));
  return DZN_OK;
}
