// linkage.hip — centroid-linkage agglomerative clustering with the distance matrix resident in HBM
// (row f1 of SURVEY.md §8f: host clustering scale-out).
//
// Replaces, for large inputs, the call
//     scipy.cluster.hierarchy.linkage(embeddings, method="centroid", metric="euclidean")
// of AgglomerativeClustering.cluster (PA/pipelines/clustering.py:407-416) and of the AHC
// initialisation of VBxClustering (PA/pipelines/clustering.py:656-658).  scipy forms the
// condensed n(n-1)/2 float64 distance matrix on the host and runs the generic O(n^2)..O(n^3)
// nearest-neighbour algorithm single-threaded: 17 s at 2 h of audio (36 k embeddings), minutes and
// 20 GB at 4 h — longer than the whole device stage on 8 GPUs.  Here:
//   * D [n][n] float64 lives in HBM (41 GB at n = 72 k; 288 GB available), filled by a tiled
//     pairwise-distance kernel (inactive columns / the diagonal hold +inf, so a row scan is a bare min);
//   * the same greedy algorithm as scipy's fast_linkage runs on the device: per row a LOWER BOUND
//     lb[z] of its minimum with a candidate neighbour nb[z]; each merge = one single-workgroup
//     kernel (global argmin of lb, verify against D, rescan the row if the bound was stale, emit the
//     dendrogram row) + one wide kernel (Lance-Williams centroid update of row / column `hi`,
//     written in scipy's operation order in float64, bounds refreshed);
//   * the global argmin of lb runs over per-block minima (256 rows per block, kept current by the update kernel), and
//     the bound of the row a merge creates arrives exact from the update kernel's partial minima of that row — the
//     rescans of rows whose bound went stale (0.6 per merge) keep 8 loads in flight per thread and mask retired
//     columns by size[] (retired columns are not blanked: that was a second scattered write per row and merge);
//   * no host round trip inside the loop: 2(n-1) launches are queued back to back (replaying them from a hipGraph
//     was measured: no gain, the loop is not launch-bound).
// With no exact ties in the data the merge sequence — hence the dendrogram Z and every flat
// clustering cut from it — equals scipy's (tests/test_ops_gpu.py compares Z and fcluster output).
#include <math.h>

#include <vector>

#include "common.h"

namespace {

struct MergeState {
  int lo, hi, nlo, nhi, k;
  double dist;
};

constexpr double DINF = __builtin_huge_val();

// ---- pairwise euclidean distances, float64 accumulation: 64 x 64 tile per workgroup, j-tile >= i-tile ----
__global__ __launch_bounds__(256) void pdist_kernel(const float* __restrict__ E, int n, int dim,
                                                    double* __restrict__ D) {
  __shared__ float sa[64][17], sb[64][17];
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (bj < bi) return;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;  // thread -> rows ty*4.., cols tx*4..
  double acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
  for (int k0 = 0; k0 < dim; k0 += 16) {
    for (int idx = threadIdx.x; idx < 64 * 16; idx += 256) {
      const int r = idx >> 4, c = idx & 15;
      const int gi = bi * 64 + r, gj = bj * 64 + r, gk = k0 + c;
      sa[r][c] = (gi < n && gk < dim) ? E[(int64_t)gi * dim + gk] : 0.f;
      sb[r][c] = (gj < n && gk < dim) ? E[(int64_t)gj * dim + gk] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      double av[4], bv[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) av[a] = (double)sa[ty * 4 + a][c];
#pragma unroll
      for (int b = 0; b < 4; ++b) bv[b] = (double)sb[tx * 4 + b][c];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const double df = av[a] - bv[b];
          acc[a][b] += df * df;
        }
    }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int i = bi * 64 + ty * 4 + a, j = bj * 64 + tx * 4 + b;
      if (i < n && j < n) {
        const double d = i == j ? DINF : sqrt(acc[a][b]);
        D[(int64_t)i * n + j] = d;
        D[(int64_t)j * n + i] = d;
      }
    }
}

constexpr int SCAN_U = 8;   // loads in flight per thread in the row scans

// block-wide minimum of per-thread (value, index) pairs, lowest index on ties; result valid in every thread
__device__ __forceinline__ void block_min_pair(double v, int id, double& val, int& idx, double* sval, int* sidx) {
  // wave reduction
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double ov = __shfl_xor(v, o, 64);
    const int oi = __shfl_xor(id, o, 64);
    if (ov < v || (ov == v && oi < id)) { v = ov; id = oi; }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  __syncthreads();  // previous users of sval / sidx are done
  if (lane == 0) { sval[wave] = v; sidx[wave] = id; }
  __syncthreads();
  v = sval[0];
  id = sidx[0];
  for (int w = 1; w < nw; ++w) {
    const double ov = sval[w];
    const int oi = sidx[w];
    if (ov < v || (ov == v && oi < id)) { v = ov; id = oi; }
  }
  val = v;
  idx = id;
}

// block-wide argmin (lowest index on ties) of p[0..n); result valid in every thread
__device__ __forceinline__ void block_argmin(const double* __restrict__ p, int n, double& val, int& idx,
                                             double* sval, int* sidx) {
  double v = DINF;
  int id = 0x7fffffff;
  // SCAN_U independent loads in flight per thread: one load per iteration left the single workgroup of the
  // selection kernel waiting a full memory latency per 8 KB of the row
  for (int i0 = threadIdx.x; i0 < n; i0 += SCAN_U * blockDim.x) {
    double xs[SCAN_U];
#pragma unroll
    for (int u = 0; u < SCAN_U; ++u) {
      const int i = i0 + u * blockDim.x;
      xs[u] = i < n ? p[i] : DINF;
    }
#pragma unroll
    for (int u = 0; u < SCAN_U; ++u)
      if (xs[u] < v) { v = xs[u]; id = i0 + u * blockDim.x; }   // ascending i per thread: first occurrence kept
  }
  block_min_pair(v, id, val, idx, sval, sidx);
}

// the same over the ACTIVE columns only (size[i] != 0): retired clusters keep their last distances in D — blanking
// column `lo` in every row was a second scattered 8-byte write per row and merge, and those writes (one DRAM page
// each) are what a merge costs at n >= 30 k
__device__ __forceinline__ void block_argmin_active(const double* __restrict__ p, const int* __restrict__ size, int n,
                                                    double& val, int& idx, double* sval, int* sidx) {
  double v = DINF;
  int id = 0x7fffffff;
  for (int i0 = threadIdx.x; i0 < n; i0 += SCAN_U * blockDim.x) {
    double xs[SCAN_U];
    int sz[SCAN_U];
#pragma unroll
    for (int u = 0; u < SCAN_U; ++u) {
      const int i = i0 + u * blockDim.x;
      xs[u] = i < n ? p[i] : DINF;
      sz[u] = i < n ? size[i] : 0;
    }
#pragma unroll
    for (int u = 0; u < SCAN_U; ++u)
      if (sz[u] != 0 && xs[u] < v) { v = xs[u]; id = i0 + u * blockDim.x; }
  }
  block_min_pair(v, id, val, idx, sval, sidx);
}

// minima of the bounds per block of LB_BLK rows: the global argmin of lb then scans n / LB_BLK values instead of n
constexpr int LB_BLK = 256;
__global__ __launch_bounds__(LB_BLK) void block_minima_kernel(const double* __restrict__ lb, int n,
                                                              double* __restrict__ bmin, int* __restrict__ barg) {
  __shared__ double sval[4];
  __shared__ int sidx[4];
  const int z = blockIdx.x * LB_BLK + threadIdx.x;
  double v;
  int id;
  block_min_pair(z < n ? lb[z] : DINF, z < n ? z : 0x7fffffff, v, id, sval, sidx);
  if (threadIdx.x == 0) { bmin[blockIdx.x] = v; barg[blockIdx.x] = id; }
}

// initial bounds: one workgroup per row
__global__ __launch_bounds__(256) void init_rows_kernel(const double* __restrict__ D, int n, double* __restrict__ lb,
                                                        int* __restrict__ nb) {
  __shared__ double sval[4];
  __shared__ int sidx[4];
  const int z = blockIdx.x;
  double v;
  int id;
  block_argmin(D + (int64_t)z * n, n, v, id, sval, sidx);
  if (threadIdx.x == 0) { lb[z] = v; nb[z] = id; }
}

// one merge, part 1 (single workgroup): find the closest pair, emit the dendrogram row.
// The global argmin of the bounds runs over the per-block minima (bmin / barg, kept current by update_kernel and
// refreshed here for the one block whose row was rescanned); the bound of the row merged LAST (hi of merge k - 1)
// arrives exact from update_kernel's partial minima of the new row (hp_val / hp_idx) instead of being rescanned.
__global__ __launch_bounds__(1024) void select_kernel(double* __restrict__ D, int n, double* __restrict__ lb,
                                                      int* __restrict__ nb, int* __restrict__ size,
                                                      int* __restrict__ cid, double* __restrict__ Z,
                                                      MergeState* __restrict__ st, double* __restrict__ bmin,
                                                      int* __restrict__ barg, const double* __restrict__ hp_val,
                                                      const int* __restrict__ hp_idx) {
  __shared__ double sval[16];
  __shared__ int sidx[16];
  const int k = st->k;
  const int nblk = (n + LB_BLK - 1) / LB_BLK;
  auto refresh_block = [&](int row) {     // recompute the minimum of the block holding `row` (all threads call)
    const int b = row / LB_BLK, z = b * LB_BLK + threadIdx.x;
    double v;
    int id;
    const bool in = threadIdx.x < LB_BLK && z < n;
    block_min_pair(in ? lb[z] : DINF, in ? z : 0x7fffffff, v, id, sval, sidx);
    if (threadIdx.x == 0) { bmin[b] = v; barg[b] = id; }
    __threadfence_block();
    __syncthreads();
  };
  if (k > 0) {   // exact bound of the row created by the previous merge
    const int hp = st->hi;
    double v;
    int id;
    double tv = DINF;
    int ti = 0x7fffffff;
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) {
      const double x = hp_val[i];
      const int xi = hp_idx[i];
      if (x < tv || (x == tv && xi < ti)) { tv = x; ti = xi; }
    }
    block_min_pair(tv, ti, v, id, sval, sidx);
    __syncthreads();
    if (threadIdx.x == 0) { lb[hp] = v; nb[hp] = v < DINF ? id : -1; }
    __threadfence_block();
    __syncthreads();
    refresh_block(hp);
  }
  int x, y;
  double d;
  for (int guard = 0; guard <= n; ++guard) {
    {   // argmin over the block minima (ascending block index per thread, lowest row index on ties)
      double tv = DINF;
      int ti = 0x7fffffff;
      for (int i = threadIdx.x; i < nblk; i += blockDim.x) {
        const double v = bmin[i];
        const int id = barg[i];
        if (v < tv || (v == tv && id < ti)) { tv = v; ti = id; }
      }
      block_min_pair(tv, ti, d, x, sval, sidx);
    }
    y = nb[x];
    const bool exact = y >= 0 && D[(int64_t)x * n + y] == d;
    if (exact) break;
    double rv;
    int ri;
    block_argmin_active(D + (int64_t)x * n, size, n, rv, ri, sval, sidx);   // stale bound: rescan row x
    __syncthreads();
    if (threadIdx.x == 0) { lb[x] = rv; nb[x] = ri; }
    __threadfence_block();
    __syncthreads();
    refresh_block(x);
  }
  if (threadIdx.x == 0) {
    const int lo = x < y ? x : y, hi = x < y ? y : x;
    const int nlo = size[lo], nhi = size[hi];
    const int ia = cid[lo], ib = cid[hi];
    Z[4 * k + 0] = (double)(ia < ib ? ia : ib);
    Z[4 * k + 1] = (double)(ia < ib ? ib : ia);
    Z[4 * k + 2] = d;
    Z[4 * k + 3] = (double)(nlo + nhi);
    size[lo] = 0;            // cluster lo is dropped ...
    size[hi] = nlo + nhi;    // ... cluster hi becomes the union
    cid[hi] = n + k;
    lb[lo] = DINF;
    lb[hi] = DINF;           // set exactly by the next select_kernel from update_kernel's partial minima
    nb[hi] = -1;
    st->lo = lo; st->hi = hi; st->nlo = nlo; st->nhi = nhi; st->dist = d; st->k = k + 1;
  }
}

// one merge, part 2: Lance-Williams centroid update of row / column hi, column lo retired; every block leaves the
// minimum of its slice of the NEW row hi (hp_val / hp_idx) and the minimum of its rows' bounds (bmin / barg)
__global__ __launch_bounds__(LB_BLK) void update_kernel(double* __restrict__ D, int n, double* __restrict__ lb,
                                                        int* __restrict__ nb, const int* __restrict__ size,
                                                        const MergeState* __restrict__ st, double* __restrict__ bmin,
                                                        int* __restrict__ barg, double* __restrict__ hp_val,
                                                        int* __restrict__ hp_idx) {
  __shared__ double sval[4];
  __shared__ int sidx[4];
  const int z = blockIdx.x * LB_BLK + threadIdx.x;
  const int lo = st->lo, hi = st->hi;
  double nd = DINF;          // D[hi][z] after the merge (inf for inactive z, z == hi, z >= n)
  double myb = DINF;         // lb[z] after the merge
  if (z < n) {
    const int nz = size[z];
    if (z == hi) {
      D[(int64_t)hi * n + lo] = DINF;
    } else if (nz != 0) {    // active (z == lo has size 0)
      const int sx = st->nlo, sy = st->nhi;
      const double dxy = st->dist;
      const double dxi = D[(int64_t)lo * n + z], dyi = D[(int64_t)hi * n + z];
      // scipy _hierarchy_distance_update.pxi, _centroid(d_xi, d_yi, d_xy, size_x, size_y, size_i), same order
      nd = sqrt((((sx * dxi * dxi) + (sy * dyi * dyi)) - (sx * sy * dxy * dxy) / (sx + sy)) / (sx + sy));
      D[(int64_t)hi * n + z] = nd;
      D[(int64_t)z * n + hi] = nd;             // column lo is NOT blanked: readers of a row mask by size[] instead
      if (nb[z] == lo) nb[z] = hi;             // a guess; lb[z] stays a valid lower bound
      if (nd < lb[z]) { lb[z] = nd; nb[z] = hi; }
    }
    myb = lb[z];             // rows lo / hi hold +inf here (select_kernel); hi gets its exact bound next
  }
  double v;
  int id;
  block_min_pair(nd, z < n ? z : 0x7fffffff, v, id, sval, sidx);
  if (threadIdx.x == 0) { hp_val[blockIdx.x] = v; hp_idx[blockIdx.x] = id; }
  block_min_pair(myb, z < n ? z : 0x7fffffff, v, id, sval, sidx);
  if (threadIdx.x == 0) { bmin[blockIdx.x] = v; barg[blockIdx.x] = id; }
}

#define LCHK(call)                                   \
  do {                                               \
    if ((call) != hipSuccess) { rc = DZN_E_HIP; goto done; } \
  } while (0)

}  // namespace

extern "C" int dzn_linkage_centroid(const float* h_emb, int32_t n, int32_t dim, double* h_Z, int32_t device) {
  if (!h_emb || !h_Z || n < 2 || dim < 1) return DZN_E_INVALID;
  int rc = DZN_OK;
  float* E = nullptr;
  double *D = nullptr, *lb = nullptr, *Z = nullptr, *bmin = nullptr, *hp_val = nullptr;
  int *nb = nullptr, *size = nullptr, *cid = nullptr, *barg = nullptr, *hp_idx = nullptr;
  const int nblk = (n + LB_BLK - 1) / LB_BLK;
  MergeState* st = nullptr;
  std::vector<int> ones(n, 1), ids(n);
  for (int i = 0; i < n; ++i) ids[i] = i;
  MergeState st0{};
  DeviceGuard dg(device);      // restores the caller's device on every return path
  if (!dg.ok) return DZN_E_HIP;
  if (hipMalloc(&D, (size_t)n * n * sizeof(double)) != hipSuccess) { rc = DZN_E_NOMEM; goto done; }
  LCHK(hipMalloc(&E, (size_t)n * dim * sizeof(float)));
  LCHK(hipMalloc(&lb, (size_t)n * sizeof(double)));
  LCHK(hipMalloc(&Z, (size_t)(n - 1) * 4 * sizeof(double)));
  LCHK(hipMalloc(&nb, (size_t)n * sizeof(int)));
  LCHK(hipMalloc(&size, (size_t)n * sizeof(int)));
  LCHK(hipMalloc(&cid, (size_t)n * sizeof(int)));
  LCHK(hipMalloc(&st, sizeof(MergeState)));
  LCHK(hipMalloc(&bmin, (size_t)nblk * sizeof(double)));
  LCHK(hipMalloc(&hp_val, (size_t)nblk * sizeof(double)));
  LCHK(hipMalloc(&barg, (size_t)nblk * sizeof(int)));
  LCHK(hipMalloc(&hp_idx, (size_t)nblk * sizeof(int)));
  LCHK(hipMemcpy(E, h_emb, (size_t)n * dim * sizeof(float), hipMemcpyHostToDevice));
  LCHK(hipMemcpy(size, ones.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice));
  LCHK(hipMemcpy(cid, ids.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice));
  LCHK(hipMemcpy(st, &st0, sizeof(MergeState), hipMemcpyHostToDevice));
  {
    const int tiles = (n + 63) / 64;
    hipLaunchKernelGGL(pdist_kernel, dim3(tiles, tiles), dim3(256), 0, 0, E, n, dim, D);
    hipLaunchKernelGGL(init_rows_kernel, dim3(n), dim3(256), 0, 0, D, n, lb, nb);
    hipLaunchKernelGGL(block_minima_kernel, dim3(nblk), dim3(LB_BLK), 0, 0, lb, n, bmin, barg);
    for (int k = 0; k < n - 1; ++k) {
      hipLaunchKernelGGL(select_kernel, dim3(1), dim3(1024), 0, 0, D, n, lb, nb, size, cid, Z, st, bmin, barg, hp_val,
                         hp_idx);
      hipLaunchKernelGGL(update_kernel, dim3(nblk), dim3(LB_BLK), 0, 0, D, n, lb, nb, size, st, bmin, barg, hp_val,
                         hp_idx);
    }
  }
  LCHK(hipGetLastError());
  LCHK(hipMemcpy(h_Z, Z, (size_t)(n - 1) * 4 * sizeof(double), hipMemcpyDeviceToHost));
done:
  (void)hipFree(D); (void)hipFree(E); (void)hipFree(lb); (void)hipFree(Z);
  (void)hipFree(nb); (void)hipFree(size); (void)hipFree(cid); (void)hipFree(st);
  (void)hipFree(bmin); (void)hipFree(hp_val); (void)hipFree(barg); (void)hipFree(hp_idx);
  return rc;
}

// ---- cosine distances of every embedding to the cluster centroids (row f1: `cdist` of the assignment step) -------
// scipy.spatial.distance.cdist(E, Cn, metric="cosine") of BaseClustering.assign_embeddings
// (PA/pipelines/clustering.py:207-216): float64, row norms first, then per pair
//     d = 1 - clip(dot(u, v) / (|u| |v|))
// The kernels below use the same formula in float64 with in-order sums and separate product / sum roundings
// (contraction off), one thread per (row[, centroid]) walking the 256 dimensions, so identical rows give identical
// scores (ties stay ties).  scipy's own summation order is its build's: measured agreement 2e-15 on distances of
// order 1 (tests/test_ops_gpu.py), i.e. the level at which scipy differs from a plain in-order loop.  72 k
// embeddings x 13 centroids at 4 h of audio is 0.24 G flop in float64 — the point is the 0.3-0.7 s scipy spends
// converting and scanning 147 MB on one host core.
namespace {

__global__ __launch_bounds__(256) void row_norm_f32_kernel(const float* __restrict__ E, int n, int dim,
                                                           double* __restrict__ nrm) {
#pragma clang fp contract(off)
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  const float* e = E + (int64_t)r * dim;
  double s = 0.0;
  for (int i = 0; i < dim; ++i) {
    const double c = (double)e[i];
    const double p = c * c;
    s = s + p;
  }
  nrm[r] = sqrt(s);
}

__global__ __launch_bounds__(256) void row_norm_f64_kernel(const double* __restrict__ X, int n, int dim,
                                                           double* __restrict__ nrm) {
#pragma clang fp contract(off)
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  const double* x = X + (int64_t)r * dim;
  double s = 0.0;
  for (int i = 0; i < dim; ++i) {
    const double p = x[i] * x[i];
    s = s + p;
  }
  nrm[r] = sqrt(s);
}

// thread -> (row, centroid) with the centroid fastest: the lanes of a wavefront share a handful of rows
__global__ __launch_bounds__(256) void cdist_cosine_kernel(const float* __restrict__ E, const double* __restrict__ Cn,
                                                           const double* __restrict__ ne, const double* __restrict__ nc,
                                                           int n, int dim, int k, double* __restrict__ out) {
#pragma clang fp contract(off)
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)n * k) return;
  const int r = (int)(idx / k), c = (int)(idx - (int64_t)r * k);
  const float* e = E + (int64_t)r * dim;
  const double* v = Cn + (int64_t)c * dim;
  double s = 0.0;
  for (int i = 0; i < dim; ++i) {
    const double p = (double)e[i] * v[i];
    s = s + p;
  }
  const double den = ne[r] * nc[c];
  double cosine = s / den;
  if (fabs(cosine) > 1.0) cosine = copysign(1.0, cosine);      // scipy clips the rounding error
  out[idx] = 1.0 - cosine;
}

}  // namespace

extern "C" int dzn_cdist_cosine(const float* h_emb, int32_t n, int32_t dim, const double* h_cent, int32_t k,
                                double* h_dist, int32_t device) {
  if (!h_emb || !h_cent || !h_dist || n < 1 || dim < 1 || k < 1) return DZN_E_INVALID;
  int rc = DZN_OK;
  float* E = nullptr;
  double *Cn = nullptr, *ne = nullptr, *nc = nullptr, *D = nullptr;
  DeviceGuard dg(device);      // restores the caller's device on every return path
  if (!dg.ok) return DZN_E_HIP;
  if (hipMalloc(&E, (size_t)n * dim * sizeof(float)) != hipSuccess) { rc = DZN_E_NOMEM; goto done; }
  LCHK(hipMalloc(&Cn, (size_t)k * dim * sizeof(double)));
  LCHK(hipMalloc(&ne, (size_t)n * sizeof(double)));
  LCHK(hipMalloc(&nc, (size_t)k * sizeof(double)));
  LCHK(hipMalloc(&D, (size_t)n * k * sizeof(double)));
  LCHK(hipMemcpy(E, h_emb, (size_t)n * dim * sizeof(float), hipMemcpyHostToDevice));
  LCHK(hipMemcpy(Cn, h_cent, (size_t)k * dim * sizeof(double), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(row_norm_f32_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, E, n, dim, ne);
  hipLaunchKernelGGL(row_norm_f64_kernel, dim3((k + 255) / 256), dim3(256), 0, 0, Cn, k, dim, nc);
  hipLaunchKernelGGL(cdist_cosine_kernel, dim3((unsigned)(((int64_t)n * k + 255) / 256)), dim3(256), 0, 0, E, Cn, ne, nc,
                     n, dim, k, D);
  LCHK(hipGetLastError());
  LCHK(hipMemcpy(h_dist, D, (size_t)n * k * sizeof(double), hipMemcpyDeviceToHost));
done:
  (void)hipFree(E); (void)hipFree(Cn); (void)hipFree(ne); (void)hipFree(nc); (void)hipFree(D);
  return rc;
}
