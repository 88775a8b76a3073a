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
//     lb[z] of its minimum with a candidate neighbour nb[z]; the closest pair is the row with the smallest bound once
//     that bound is exact, a stale bound is refreshed by rescanning its row; a merge is the Lance-Williams centroid
//     update of row / column `hi`, written in scipy's operation order in float64;
//   * r3: ONE launch per step (step_kernel, below) — every workgroup selects redundantly from per-workgroup records,
//     then does its slice of the merge or of the row rescan; 6.2 us per launch, 718 ms at n = 35 790 with 2.2 rescans per
//     merge (profiles/r3_linkage_step_vs_two_kernel.txt).  The r2 loop — a single-workgroup selection kernel + a wide
//     update kernel per merge, 2253 ms on the same input — stays behind DZN_LINKAGE_TWO_KERNEL=1 as the cross-check of
//     tests/test_ops_gpu.py;
//   * no host round trip inside a batch of launches; retired columns are masked by size[], not blanked.
// With no exact ties in the data the merge sequence — hence the dendrogram Z and every flat
// clustering cut from it — equals scipy's (tests/test_ops_gpu.py compares Z and fcluster output).
#include <math.h>
#include <stdio.h>

#include <chrono>
#include <mutex>
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


// ---- r3: ONE launch per step --------------------------------------------------------------------------------------
// The two-kernel loop above spends its time in the single workgroup of select_kernel (the dependent global round trips
// of the bound fix-up, the block refresh, the argmin, the D look-up and — 0.6 times per merge — a 288 KB row rescan by
// ONE workgroup) plus two launch gaps: ~30 us per merge at any n.  step_kernel folds the selection into the wide
// kernel: every workgroup (256 rows / columns each) redundantly reduces the SAME published state — one record per
// workgroup: the minimum bound of its rows with that row's candidate neighbour and whether the bound is exact
// (D[x][nb[x]] == lb[x], checked by the owner when it publishes) — so all of them reach the same decision without
// talking to each other, then each performs its slice of the step:
//   * MERGE  (bound exact): the Lance-Williams update of its columns (as update_kernel), its partial minimum of the
//     new row, its new record;
//   * RESCAN (bound stale): its slice of the row scan — the scan that one workgroup did alone is spread over the chip —
//     and its record with that row left out.
// What a step publishes (records, partial minima, the step descriptor) is double-buffered by launch parity: a launch
// only READS what the previous launch wrote and only WRITES the other copy, so the kernel boundary is the only
// synchronisation.  State with a single owner (lb / nb / size / cid of a row, the columns of D) is fixed up by the
// owner's thread at the start of the NEXT launch (the bound of the merged row from the partial minima, the sizes of
// the merged pair); other workgroups substitute the previous step's values instead of reading those words.
// Launches: (n - 1) merges + ~0.6 (n - 1) rescans instead of 2 (n - 1), each without the serial section.
enum { STEP_NONE = 0, STEP_MERGE = 1, STEP_RESCAN = 2, STEP_DONE = 3, STEP_FAIL = 4 };
struct StepState {
  int kind, lo, hi, nlo, nhi, k, x, pad;
  double dist;
};
struct StepRec {      // per workgroup: minimum bound of its rows
  double v;
  int x, y, exact, pad;
};
struct StepPart {     // per workgroup: minimum of its slice of a row (the merged row / the rescanned row)
  double v;
  int idx, pad;
};

// two block-wide (value, index) minima in one pass (shared barriers); results valid in every thread
__device__ __forceinline__ void block_min_pair2(double v0, int i0, double v1, int i1, double& o0, int& oi0, double& o1,
                                                int& oi1, double* sval, int* sidx) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double a = __shfl_xor(v0, o, 64);
    const int ai = __shfl_xor(i0, o, 64);
    const double b = __shfl_xor(v1, o, 64);
    const int bi = __shfl_xor(i1, o, 64);
    if (a < v0 || (a == v0 && ai < i0)) { v0 = a; i0 = ai; }
    if (b < v1 || (b == v1 && bi < i1)) { v1 = b; i1 = bi; }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;   // nw <= 4: sval / sidx hold 2 x 4
  __syncthreads();
  if (lane == 0) { sval[wave] = v0; sidx[wave] = i0; sval[4 + wave] = v1; sidx[4 + wave] = i1; }
  __syncthreads();
  v0 = sval[0]; i0 = sidx[0]; v1 = sval[4]; i1 = sidx[4];
  for (int w = 1; w < nw; ++w) {
    const double a = sval[w], b = sval[4 + w];
    const int ai = sidx[w], bi = sidx[4 + w];
    if (a < v0 || (a == v0 && ai < i0)) { v0 = a; i0 = ai; }
    if (b < v1 || (b == v1 && bi < i1)) { v1 = b; i1 = bi; }
  }
  o0 = v0; oi0 = i0; o1 = v1; oi1 = i1;
}

// first records: bounds from init_rows_kernel are exact
__global__ __launch_bounds__(LB_BLK) void init_rec_kernel(int n, const double* __restrict__ lb, const int* __restrict__ nb,
                                                          int* __restrict__ exf, StepRec* __restrict__ rec) {
  __shared__ double sval[4];
  __shared__ int sidx[4];
  const int z = blockIdx.x * LB_BLK + threadIdx.x;
  if (z < n) exf[z] = 1;
  double v;
  int x;
  block_min_pair(z < n ? lb[z] : DINF, z < n ? z : 0x7fffffff, v, x, sval, sidx);
  if (z == x) {
    StepRec r;
    r.v = v; r.x = x; r.y = nb[x]; r.exact = r.y >= 0; r.pad = 0;
    rec[blockIdx.x] = r;
  }
}

// A launch is latency-bound (a chain of global round trips and block reductions on ~141 workgroups), so the kernel is
// written around the chain: (1) ONE round trip fetches everything that does not depend on the decision — the step
// descriptor, the published records and partial minima, this thread's own row words (lb, nb, size, exact flag);
// (2) one double reduction gives the fixed-up row's bound and the winner; (3) one round trip reads the two rows of D
// (or the slice of the rescanned row); (4) one double reduction publishes the partial minimum and the record.  Whether a
// bound is exact is tracked per row (exf[z], maintained by the row's own thread: the bound is exact until the distance to
// the remembered neighbour changes without undercutting it) instead of looked up in D, which was a third round trip.
__global__ __launch_bounds__(LB_BLK) void step_kernel(double* __restrict__ D, int n, double* __restrict__ lb,
                                                      int* __restrict__ nb, int* __restrict__ size,
                                                      int* __restrict__ cid, int* __restrict__ exf,
                                                      double* __restrict__ Z, StepState* __restrict__ st2,
                                                      StepRec* __restrict__ rec2, StepPart* __restrict__ part2,
                                                      int parity) {
  __shared__ double sval[8];
  __shared__ int sidx[8];
  __shared__ int sy[2];
  const int G = gridDim.x, w = blockIdx.x, t = threadIdx.x;
  const int z = w * LB_BLK + t;
  const bool in = z < n;
  const StepRec* rec = rec2 + (int64_t)parity * G;
  StepRec* rec_out = rec2 + (int64_t)(parity ^ 1) * G;
  const StepPart* part = part2 + (int64_t)parity * G;
  StepPart* part_out = part2 + (int64_t)(parity ^ 1) * G;
  StepState* st_out = st2 + (parity ^ 1);
  // ---- (1) everything that does not depend on the decision ----
  const StepState prev = st2[parity];
  StepRec r0;
  r0.v = DINF; r0.x = 0x7fffffff; r0.y = -1; r0.exact = 0;
  StepPart p0;
  p0.v = DINF; p0.idx = 0x7fffffff;
  if (t < G) { r0 = rec[t]; p0 = part[t]; }
  double my_lb = in ? lb[z] : DINF;
  int my_nb = in ? nb[z] : -1;
  int my_sz = in ? size[z] : 0;
  int my_ex = in ? exf[z] : 0;
  if (prev.kind == STEP_DONE || prev.kind == STEP_FAIL) {   // surplus launch: keep both copies of the descriptor final
    if (w == 0 && t == 0) *st_out = prev;
    return;
  }
  const bool owed = prev.kind == STEP_MERGE || prev.kind == STEP_RESCAN;
  // ---- (2) the bound owed by the previous step + the winner over the records ----
  double tv = owed ? p0.v : DINF, uv = r0.v;
  int ti = owed ? p0.idx : 0x7fffffff, ui = r0.x;
  for (int i = t + LB_BLK; i < G; i += LB_BLK) {            // n > 65 536 only
    const StepRec r = rec[i];
    const StepPart p = part[i];
    if (owed && (p.v < tv || (p.v == tv && p.idx < ti))) { tv = p.v; ti = p.idx; }
    if (r.v < uv || (r.v == uv && r.x < ui)) { uv = r.v; ui = r.x; }
  }
  double ev, d;
  int eid, x;
  block_min_pair2(tv, ti, uv, ui, ev, eid, d, x, sval, sidx);
  const int ex = !owed ? 0x7fffffff : prev.kind == STEP_MERGE ? prev.hi : prev.x;   // the row left out of the records
  const int ey = ev < DINF ? eid : -1;
  bool lb_dirty = false, sz_dirty = false;
  if (owed && z == ex) { my_lb = ev; my_nb = ey; my_ex = ey >= 0; lb_dirty = true; }
  if (prev.kind == STEP_MERGE) {
    if (z == prev.hi) { my_sz = prev.nlo + prev.nhi; cid[z] = n + prev.k - 1; sz_dirty = true; }
    if (z == prev.lo) { my_sz = 0; sz_dirty = true; }
  }
  const int k = prev.k;
  int y;
  bool exact;
  const bool extra_wins = owed && (ev < d || (ev == d && ex < x));
  if (extra_wins) {
    d = ev; x = ex; y = ey; exact = ey >= 0;
  } else {
    __syncthreads();
    if (t < G && r0.x == x && r0.v == d) { sy[0] = r0.y; sy[1] = r0.exact; }
    for (int i = t + LB_BLK; i < G; i += LB_BLK)
      if (rec[i].x == x && rec[i].v == d) { sy[0] = rec[i].y; sy[1] = rec[i].exact; }
    __syncthreads();
    y = sy[0];
    exact = sy[1] != 0 && y >= 0;
  }
  if (!(d < DINF) || x < 0 || x >= n) {    // NaN / inf distances: no pair left to merge
    if (w == 0 && t == 0) { StepState s = prev; s.kind = STEP_FAIL; *st_out = s; }
    return;
  }
  double pv = DINF;       // this thread's element of the row whose minimum the step publishes
  int excl = -1;          // row left out of this step's records
  if (!exact) {
    // ---- RESCAN: this workgroup's slice of row x over the active columns ----
    if (w == 0 && t == 0) {
      StepState s = prev;
      s.kind = STEP_RESCAN; s.x = x; s.k = k; s.pad = prev.pad + 1;   // pad counts the rescans (diagnostics)
      *st_out = s;
    }
    if (in && my_sz != 0) pv = D[(int64_t)x * n + z];
    excl = x;
  } else {
    // ---- MERGE: Lance-Williams update of this workgroup's columns ----
    const int lo = x < y ? x : y, hi = x < y ? y : x;
    const bool sub = prev.kind == STEP_MERGE;    // words of the previous pair are being rewritten by their owners
    const int nlo = sub && lo == prev.hi ? prev.nlo + prev.nhi : size[lo];   // (lo / hi are active: never prev.lo)
    const int nhi = sub && hi == prev.hi ? prev.nlo + prev.nhi : size[hi];
    if (w == 0 && t == 0) {
      const int ia = sub && lo == prev.hi ? n + prev.k - 1 : cid[lo];
      const int ib = sub && hi == prev.hi ? n + prev.k - 1 : cid[hi];
      Z[4 * k + 0] = (double)(ia < ib ? ia : ib);
      Z[4 * k + 1] = (double)(ia < ib ? ib : ia);
      Z[4 * k + 2] = d;
      Z[4 * k + 3] = (double)(nlo + nhi);
      StepState s;
      s.kind = k + 1 >= n - 1 ? STEP_DONE : STEP_MERGE;
      s.lo = lo; s.hi = hi; s.nlo = nlo; s.nhi = nhi; s.k = k + 1; s.x = -1; s.pad = prev.pad; s.dist = d;
      *st_out = s;
    }
    if (in) {
      if (z == hi) {
        D[(int64_t)hi * n + lo] = DINF;
        my_lb = DINF; my_nb = -1; my_ex = 0; lb_dirty = true;   // exact bound at the start of the next launch
      } else if (z == lo) {
        my_lb = DINF; my_ex = 0; lb_dirty = true;               // retired
      } else if (my_sz != 0) {
        const double dxi = D[(int64_t)lo * n + z], dyi = D[(int64_t)hi * n + z];
        // scipy _hierarchy_distance_update.pxi, _centroid(d_xi, d_yi, d_xy, size_x, size_y, size_i), same order
        const double nd = sqrt((((nlo * dxi * dxi) + (nhi * dyi * dyi)) - (nlo * nhi * d * d) / (nlo + nhi)) / (nlo + nhi));
        D[(int64_t)hi * n + z] = nd;
        D[(int64_t)z * n + hi] = nd;             // column lo is NOT blanked: readers of a row mask by size[]
        if (my_nb == lo) { my_nb = hi; my_ex = nd == my_lb; lb_dirty = true; }      // a guess; lb stays a lower bound
        else if (my_nb == hi) { my_ex = nd == my_lb; lb_dirty = true; }             // the neighbour's distance moved
        if (nd < my_lb) { my_lb = nd; my_nb = hi; my_ex = 1; lb_dirty = true; }
        pv = nd;
      }
    }
  }
  // ---- (4) publish: partial minimum of the new / rescanned row, record of this workgroup's rows ----
  double bv, rv;
  int bi, rx;
  block_min_pair2(pv, in ? z : 0x7fffffff, (in && z != excl) ? my_lb : DINF, in ? z : 0x7fffffff, bv, bi, rv, rx, sval, sidx);
  if (t == 0) {
    StepPart p;
    p.v = bv; p.idx = bi; p.pad = 0;
    part_out[w] = p;
  }
  if (z == rx) {          // rx is always a row of this workgroup (every thread contributes its own index)
    StepRec r;
    r.v = rv; r.x = rx; r.y = rv < DINF ? my_nb : -1; r.exact = rv < DINF && my_ex != 0 && my_nb >= 0; r.pad = 0;
    rec_out[w] = r;
  }
  if (in && lb_dirty) { lb[z] = my_lb; nb[z] = my_nb; exf[z] = my_ex; }
  if (in && sz_dirty) size[z] = my_sz;
}

// ---- r6: the step loop with TWO remembered neighbours per row ----------------------------------------------------------
// In step_kernel's loop 1.9-2.2 of every 3 launches are RESCANS: a merge retires the remembered neighbour of every row that
// pointed at `lo` or `hi`, those rows are the likeliest next winners, and each of them costs a whole launch (a row scan spread
// over the chip) before it can merge.  step2_kernel keeps the TWO nearest neighbours of every row:
//   I1: l1 <= D[z][c] for every active c != z;          e1: D[z][n1] == l1 (the exact minimum)
//   I2: l2 <= D[z][c] for every active c not in {z, n1}; e2: D[z][n2] == l2 (the exact second minimum, given e1)
// When a merge retires n1, the row's new minimum is min(new distance to the union, l2) and it is EXACT whenever e2 held (or the
// new distance undercuts the bound): the row merges without a rescan; the second slot then only keeps its bound (e2 = 0) until
// the next scan of the row refills both.  A numpy model of the rules is checked against scipy on the CPU
// (tests/test_design_math.py::test_linkage_top2_model_equals_scipy); on the 4 h recording's 20 888 embeddings the loop needs
// 1.1-1.3 launches per merge instead of 2.9 (profiles/r6_linkage_top2.txt).  Same structure otherwise: one launch per step,
// every workgroup reduces the same published records, state double-buffered by launch parity, single-owner words fixed up by
// the owner at the start of the next launch.  The partial results of a row scan are top-2 pairs.
struct StepPart2 {    // per workgroup: the two smallest of its slice of a row (the merged row / the rescanned row)
  double v1, v2;
  int i1, i2;
};

__device__ __forceinline__ bool pair_lt(double a, int ai, double b, int bi) { return a < b || (a == b && ai < bi); }

// (v1, i1, v2, i2) <- the two smallest of {v1, v2, b1, b2} by (value, index); both inputs sorted, their elements distinct.
// Written as selects on values: the branchy form (assignments through references) was lowered to indexed SCRATCH stores -
// 64 B per lane of private memory and a memory round trip per merge, 10.8 us per launch where this form has 6.x.
__device__ __forceinline__ void top2_merge(double& v1, int& i1, double& v2, int& i2, double b1, int bi1, double b2, int bi2) {
  const double a1 = v1, a2 = v2;
  const int ai1 = i1, ai2 = i2;
  const bool bf = pair_lt(b1, bi1, a1, ai1);              // b's first is the overall first
  const double c1 = bf ? a1 : b1, c2 = bf ? b2 : a2;      // the two candidates for the second place
  const int ci1 = bf ? ai1 : bi1, ci2 = bf ? bi2 : ai2;
  const bool cs = pair_lt(c1, ci1, c2, ci2);
  v1 = bf ? b1 : a1;
  i1 = bf ? bi1 : ai1;
  v2 = cs ? c1 : c2;
  i2 = cs ? ci1 : ci2;
}

// Cross-lane moves of the wave reductions below as DPP controls (VALU data path, a few cycles) instead of `__shfl_xor`, which
// hipcc lowers to ds_bpermute_b32 (a round trip through the LDS crossbar per dword and stage: 108 of them chained in twelve
// dependent stages were ~1 us of every launch of a latency-bound kernel).  quad_perm [1,0,3,2] / [2,3,0,1], row_ror:4 / :8 give
// every lane of a 16-lane row the row's result (each stage joins DISJOINT lane sets, which top2_merge requires); row_bcast:15
// (into rows 1, 3) and row_bcast:31 (into rows 2, 3) carry it across rows - lanes outside the row mask receive `idle`, the
// identity of the reduction - and lane 63 ends up with the wave's result, read back with v_readlane.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i(int idle, int v) {
  return __builtin_amdgcn_update_dpp(idle, v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_d(double idle, double v) {
  const int lo = dpp_i<CTRL, ROW_MASK>(__double2loint(idle), __double2loint(v));
  const int hi = dpp_i<CTRL, ROW_MASK>(__double2hiint(idle), __double2hiint(v));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane63_d(double v) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void top2_min_stage(double& a1, int& ai1, double& a2, int& ai2, double& m, int& mi) {
  const double b1 = dpp_d<CTRL, ROW_MASK>(DINF, a1), b2 = dpp_d<CTRL, ROW_MASK>(DINF, a2), bm = dpp_d<CTRL, ROW_MASK>(DINF, m);
  const int bi1 = dpp_i<CTRL, ROW_MASK>(0x7fffffff, ai1), bi2 = dpp_i<CTRL, ROW_MASK>(0x7fffffff, ai2),
            bmi = dpp_i<CTRL, ROW_MASK>(0x7fffffff, mi);
  top2_merge(a1, ai1, a2, ai2, b1, bi1, b2, bi2);
  if (pair_lt(bm, bmi, m, mi)) { m = bm; mi = bmi; }
}

// block-wide: the top-2 of per-thread sorted pairs (a1, ai1, a2, ai2) AND the minimum of per-thread (m, mi); valid in every
// thread.  sval: 12 doubles, sidx: 12 ints.
__device__ __forceinline__ void block_top2_min(double a1, int ai1, double a2, int ai2, double m, int mi, double& o1, int& oi1,
                                               double& o2, int& oi2, double& om, int& omi, double* sval, int* sidx) {
  top2_min_stage<0xB1, 0xf>(a1, ai1, a2, ai2, m, mi);      // quad_perm [1,0,3,2]
  top2_min_stage<0x4E, 0xf>(a1, ai1, a2, ai2, m, mi);      // quad_perm [2,3,0,1]
  top2_min_stage<0x124, 0xf>(a1, ai1, a2, ai2, m, mi);     // row_ror:4
  top2_min_stage<0x128, 0xf>(a1, ai1, a2, ai2, m, mi);     // row_ror:8
  top2_min_stage<0x142, 0xa>(a1, ai1, a2, ai2, m, mi);     // row_bcast:15 -> rows 1, 3
  top2_min_stage<0x143, 0xc>(a1, ai1, a2, ai2, m, mi);     // row_bcast:31 -> rows 2, 3
  a1 = readlane63_d(a1); a2 = readlane63_d(a2); m = readlane63_d(m);
  ai1 = __builtin_amdgcn_readlane(ai1, 63); ai2 = __builtin_amdgcn_readlane(ai2, 63); mi = __builtin_amdgcn_readlane(mi, 63);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;   // nw <= 4
  __syncthreads();
  if (lane == 0) {
    sval[wave] = a1; sval[4 + wave] = a2; sval[8 + wave] = m;
    sidx[wave] = ai1; sidx[4 + wave] = ai2; sidx[8 + wave] = mi;
  }
  __syncthreads();
  a1 = sval[0]; a2 = sval[4]; m = sval[8];
  ai1 = sidx[0]; ai2 = sidx[4]; mi = sidx[8];
  for (int w = 1; w < nw; ++w) {
    top2_merge(a1, ai1, a2, ai2, sval[w], sidx[w], sval[4 + w], sidx[4 + w]);
    if (pair_lt(sval[8 + w], sidx[8 + w], m, mi)) { m = sval[8 + w]; mi = sidx[8 + w]; }
  }
  o1 = a1; oi1 = ai1; o2 = a2; oi2 = ai2; om = m; omi = mi;
}

// initial state: the exact two nearest neighbours of every row (one workgroup per row) — the diagonal holds +inf
__global__ __launch_bounds__(256) void init_rows2_kernel(const double* __restrict__ D, int n, double* __restrict__ l1a,
                                                         double* __restrict__ l2a, int* __restrict__ n1a,
                                                         int* __restrict__ n2a, int* __restrict__ fla) {
  __shared__ double sval[12];
  __shared__ int sidx[12];
  const int z = blockIdx.x;
  const double* p = D + (int64_t)z * n;
  double v1 = DINF, v2 = DINF;
  int i1 = 0x7fffffff, i2 = 0x7fffffff;
  for (int i0 = threadIdx.x; i0 < n; i0 += SCAN_U * blockDim.x) {
    double xs[SCAN_U];
#pragma unroll
    for (int u = 0; u < SCAN_U; ++u) {
      const int i = i0 + u * blockDim.x;
      xs[u] = i < n ? p[i] : DINF;
    }
#pragma unroll
    for (int u = 0; u < SCAN_U; ++u) {
      const int i = i0 + u * blockDim.x;     // ascending per thread: strict comparisons keep the first occurrence
      if (xs[u] < v1) { v2 = v1; i2 = i1; v1 = xs[u]; i1 = i; }
      else if (xs[u] < v2) { v2 = xs[u]; i2 = i; }
    }
  }
  double o1, o2, om;
  int oi1, oi2, omi;
  block_top2_min(v1, i1, v2, i2, DINF, 0x7fffffff, o1, oi1, o2, oi2, om, omi, sval, sidx);
  if (threadIdx.x == 0) {
    const bool h1 = o1 < DINF, h2 = o2 < DINF;
    l1a[z] = o1; l2a[z] = o2; n1a[z] = h1 ? oi1 : -1; n2a[z] = h2 ? oi2 : -1; fla[z] = (h1 ? 1 : 0) | (h2 ? 2 : 0);
  }
}

__global__ __launch_bounds__(LB_BLK) void init_rec2_kernel(int n, const double* __restrict__ l1a, const int* __restrict__ n1a,
                                                           StepRec* __restrict__ rec) {
  __shared__ double sval[4];
  __shared__ int sidx[4];
  const int z = blockIdx.x * LB_BLK + threadIdx.x;
  double v;
  int x;
  block_min_pair(z < n ? l1a[z] : DINF, z < n ? z : 0x7fffffff, v, x, sval, sidx);
  if (z == x) {
    StepRec r;
    r.v = v; r.x = x; r.y = n1a[x]; r.exact = r.y >= 0; r.pad = 0;
    rec[blockIdx.x] = r;
  }
}

__global__ __launch_bounds__(LB_BLK) void step2_kernel(double* __restrict__ D, int n, double* __restrict__ l1a,
                                                       double* __restrict__ l2a, int* __restrict__ n1a, int* __restrict__ n2a,
                                                       int* __restrict__ fla, int* __restrict__ size, int* __restrict__ cid,
                                                       double* __restrict__ Z, StepState* __restrict__ st2,
                                                       StepRec* __restrict__ rec2, StepPart2* __restrict__ part2, int parity) {
  __shared__ double sval[12];
  __shared__ int sidx[12];
  __shared__ int sy[2];
  const int G = gridDim.x, w = blockIdx.x, t = threadIdx.x;
  const int z = w * LB_BLK + t;
  const bool in = z < n;
  const StepRec* rec = rec2 + (int64_t)parity * G;
  StepRec* rec_out = rec2 + (int64_t)(parity ^ 1) * G;
  const StepPart2* part = part2 + (int64_t)parity * G;
  StepPart2* part_out = part2 + (int64_t)(parity ^ 1) * G;
  StepState* st_out = st2 + (parity ^ 1);
  // ---- (1) everything that does not depend on the decision: ONE round trip ----
  const StepState prev = st2[parity];
  StepRec r0;
  r0.v = DINF; r0.x = 0x7fffffff; r0.y = -1; r0.exact = 0;
  StepPart2 p0;
  p0.v1 = p0.v2 = DINF; p0.i1 = p0.i2 = 0x7fffffff;
  if (t < G) { r0 = rec[t]; p0 = part[t]; }
  double my_l1 = in ? l1a[z] : DINF, my_l2 = in ? l2a[z] : DINF;
  int my_n1 = in ? n1a[z] : -1, my_n2 = in ? n2a[z] : -1, my_fl = in ? fla[z] : 0;
  int my_sz = in ? size[z] : 0;
  if (prev.kind == STEP_DONE || prev.kind == STEP_FAIL) {   // surplus launch: keep both copies of the descriptor final
    if (w == 0 && t == 0) *st_out = prev;
    return;
  }
  const bool owed = prev.kind == STEP_MERGE || prev.kind == STEP_RESCAN;
  // ---- (2) the two neighbours owed to the row of the previous step + the winner over the records ----
  double a1 = owed ? p0.v1 : DINF, a2 = owed ? p0.v2 : DINF, uv = r0.v;
  int ai1 = owed ? p0.i1 : 0x7fffffff, ai2 = owed ? p0.i2 : 0x7fffffff, ui = r0.x;
  for (int i = t + LB_BLK; i < G; i += LB_BLK) {            // n > 65 536 only
    const StepRec r = rec[i];
    const StepPart2 p = part[i];
    if (owed) top2_merge(a1, ai1, a2, ai2, p.v1, p.i1, p.v2, p.i2);
    if (pair_lt(r.v, r.x, uv, ui)) { uv = r.v; ui = r.x; }
  }
  double ev1, ev2, d;
  int eid1, eid2, x;
  block_top2_min(a1, ai1, a2, ai2, uv, ui, ev1, eid1, ev2, eid2, d, x, sval, sidx);
  const int ex = !owed ? 0x7fffffff : prev.kind == STEP_MERGE ? prev.hi : prev.x;   // the row left out of the records
  const int ey = ev1 < DINF ? eid1 : -1;
  bool dirty = false, sz_dirty = false;
  if (owed && z == ex) {
    my_l1 = ev1; my_n1 = ey; my_l2 = ev2; my_n2 = ev2 < DINF ? eid2 : -1;
    my_fl = (my_n1 >= 0 ? 1 : 0) | (my_n2 >= 0 ? 2 : 0);
    dirty = true;
  }
  if (prev.kind == STEP_MERGE) {
    if (z == prev.hi) { my_sz = prev.nlo + prev.nhi; cid[z] = n + prev.k - 1; sz_dirty = true; }
    if (z == prev.lo) { my_sz = 0; sz_dirty = true; }
  }
  const int k = prev.k;
  int y;
  bool exact;
  const bool extra_wins = owed && pair_lt(ev1, ex, d, x);
  if (extra_wins) {
    d = ev1; x = ex; y = ey; exact = ey >= 0;
  } else {
    __syncthreads();
    if (t < G && r0.x == x && r0.v == d) { sy[0] = r0.y; sy[1] = r0.exact; }
    for (int i = t + LB_BLK; i < G; i += LB_BLK)
      if (rec[i].x == x && rec[i].v == d) { sy[0] = rec[i].y; sy[1] = rec[i].exact; }
    __syncthreads();
    y = sy[0];
    exact = sy[1] != 0 && y >= 0;
  }
  if (!(d < DINF) || x < 0 || x >= n) {    // NaN / inf distances: no pair left to merge
    if (w == 0 && t == 0) { StepState s = prev; s.kind = STEP_FAIL; *st_out = s; }
    return;
  }
  double pv = DINF;       // this thread's element of the row whose two smallest the step publishes
  int excl = -1;          // row left out of this step's records
  if (!exact) {
    // ---- RESCAN: this workgroup's slice of row x over the active columns (D[x][x] is +inf) ----
    if (w == 0 && t == 0) {
      StepState s = prev;
      s.kind = STEP_RESCAN; s.x = x; s.k = k; s.pad = prev.pad + 1;   // pad counts the rescans (diagnostics)
      *st_out = s;
    }
    if (in && my_sz != 0) pv = D[(int64_t)x * n + z];
    excl = x;
  } else {
    // ---- MERGE: Lance-Williams update of this workgroup's columns ----
    const int lo = x < y ? x : y, hi = x < y ? y : x;
    const bool sub = prev.kind == STEP_MERGE;    // words of the previous pair are being rewritten by their owners
    const int nlo = sub && lo == prev.hi ? prev.nlo + prev.nhi : size[lo];   // (lo / hi are active: never prev.lo)
    const int nhi = sub && hi == prev.hi ? prev.nlo + prev.nhi : size[hi];
    if (w == 0 && t == 0) {
      const int ia = sub && lo == prev.hi ? n + prev.k - 1 : cid[lo];
      const int ib = sub && hi == prev.hi ? n + prev.k - 1 : cid[hi];
      Z[4 * k + 0] = (double)(ia < ib ? ia : ib);
      Z[4 * k + 1] = (double)(ia < ib ? ib : ia);
      Z[4 * k + 2] = d;
      Z[4 * k + 3] = (double)(nlo + nhi);
      StepState s;
      s.kind = k + 1 >= n - 1 ? STEP_DONE : STEP_MERGE;
      s.lo = lo; s.hi = hi; s.nlo = nlo; s.nhi = nhi; s.k = k + 1; s.x = -1; s.pad = prev.pad; s.dist = d;
      *st_out = s;
    }
    if (in) {
      if (z == hi) {
        D[(int64_t)hi * n + lo] = DINF;
        my_l1 = my_l2 = DINF; my_n1 = my_n2 = -1; my_fl = 0; dirty = true;   // both neighbours at the start of the next launch
      } else if (z == lo) {
        my_l1 = my_l2 = DINF; my_fl = 0; dirty = true;                       // retired
      } else if (my_sz != 0) {
        const double dxi = D[(int64_t)lo * n + z], dyi = D[(int64_t)hi * n + z];
        // scipy _hierarchy_distance_update.pxi, _centroid(d_xi, d_yi, d_xy, size_x, size_y, size_i), same order
        const double v = sqrt((((nlo * dxi * dxi) + (nhi * dyi * dyi)) - (nlo * nhi * d * d) / (nlo + nhi)) / (nlo + nhi));
        D[(int64_t)hi * n + z] = v;
        D[(int64_t)z * n + hi] = v;              // column lo is NOT blanked: readers of a row mask by size[]
        pv = v;
        // columns lo and (old) hi leave the row, column hi re-enters with v: the rules of the header (I1 / I2 / e1 / e2)
        const bool d1 = my_n1 == lo || my_n1 == hi, d2 = my_n2 == lo || my_n2 == hi;
        const bool e1 = my_fl & 1, e2 = my_fl & 2;
        if (!d1) {
          if (v < my_l1) {                       // the union is the new nearest: exact whatever l1 was
            my_l2 = my_l1; my_n2 = my_n1; my_fl = 1 | (e1 ? 2 : 0);
            my_l1 = v; my_n1 = hi; dirty = true;
          } else if (!d2) {
            if (v < my_l2) { my_l2 = v; my_n2 = hi; my_fl |= 2; dirty = true; }
          } else if (v <= my_l2) {               // second slot retired, everything else is >= l2 >= v
            my_l2 = v; my_n2 = hi; my_fl |= 2; dirty = true;
          } else {
            my_n2 = hi; my_fl &= ~2; dirty = true;       // l2 stays a lower bound of the rest
          }
        } else {
          dirty = true;
          if (!d2 && e2) {                       // nearest retired, the exact second takes over unless the union undercuts it
            if (v < my_l2) { my_l1 = v; my_n1 = hi; my_fl = 3; }
            else { my_l1 = my_l2; my_n1 = my_n2; my_n2 = hi; my_fl = 1; }
          } else {                               // only bounds left: exact iff the union is under the bound of the rest
            if (v <= my_l2) { my_l1 = v; my_n1 = hi; my_fl = 1; }
            else { my_l1 = my_l2; my_n1 = hi; my_fl = 0; }
            if (d2) my_n2 = hi;
          }
        }
      }
    }
  }
  // ---- (4) publish: the two smallest of the new / rescanned row's slice, the record of this workgroup's rows ----
  double b1, b2, rv;
  int bi1, bi2, rx;
  block_top2_min(pv, in ? z : 0x7fffffff, DINF, 0x7fffffff, (in && z != excl) ? my_l1 : DINF, in ? z : 0x7fffffff, b1, bi1, b2,
                 bi2, rv, rx, sval, sidx);
  if (t == 0) {
    StepPart2 p;
    p.v1 = b1; p.v2 = b2; p.i1 = bi1; p.i2 = bi2;
    part_out[w] = p;
  }
  if (z == rx) {          // rx is always a row of this workgroup (every thread contributes its own index)
    StepRec r;
    r.v = rv; r.x = rx; r.y = rv < DINF ? my_n1 : -1; r.exact = rv < DINF && (my_fl & 1) != 0 && my_n1 >= 0; r.pad = 0;
    rec_out[w] = r;
  }
  if (in && dirty) { l1a[z] = my_l1; l2a[z] = my_l2; n1a[z] = my_n1; n2a[z] = my_n2; fla[z] = my_fl; }
  if (in && sz_dirty) size[z] = my_sz;
}

// ---- r6b: the whole loop in ONE launch (persistent workgroups) ----------------------------------------------------------
// step2_kernel's loop pays a kernel boundary per step: 8.5 us per launch back to back, of which the step's own dependent chain
// (two memory round trips and two block reductions) is about half.  persist2_kernel keeps the G workgroups of one launch
// resident for the whole dendrogram and replaces the boundary by what the step needs anyway — every workgroup reading every
// workgroup's published record:
//   * what a workgroup publishes per step (its record and its top-2 slice of the new / rescanned row: 10 payload dwords) goes
//     out as 10 64-bit words, each {step number : payload dword}.  A 64-bit store is one transaction, so a reader that finds
//     the current step number in all 10 words of a slot has that step's payload — the poll of the slots IS the grid barrier
//     (no counter, no second round trip to fetch the data behind a flag).  Slots are double-buffered by step parity: a
//     workgroup can only reach step s + 2's publish after it saw every other workgroup's step s + 1 words, which those wrote
//     after they had finished reading step s;
//   * the per-row state (two neighbours, bounds, flags, size) lives in the REGISTERS of the row's thread for the whole run;
//   * the 8 XCDs' L2s are not coherent with each other inside a launch, so every word another workgroup may read — the slots,
//     D (thread z writes D[z][hi], thread hi reads it when row z merges later), size[], cid[] — is accessed with agent-scope
//     relaxed atomics (sc1: loads miss the local L2, stores write through), and a thread drains its own stores (vmcnt 0) before
//     the workgroup publishes.  No release / acquire fences: buffer_wbl2 / buffer_inv would write back and invalidate the
//     whole L2 of an XCD that the engine's kernels of the NEXT recording are using at the same time (pipeline.diarize_many).
// Every workgroup takes the same decision from the same words, so all of them leave the loop in the same step.  A poll that
// does not complete in ~10 s (a workgroup that was never placed) ends the launch with PERSIST_TIMEOUT and the host falls back
// to the launch-per-step loop.  Same dendrogram bits: tests/test_ops_gpu.py (vs scipy, vs the step loops, 30 k golden).
constexpr int SLOT_W = 10;
enum { PERSIST_RUNNING = 0, PERSIST_DONE = 1, PERSIST_FAIL = 2, PERSIST_TIMEOUT = 3 };

__device__ __forceinline__ uint64_t ld_agent(const uint64_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(uint64_t* p, uint64_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ldd_agent(const double* p) {
  return __longlong_as_double((long long)ld_agent(reinterpret_cast<const uint64_t*>(p)));
}
__device__ __forceinline__ void std_agent(double* p, double v) {
  st_agent(reinterpret_cast<uint64_t*>(p), (uint64_t)__double_as_longlong(v));
}
__device__ __forceinline__ int ldi_agent(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void sti_agent(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// block_top2_min for NW wavefronts (sval: 3 NW doubles, sidx: 3 NW ints)
template <int NW>
__device__ __forceinline__ void block_top2_min_n(double a1, int ai1, double a2, int ai2, double m, int mi, double& o1, int& oi1,
                                                 double& o2, int& oi2, double& om, int& omi, double* sval, int* sidx) {
  top2_min_stage<0xB1, 0xf>(a1, ai1, a2, ai2, m, mi);
  top2_min_stage<0x4E, 0xf>(a1, ai1, a2, ai2, m, mi);
  top2_min_stage<0x124, 0xf>(a1, ai1, a2, ai2, m, mi);
  top2_min_stage<0x128, 0xf>(a1, ai1, a2, ai2, m, mi);
  top2_min_stage<0x142, 0xa>(a1, ai1, a2, ai2, m, mi);
  top2_min_stage<0x143, 0xc>(a1, ai1, a2, ai2, m, mi);
  a1 = readlane63_d(a1); a2 = readlane63_d(a2); m = readlane63_d(m);
  ai1 = __builtin_amdgcn_readlane(ai1, 63); ai2 = __builtin_amdgcn_readlane(ai2, 63); mi = __builtin_amdgcn_readlane(mi, 63);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) {
    sval[wave] = a1; sval[NW + wave] = a2; sval[2 * NW + wave] = m;
    sidx[wave] = ai1; sidx[NW + wave] = ai2; sidx[2 * NW + wave] = mi;
  }
  __syncthreads();
  a1 = sval[0]; a2 = sval[NW]; m = sval[2 * NW];
  ai1 = sidx[0]; ai2 = sidx[NW]; mi = sidx[2 * NW];
#pragma unroll
  for (int w = 1; w < NW; ++w) {
    top2_merge(a1, ai1, a2, ai2, sval[w], sidx[w], sval[NW + w], sidx[NW + w]);
    if (pair_lt(sval[2 * NW + w], sidx[2 * NW + w], m, mi)) { m = sval[2 * NW + w]; mi = sidx[2 * NW + w]; }
  }
  o1 = a1; oi1 = ai1; o2 = a2; oi2 = ai2; om = m; omi = mi;
}

template <int BS>
__global__ __launch_bounds__(BS) void persist2_kernel(double* __restrict__ D, int n, const double* __restrict__ l1a,
                                                      const double* __restrict__ l2a, const int* __restrict__ n1a,
                                                      const int* __restrict__ n2a, const int* __restrict__ fla,
                                                      int* __restrict__ size, int* __restrict__ cid, double* __restrict__ Z,
                                                      uint64_t* __restrict__ slots, int* __restrict__ status,
                                                      int spin_limit) {
  constexpr int NW = BS / 64;
  __shared__ double sval[3 * NW];
  __shared__ int sidx[3 * NW];
  __shared__ int sy[2];
  const int G = gridDim.x, w = blockIdx.x, t = threadIdx.x;     // G <= BS (host)
  const int z = w * BS + t;
  const bool in = z < n;
  // the row's state, in registers from here on (init_rows2_kernel left the exact two nearest neighbours; sizes are all 1)
  double my_l1 = in ? l1a[z] : DINF, my_l2 = in ? l2a[z] : DINF;
  int my_n1 = in ? n1a[z] : -1, my_n2 = in ? n2a[z] : -1, my_fl = in ? fla[z] : 0;
  int my_sz = in ? 1 : 0;
  bool owed = false;      // the row of the previous step (merged: hi, rescanned: x) gets its two neighbours from the parts
  int ex = 0x7fffffff;    // that row
  int k = 0, rescans = 0;
  double pv = DINF;       // this thread's element of the row whose two smallest the step publishes
  int excl = -1;          // row left out of this step's records
  bool pend = false;      // the previous step was a merge: its pair's size[] / cid[] words are written in this step
  int p_lo = -1, p_hi = -1, p_sum = 0, p_cid = 0;
  for (unsigned seq = 1;; ++seq) {
    // ---- publish: the two smallest of the new / rescanned row's slice, the record of this workgroup's rows ----
    __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0): this thread's stores to D / size / cid are written through
    double b1, b2, rv;
    int bi1, bi2, rx;
    block_top2_min_n<NW>(pv, in ? z : 0x7fffffff, DINF, 0x7fffffff, (in && z != excl) ? my_l1 : DINF, in ? z : 0x7fffffff, b1,
                         bi1, b2, bi2, rv, rx, sval, sidx);      // (its barriers order every thread's drain before the stores below)
    uint64_t* mine = slots + ((size_t)(seq & 1) * G + w) * SLOT_W;
    const uint64_t stamp = (uint64_t)seq << 32;
    if (t == 0) {
      const uint64_t u1 = (uint64_t)__double_as_longlong(b1), u2 = (uint64_t)__double_as_longlong(b2);
      st_agent(mine + 4, stamp | (u1 & 0xffffffffu));
      st_agent(mine + 5, stamp | (u1 >> 32));
      st_agent(mine + 6, stamp | (u2 & 0xffffffffu));
      st_agent(mine + 7, stamp | (u2 >> 32));
      st_agent(mine + 8, stamp | (uint32_t)bi1);
      st_agent(mine + 9, stamp | (uint32_t)bi2);
    }
    if (z == rx) {          // rx is always a row of this workgroup (every thread contributes its own index)
      const uint64_t uv = (uint64_t)__double_as_longlong(rv);
      const int ry = rv < DINF ? my_n1 : -1;
      const uint32_t rexact = rv < DINF && (my_fl & 1) != 0 && my_n1 >= 0;
      st_agent(mine + 0, stamp | (uv & 0xffffffffu));
      st_agent(mine + 1, stamp | (uv >> 32));
      st_agent(mine + 2, stamp | ((uint32_t)rx | (rexact << 31)));
      st_agent(mine + 3, stamp | (uint32_t)ry);
    }
    // ---- poll every workgroup's slot of this step ----
    uint64_t wd[SLOT_W];
    const uint64_t* theirs = slots + ((size_t)(seq & 1) * G + (t < G ? t : 0)) * SLOT_W;
    for (int spins = 0;; ++spins) {
      bool ok = true;
      if (t < G) {
#pragma unroll
        for (int i = 0; i < SLOT_W; ++i) wd[i] = ld_agent(theirs + i);
#pragma unroll
        for (int i = 0; i < SLOT_W; ++i) ok = ok && (uint32_t)(wd[i] >> 32) == seq;
      }
      if (__syncthreads_and(ok)) break;
      if (spins > spin_limit) {       // the same value in every thread of the workgroup
        if (t == 0) sti_agent(status, PERSIST_TIMEOUT);
        return;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    double r0v = DINF, p0v1 = DINF, p0v2 = DINF;
    int r0x = 0x7fffffff, r0y = -1, r0exact = 0, p0i1 = 0x7fffffff, p0i2 = 0x7fffffff;
    if (t < G) {
      r0v = __longlong_as_double((long long)((wd[0] & 0xffffffffu) | (wd[1] << 32)));
      r0x = (int)((uint32_t)wd[2] & 0x7fffffffu);
      r0exact = (int)(((uint32_t)wd[2]) >> 31);
      r0y = (int)(uint32_t)wd[3];
      p0v1 = __longlong_as_double((long long)((wd[4] & 0xffffffffu) | (wd[5] << 32)));
      p0v2 = __longlong_as_double((long long)((wd[6] & 0xffffffffu) | (wd[7] << 32)));
      p0i1 = (int)(uint32_t)wd[8];
      p0i2 = (int)(uint32_t)wd[9];
    }
    // ---- the two neighbours owed to the row of the previous step + the winner over the records ----
    double ev1, ev2, d;
    int eid1, eid2, x;
    block_top2_min_n<NW>(owed ? p0v1 : DINF, owed ? p0i1 : 0x7fffffff, owed ? p0v2 : DINF, owed ? p0i2 : 0x7fffffff, r0v, r0x,
                         ev1, eid1, ev2, eid2, d, x, sval, sidx);
    const int ey = ev1 < DINF ? eid1 : -1;
    if (owed && z == ex) {
      my_l1 = ev1; my_n1 = ey; my_l2 = ev2; my_n2 = ev2 < DINF ? eid2 : -1;
      my_fl = (my_n1 >= 0 ? 1 : 0) | (my_n2 >= 0 ? 2 : 0);
    }
    int y;
    bool exact;
    if (owed && pair_lt(ev1, ex, d, x)) {
      d = ev1; x = ex; y = ey; exact = ey >= 0;
    } else {
      __syncthreads();
      if (t < G && r0x == x && r0v == d) { sy[0] = r0y; sy[1] = r0exact; }
      __syncthreads();
      y = sy[0];
      exact = sy[1] != 0 && y >= 0;
    }
    if (!(d < DINF) || x < 0 || x >= n) {    // NaN / inf distances: no pair left to merge
      if (w == 0 && t == 0) { status[1] = k; status[2] = rescans; sti_agent(status, PERSIST_FAIL); }
      return;
    }
    // the words of the previous merge's pair that other workgroups look up: written one step late, when every workgroup has
    // finished the step that still read the old values (it published this step's words after those loads had returned)
    if (pend) {
      if (z == p_hi) { sti_agent(size + p_hi, p_sum); sti_agent(cid + p_hi, p_cid); }
      if (z == p_lo) sti_agent(size + p_lo, 0);
    }
    bool pend_next = false;
    pv = DINF;
    excl = -1;
    if (!exact) {
      // ---- RESCAN: this workgroup's slice of row x over the active columns (D[x][x] is +inf) ----
      ++rescans;
      if (in && my_sz != 0) pv = ldd_agent(D + (int64_t)x * n + z);
      excl = x;
      owed = true;
      ex = x;
    } else {
      // ---- MERGE: Lance-Williams update of this workgroup's columns ----
      const int lo = x < y ? x : y, hi = x < y ? y : x;
      // size[] / cid[] of the previous pair are rewritten by their owners in THIS step (above): substitute instead of reading them
      const int nlo = pend && lo == p_hi ? p_sum : ldi_agent(size + lo);      // (lo / hi are active: never p_lo)
      const int nhi = pend && hi == p_hi ? p_sum : ldi_agent(size + hi);
      if (w == 0 && t == 0) {
        const int ia = pend && lo == p_hi ? p_cid : ldi_agent(cid + lo), ib = pend && hi == p_hi ? p_cid : ldi_agent(cid + hi);
        Z[4 * k + 0] = (double)(ia < ib ? ia : ib);
        Z[4 * k + 1] = (double)(ia < ib ? ib : ia);
        Z[4 * k + 2] = d;
        Z[4 * k + 3] = (double)(nlo + nhi);
      }
      if (in) {
        if (z == hi) {
          std_agent(D + (int64_t)hi * n + lo, DINF);
          my_l1 = my_l2 = DINF; my_n1 = my_n2 = -1; my_fl = 0;      // both neighbours after the next poll
          my_sz = nlo + nhi;
        } else if (z == lo) {
          my_l1 = my_l2 = DINF; my_fl = 0;                          // retired
          my_sz = 0;
        } else if (my_sz != 0) {
          const double dxi = ldd_agent(D + (int64_t)lo * n + z), dyi = ldd_agent(D + (int64_t)hi * n + z);
          // scipy _hierarchy_distance_update.pxi, _centroid(d_xi, d_yi, d_xy, size_x, size_y, size_i), same order
          const double v = sqrt((((nlo * dxi * dxi) + (nhi * dyi * dyi)) - (nlo * nhi * d * d) / (nlo + nhi)) / (nlo + nhi));
          std_agent(D + (int64_t)hi * n + z, v);
          std_agent(D + (int64_t)z * n + hi, v);       // column lo is NOT blanked: readers of a row mask by their size
          pv = v;
          // columns lo and (old) hi leave the row, column hi re-enters with v: the rules of step2_kernel's header
          const bool d1 = my_n1 == lo || my_n1 == hi, d2 = my_n2 == lo || my_n2 == hi;
          const bool e1 = my_fl & 1, e2 = my_fl & 2;
          if (!d1) {
            if (v < my_l1) {
              my_l2 = my_l1; my_n2 = my_n1; my_fl = 1 | (e1 ? 2 : 0);
              my_l1 = v; my_n1 = hi;
            } else if (!d2) {
              if (v < my_l2) { my_l2 = v; my_n2 = hi; my_fl |= 2; }
            } else if (v <= my_l2) {
              my_l2 = v; my_n2 = hi; my_fl |= 2;
            } else {
              my_n2 = hi; my_fl &= ~2;
            }
          } else {
            if (!d2 && e2) {
              if (v < my_l2) { my_l1 = v; my_n1 = hi; my_fl = 3; }
              else { my_l1 = my_l2; my_n1 = my_n2; my_n2 = hi; my_fl = 1; }
            } else {
              if (v <= my_l2) { my_l1 = v; my_n1 = hi; my_fl = 1; }
              else { my_l1 = my_l2; my_n1 = hi; my_fl = 0; }
              if (d2) my_n2 = hi;
            }
          }
        }
      }
      owed = true;
      ex = hi;
      pend_next = true; p_lo = lo; p_hi = hi; p_sum = nlo + nhi; p_cid = n + k;
      if (++k >= n - 1) {
        if (w == 0 && t == 0) { status[1] = k; status[2] = rescans; status[3] = (int)seq; sti_agent(status, PERSIST_DONE); }
        return;
      }
    }
    pend = pend_next;
  }
}

#define LCHK(call)                                   \
  do {                                               \
    if ((call) != hipSuccess) { rc = DZN_E_HIP; goto done; } \
  } while (0)



// ---- the host stage's device context (r5) -----------------------------------------------------------------------------
// The clustering entry points below are called from the HOST stage of the pipeline, which (pipeline.diarize_many, bench.py)
// now runs in its own thread WHILE the engine executes the device stage of the next recording.  So they must not touch the
// legacy null stream (its copies and launches order themselves behind every kernel the engine has queued) and must not
// hipFree (a device-wide synchronisation: the host thread would sit out the engine's whole queue).  Per device:
//   * one non-blocking stream of the highest priority the device offers — the single-workgroup-per-256-rows step kernels are
//     latency bound, their workgroups are placed as soon as a CU of the engine's launches drains;
//   * one grow-only arena that every call carves its buffers from (re-allocated only when a call needs more than any before
//     it; dzn_host_workspace_release() returns it).  A mutex serialises the calls of one process.
struct HostCtx {
  hipStream_t stream = nullptr;
  char* base = nullptr;
  size_t cap = 0;
  char* sbase = nullptr;      // second arena: a STATE that lives across calls (vbx.hip), leased to one state at a time
  size_t scap = 0;
  bool sleased = false;
};
constexpr int MAX_DEV = 64;
std::mutex g_host_mu;
HostCtx g_host[MAX_DEV];

// the context of the CURRENT device with an arena of at least `bytes` (g_host_mu held)
int host_ctx(int device, size_t bytes, HostCtx** out) {
  int dev = device;
  if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return DZN_E_HIP;
  if (dev < 0 || dev >= MAX_DEV) return DZN_E_INVALID;
  HostCtx& c = g_host[dev];
  if (!c.stream) {
    int lo = 0, hi = 0;      // "least" and "greatest" priority: numerically greatest = lowest
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) lo = hi = 0;
    if (hipStreamCreateWithPriority(&c.stream, hipStreamNonBlocking, hi) != hipSuccess) {
      c.stream = nullptr;
      return DZN_E_HIP;
    }
  }
  if (bytes == 0) {      // stream only
    *out = &c;
    return DZN_OK;
  }
  if (bytes > c.cap) {
    if (c.base) {
      (void)hipStreamSynchronize(c.stream);
      (void)hipFree(c.base);
      c.base = nullptr;
      c.cap = 0;
    }
    const size_t want = bytes + bytes / 8;      // a little headroom: recordings of one corpus differ by a few rows
    if (hipMalloc(&c.base, want) != hipSuccess) {
      (void)hipGetLastError();
      if (hipMalloc(&c.base, bytes) != hipSuccess) { (void)hipGetLastError(); c.base = nullptr; return DZN_E_NOMEM; }
      c.cap = bytes;
    } else {
      c.cap = want;
    }
  }
  *out = &c;
  return DZN_OK;
}

struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(char* b) : base(b) {}
  template <typename T>
  T* take(size_t count) {
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += (count * sizeof(T) + 255) & ~(size_t)255;
    return p;
  }
};

}  // namespace

int host_stage_stream(int device, hipStream_t* stream) {
  std::lock_guard<std::mutex> lk(g_host_mu);
  HostCtx* c = nullptr;
  const int rc = host_ctx(device, 0, &c);
  if (rc == DZN_OK) *stream = c->stream;
  return rc;
}

int host_state_lease(int device, size_t bytes, hipStream_t* stream, char** base) {
  std::lock_guard<std::mutex> lk(g_host_mu);
  HostCtx* c = nullptr;
  const int rc = host_ctx(device, 0, &c);
  if (rc != DZN_OK) return rc;
  *stream = c->stream;
  *base = nullptr;
  if (c->sleased) return DZN_OK;                 // a second live state: the caller allocates for itself
  if (bytes > c->scap) {
    if (c->sbase) {
      (void)hipStreamSynchronize(c->stream);
      (void)hipFree(c->sbase);
      c->sbase = nullptr;
      c->scap = 0;
    }
    if (hipMalloc(&c->sbase, bytes + bytes / 8) == hipSuccess) c->scap = bytes + bytes / 8;
    else {
      (void)hipGetLastError();
      if (hipMalloc(&c->sbase, bytes) != hipSuccess) { (void)hipGetLastError(); c->sbase = nullptr; return DZN_E_NOMEM; }
      c->scap = bytes;
    }
  }
  c->sleased = true;
  *base = c->sbase;
  return DZN_OK;
}

void host_state_release(int device) {
  std::lock_guard<std::mutex> lk(g_host_mu);
  int dev = device;
  if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return;
  if (dev >= 0 && dev < MAX_DEV) g_host[dev].sleased = false;
}

extern "C" int dzn_host_workspace_release(int32_t device) {
  std::lock_guard<std::mutex> lk(g_host_mu);
  for (int dev = 0; dev < MAX_DEV; ++dev) {
    if (device >= 0 && dev != device) continue;
    HostCtx& c = g_host[dev];
    if (!c.base && !c.sbase && !c.stream) continue;
    DeviceGuard dg(dev);
    if (!dg.ok) return DZN_E_HIP;
    if (c.stream) (void)hipStreamSynchronize(c.stream);
    if (c.base) (void)hipFree(c.base);
    c.base = nullptr;
    c.cap = 0;
    if (c.sbase && !c.sleased) {      // a live state keeps its arena
      (void)hipFree(c.sbase);
      c.sbase = nullptr;
      c.scap = 0;
    }
  }
  return DZN_OK;
}

extern "C" int64_t dzn_host_workspace_bytes(int32_t device) {
  std::lock_guard<std::mutex> lk(g_host_mu);
  return device >= 0 && device < MAX_DEV ? (int64_t)(g_host[device].cap + g_host[device].scap) : 0;
}

extern "C" int dzn_linkage_centroid(const float* h_emb, int32_t n, int32_t dim, double* h_Z, int32_t device) {
  if (!h_emb || !h_Z || n < 2 || dim < 1) return DZN_E_INVALID;
  int rc = DZN_OK;
  const int nblk = (n + LB_BLK - 1) / LB_BLK;
  // DZN_LINKAGE_TWO_KERNEL=1: the r2 loop (one single-workgroup selection + one wide update per merge), kept for A/B timing;
  // DZN_LINKAGE_TOP1=1: r3-r5's step loop with ONE remembered neighbour per row (step_kernel) instead of r6's two
  const bool two_kernel = getenv("DZN_LINKAGE_TWO_KERNEL") != nullptr;
  const bool top2 = !two_kernel && getenv("DZN_LINKAGE_TOP1") == nullptr;
  // DZN_LINKAGE_PERSIST=1: r6b's single persistent launch instead of the launch-per-step loop (measured SLOWER: 9.4 vs 8.2 us
  // per step at n = 20 888 — a kernel boundary costs ~1.5 us here, an all-to-all exchange of records inside a launch >= 3 us;
  // profiles/r6_linkage_persist.txt).  Kept as the measured record of VERDICT r5 item 2a and as a cross-check of the rules.
  bool persist = top2 && getenv("DZN_LINKAGE_PERSIST") != nullptr;
  const int pbs = n <= 32768 ? 256 : 1024;                       // workgroup size of the persistent launch: G <= 128 up to n = 131 072
  const int pgrid = (n + pbs - 1) / pbs;
  std::vector<int> ones(n, 1), ids(n);
  for (int i = 0; i < n; ++i) ids[i] = i;
  MergeState st0{};
  DeviceGuard dg(device);      // restores the caller's device on every return path
  if (!dg.ok) return DZN_E_HIP;
  std::lock_guard<std::mutex> lk(g_host_mu);
  // the buffers of one call, carved from the arena (first pass: sizes only)
  float* E = nullptr;
  double *D = nullptr, *lb = nullptr, *Z = nullptr, *bmin = nullptr, *hp_val = nullptr;
  int *nb = nullptr, *size = nullptr, *cid = nullptr, *barg = nullptr, *hp_idx = nullptr, *exf = nullptr;
  MergeState* st = nullptr;
  StepState* st2 = nullptr;
  StepRec* rec2 = nullptr;
  StepPart* part2 = nullptr;
  StepPart2* part22 = nullptr;
  double* l2a = nullptr;
  int* n2a = nullptr;
  uint64_t* slots = nullptr;
  int* pstatus = nullptr;
  auto carve = [&](char* base) {
    Carver c(base);
    D = c.take<double>((size_t)n * n);
    E = c.take<float>((size_t)n * dim);
    lb = c.take<double>(n);
    Z = c.take<double>((size_t)(n - 1) * 4);
    nb = c.take<int>(n);
    size = c.take<int>(n);
    cid = c.take<int>(n);
    st = c.take<MergeState>(1);
    bmin = c.take<double>(nblk);
    hp_val = c.take<double>(nblk);
    barg = c.take<int>(nblk);
    hp_idx = c.take<int>(nblk);
    st2 = c.take<StepState>(2);
    rec2 = c.take<StepRec>((size_t)2 * nblk);
    part2 = c.take<StepPart>((size_t)2 * nblk);
    exf = c.take<int>(n);
    part22 = c.take<StepPart2>((size_t)2 * nblk);
    l2a = c.take<double>(n);
    n2a = c.take<int>(n);
    slots = c.take<uint64_t>((size_t)2 * pgrid * SLOT_W);
    pstatus = c.take<int>(4);
    return c.off;
  };
  HostCtx* ctx = nullptr;
  rc = host_ctx(device, carve(nullptr), &ctx);
  if (rc != DZN_OK) return rc;
  carve(ctx->base);
  hipStream_t s = ctx->stream;
  LCHK(hipMemcpyAsync(E, h_emb, (size_t)n * dim * sizeof(float), hipMemcpyHostToDevice, s));
  LCHK(hipMemcpyAsync(size, ones.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, s));
  LCHK(hipMemcpyAsync(cid, ids.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, s));
  LCHK(hipMemcpyAsync(st, &st0, sizeof(MergeState), hipMemcpyHostToDevice, s));
  {
    const int tiles = (n + 63) / 64;
    hipLaunchKernelGGL(pdist_kernel, dim3(tiles, tiles), dim3(256), 0, s, E, n, dim, D);
    if (top2) hipLaunchKernelGGL(init_rows2_kernel, dim3(n), dim3(256), 0, s, D, n, lb, l2a, nb, n2a, exf);
    else hipLaunchKernelGGL(init_rows_kernel, dim3(n), dim3(256), 0, s, D, n, lb, nb);
    if (persist) {
      // every workgroup must be resident at once (they wait for each other inside the launch)
      int per_cu = 0, cus = 0, dev_now = 0;
      const void* kfn = pbs == 256 ? (const void*)persist2_kernel<256> : (const void*)persist2_kernel<1024>;
      if (pgrid > pbs || hipGetDevice(&dev_now) != hipSuccess ||
          hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev_now) != hipSuccess ||
          hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, pbs, 0) != hipSuccess || (int64_t)per_cu * cus < pgrid) {
        (void)hipGetLastError();
        persist = false;
      }
    }
    if (persist) {
      LCHK(hipMemsetAsync(slots, 0, (size_t)2 * pgrid * SLOT_W * sizeof(uint64_t), s));
      LCHK(hipMemsetAsync(pstatus, 0, 4 * sizeof(int), s));
      const auto t_loop = std::chrono::steady_clock::now();
      const int spin_limit = 8 << 20;
      if (pbs == 256)
        hipLaunchKernelGGL(persist2_kernel<256>, dim3(pgrid), dim3(256), 0, s, D, n, lb, l2a, nb, n2a, exf, size, cid, Z, slots,
                           pstatus, spin_limit);
      else
        hipLaunchKernelGGL(persist2_kernel<1024>, dim3(pgrid), dim3(1024), 0, s, D, n, lb, l2a, nb, n2a, exf, size, cid, Z, slots,
                           pstatus, spin_limit);
      LCHK(hipGetLastError());
      int hst[4] = {0, 0, 0, 0};
      LCHK(hipMemcpyAsync(hst, pstatus, sizeof(hst), hipMemcpyDeviceToHost, s));
      LCHK(hipStreamSynchronize(s));
      if (getenv("DZN_LINKAGE_DEBUG"))
        fprintf(stderr, "linkage[persistent, %d x %d]: n %d, status %d, %d merges, %d rescans, %d steps, loop %.1f ms\n", pgrid, pbs, n,
                hst[0], hst[1], hst[2], hst[3], std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_loop).count());
      if (hst[0] == PERSIST_FAIL) { rc = DZN_E_INVALID; goto done; }       // non-finite distances
      if (hst[0] != PERSIST_DONE) {
        // a workgroup was not placed in time: D is part-way through the dendrogram, start over with the launch-per-step loop
        fprintf(stderr, "dzn_linkage_centroid: the persistent launch ended with status %d, repeating with the step loop\n", hst[0]);
        persist = false;
        LCHK(hipMemcpyAsync(size, ones.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, s));
        LCHK(hipMemcpyAsync(cid, ids.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(pdist_kernel, dim3(tiles, tiles), dim3(256), 0, s, E, n, dim, D);
        hipLaunchKernelGGL(init_rows2_kernel, dim3(n), dim3(256), 0, s, D, n, lb, l2a, nb, n2a, exf);
      }
    }
    if (persist) {
    } else if (two_kernel) {
      hipLaunchKernelGGL(block_minima_kernel, dim3(nblk), dim3(LB_BLK), 0, s, lb, n, bmin, barg);
      for (int k = 0; k < n - 1; ++k) {
        hipLaunchKernelGGL(select_kernel, dim3(1), dim3(1024), 0, s, D, n, lb, nb, size, cid, Z, st, bmin, barg, hp_val,
                           hp_idx);
        hipLaunchKernelGGL(update_kernel, dim3(nblk), dim3(LB_BLK), 0, s, D, n, lb, nb, size, st, bmin, barg, hp_val,
                           hp_idx);
      }
    } else {
      // one launch per step (merge or rescan); the number of rescans is data dependent, so launches are queued in
      // batches sized from the merges still missing and the step descriptor is read back between batches (surplus
      // launches after the last merge return at once)
      LCHK(hipMemsetAsync(st2, 0, 2 * sizeof(StepState), s));
      if (top2) hipLaunchKernelGGL(init_rec2_kernel, dim3(nblk), dim3(LB_BLK), 0, s, n, lb, nb, rec2);
      else hipLaunchKernelGGL(init_rec_kernel, dim3(nblk), dim3(LB_BLK), 0, s, n, lb, nb, exf, rec2);
      int64_t launched = 0;
      int remaining = n - 1;
      const auto t_loop = std::chrono::steady_clock::now();
      for (int round = 0; remaining > 0; ++round) {
        if (round > 64 + n) { rc = DZN_E_INVALID; goto done; }   // cannot happen: every batch completes >= 1 merge
        // surplus launches return at once but still cost a kernel boundary each: size the batch for the expected rescans
        // (1.1-1.3 launches per merge with two remembered neighbours, 2-3 with one) and let the next round finish the rest
        const int batch = top2 ? remaining + remaining / 8 + 32 : remaining + remaining / 2 + 32;
        for (int i = 0; i < batch; ++i, ++launched) {
          if (top2)
            hipLaunchKernelGGL(step2_kernel, dim3(nblk), dim3(LB_BLK), 0, s, D, n, lb, l2a, nb, n2a, exf, size, cid, Z, st2,
                               rec2, part22, (int)(launched & 1));
          else
            hipLaunchKernelGGL(step_kernel, dim3(nblk), dim3(LB_BLK), 0, s, D, n, lb, nb, size, cid, exf, Z, st2, rec2, part2,
                               (int)(launched & 1));
        }
        LCHK(hipGetLastError());
        StepState hs;
        LCHK(hipMemcpyAsync(&hs, st2 + (launched & 1), sizeof(StepState), hipMemcpyDeviceToHost, s));
        LCHK(hipStreamSynchronize(s));
        if (hs.kind == STEP_FAIL) { rc = DZN_E_INVALID; goto done; }   // non-finite distances
        remaining = hs.kind == STEP_DONE ? 0 : n - 1 - hs.k;
        if (remaining == 0 && getenv("DZN_LINKAGE_DEBUG"))
          fprintf(stderr, "linkage[%s]: n %d, %lld launches in %d batches, %d rescans, loop %.1f ms\n",
                  top2 ? "two neighbours" : "one neighbour", n, (long long)launched, round + 1, hs.pad, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_loop).count());
      }
    }
  }
  LCHK(hipGetLastError());
  LCHK(hipMemcpyAsync(h_Z, Z, (size_t)(n - 1) * 4 * sizeof(double), hipMemcpyDeviceToHost, s));
done:
  // the host vectors above (ones, ids, st0) are sources of queued copies; an asynchronous fault of a step kernel or of the final
  // copy surfaces HERE, and h_Z is garbage then: it must not return DZN_OK
  if (hipStreamSynchronize(s) != hipSuccess && rc == DZN_OK) rc = DZN_E_HIP;
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
  DeviceGuard dg(device);      // restores the caller's device on every return path
  if (!dg.ok) return DZN_E_HIP;
  std::lock_guard<std::mutex> lk(g_host_mu);       // own stream + arena: see "the host stage's device context" above
  float* E = nullptr;
  double *Cn = nullptr, *ne = nullptr, *nc = nullptr, *D = nullptr;
  auto carve = [&](char* base) {
    Carver c(base);
    E = c.take<float>((size_t)n * dim);
    Cn = c.take<double>((size_t)k * dim);
    ne = c.take<double>(n);
    nc = c.take<double>(k);
    D = c.take<double>((size_t)n * k);
    return c.off;
  };
  HostCtx* ctx = nullptr;
  rc = host_ctx(device, carve(nullptr), &ctx);
  if (rc != DZN_OK) return rc;
  carve(ctx->base);
  hipStream_t s = ctx->stream;
  LCHK(hipMemcpyAsync(E, h_emb, (size_t)n * dim * sizeof(float), hipMemcpyHostToDevice, s));
  LCHK(hipMemcpyAsync(Cn, h_cent, (size_t)k * dim * sizeof(double), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(row_norm_f32_kernel, dim3((n + 255) / 256), dim3(256), 0, s, E, n, dim, ne);
  hipLaunchKernelGGL(row_norm_f64_kernel, dim3((k + 255) / 256), dim3(256), 0, s, Cn, k, dim, nc);
  hipLaunchKernelGGL(cdist_cosine_kernel, dim3((unsigned)(((int64_t)n * k + 255) / 256)), dim3(256), 0, s, E, Cn, ne, nc,
                     n, dim, k, D);
  LCHK(hipGetLastError());
  LCHK(hipMemcpyAsync(h_dist, D, (size_t)n * k * sizeof(double), hipMemcpyDeviceToHost, s));
done:
  if (hipStreamSynchronize(s) != hipSuccess && rc == DZN_OK) rc = DZN_E_HIP;   // h_dist is only valid after a clean drain
  return rc;
}
