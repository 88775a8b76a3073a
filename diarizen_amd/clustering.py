"""Host-side clustering of the per-(window, local speaker) embeddings.

Own implementation of the behaviour of (PA/ = pyannote-audio/pyannote/audio/):
  * BaseClustering.filter_embeddings / constrained_argmax / assign_embeddings
        PA/pipelines/clustering.py:111-245, filter_embeddings_by_frames :47-73
  * AgglomerativeClustering.cluster (centroid linkage on unit-normalised embeddings, distance
    threshold, small clusters merged into the nearest large one)      :363-513
  * VBxClustering.__call__ (AHC init -> PLDA-space VB-GMM -> constrained assignment) :632-700
  * diarizen/clustering/VBx.py:27-194 (VBx with loopProb = 0 => GMM update branch; vbx_setup)
This stays on the host by design (BASELINE.json north_star); scipy does linkage / Hungarian
assignment exactly as in the reference so that cluster ids — hence RTTM labels — are equal.
"""
from __future__ import annotations

import random
from typing import Optional, Tuple

import numpy as np
from scipy.cluster.hierarchy import fcluster, linkage
from scipy.linalg import eigh
from scipy.optimize import linear_sum_assignment
from scipy.spatial.distance import cdist
from scipy.special import logsumexp, softmax


# --------------------------------------------------------------------------- centroid linkage
# From this many embeddings up the device loop is used.  (r6) with two remembered neighbours per row the device overtakes scipy's host
# loop at n ~ 300 on an idle device (3.9 vs 7.0 ms at 512, 15 vs 137 ms at 2048: profiles/r6_linkage_crossover.txt); 512 leaves a
# factor for a device that is busy with the next recording's windows (pipeline.diarize_many).  Was 2048 (tuned on r3's loop).
HIP_LINKAGE_MIN = 512


def _hip_ready() -> bool:
    try:
        import torch
        from . import _lib
        return torch.cuda.is_available() and _lib.lib_path().exists()
    except Exception:       # pragma: no cover
        return False


def centroid_linkage(emb: np.ndarray, backend: str = "auto", device: int = -1) -> np.ndarray:
    """linkage(emb, method="centroid", metric="euclidean") (PA/pipelines/clustering.py:407-416, :656).
    backend "scipy": the reference's own call.  "hip": csrc/linkage.hip — the same greedy algorithm with
    the float64 distance matrix resident in HBM; produces the identical dendrogram (checked bit for bit
    in tests/test_ops_gpu.py) 8x faster at 30 min of audio and without scipy's 8 n^2 / 2 bytes of host
    memory at hours of audio.  "auto": hip from HIP_LINKAGE_MIN embeddings up when a device is present."""
    if backend not in ("auto", "scipy", "hip"):
        raise ValueError(f"unknown linkage backend {backend!r}")
    # the device kernel computes its float64 distances from FLOAT32 coordinates: only float32 embeddings (what the
    # engine produces) may take it in auto mode — float64 input would be rounded first and could merge in another order
    use_hip = backend == "hip" or (backend == "auto" and len(emb) >= HIP_LINKAGE_MIN and emb.dtype == np.float32
                                   and _hip_ready())
    if use_hip and backend == "auto" and _has_duplicate_rows(emb):
        # exact distance ties: scipy's neighbour-heap order and the device's lowest-index argmin may merge tied
        # pairs in a different order -> keep the reference's own call whenever ties are certain
        use_hip = False
    if not use_hip:
        return linkage(emb, method="centroid", metric="euclidean")
    from . import ops
    from ._lib import DznError
    try:
        return ops.linkage_centroid(emb, device=device)
    except (DznError, MemoryError):
        if backend == "hip":
            raise
        return linkage(emb, method="centroid", metric="euclidean")     # e.g. the n x n matrix does not fit


def _fcluster_distance(Z: np.ndarray, t: float) -> np.ndarray:
    """`fcluster(Z, t, criterion="distance")` (PA/pipelines/clustering.py:418) for a dendrogram that `linkage` (scipy's or the
    device's) has just produced: scipy's wrapper first re-validates Z in Python (`is_valid_linkage`: 27 ms of the 4 h host
    stage for 20 k merges, r6 profile) before it calls the routine that does the work — `_hierarchy.cluster_dist`, which this
    calls directly with the same arguments.  A scipy build without that private routine takes the public call."""
    try:
        from scipy.cluster import _hierarchy
        Zc = np.ascontiguousarray(Z, dtype=np.float64)
        n = Zc.shape[0] + 1
        T = np.zeros((n,), dtype="i")
        _hierarchy.cluster_dist(Zc, T, float(t), int(n))
        return T
    except (ImportError, AttributeError, TypeError):      # pragma: no cover - other scipy layouts
        return fcluster(Z, t, criterion="distance")


def _has_duplicate_rows(emb: np.ndarray) -> bool:
    v = np.ascontiguousarray(emb).view(np.dtype((np.void, emb.dtype.itemsize * emb.shape[1]))).ravel()
    return len(np.unique(v)) < len(v)


# --------------------------------------------------------------------------- shared pieces
def single_speaker_frame_mask(seg: np.ndarray, min_frames: int) -> np.ndarray:
    """[C, L, S] -> bool [C, S]: speaker has >= min_frames frames where it is the ONLY one active
    (PA/pipelines/clustering.py:111-131: `np.sum(seg * (np.sum(seg, axis=2, keepdims=True) == 1), axis=1)`).
    Hard {0,1} decisions (the only thing the pipeline passes) are counted as bytes: 17 991 x 399 x 4 at 4 h take
    0.05 s instead of 1.5 s in float32."""
    u = seg if seg.dtype == np.uint8 else seg.astype(np.uint8)
    if seg.dtype != np.uint8 and not np.array_equal(u, seg):          # soft scores: the reference expression
        alone = np.sum(seg, axis=2, keepdims=True) == 1
        return np.sum(seg * alone, axis=1) >= min_frames
    C, L, S = u.shape
    if S == 4 and np.little_endian:
        w = np.ascontiguousarray(u).view(np.uint32).reshape(C, L)    # the 4 speakers of a frame in one word
        alone = ((w & (w - 1)) == 0) & (w != 0)                      # exactly one byte set
        n = np.stack([np.count_nonzero(alone & (w == (1 << (8 * s))), axis=1) for s in range(4)], axis=1)
    else:
        alone = u.sum(axis=2, dtype=np.uint8) == 1
        n = (u & alone[..., None].view(np.uint8)).sum(axis=1, dtype=np.int64)
    return n >= min_frames


def active_speakers(seg: np.ndarray) -> np.ndarray:
    """[C, L, S] -> bool [C, S]: the local speaker is active in at least one frame (`np.sum(seg, axis=1) > 0`,
    PA/pipelines/clustering.py:111-131, diarizen/pipelines/inference.py:160).  u8 decisions with S == 4 are OR-ed as one
    32-bit word per frame: 5 ms instead of 60 ms for the strided byte reduction at 4 h (17 991 x 399 x 4)."""
    if seg.dtype != np.uint8:
        return np.sum(seg, axis=1) > 0
    C, L, S = seg.shape
    if S == 4 and L > 0:
        w = np.ascontiguousarray(seg).view(np.uint32).reshape(C, L)
        return np.bitwise_or.reduce(w, axis=1).view(np.uint8).reshape(C, 4) != 0
    return seg.any(axis=1)


def filter_embeddings(embeddings: np.ndarray, seg: np.ndarray, min_frames_ratio: float = 0.1,
                      max_num_embeddings: float = np.inf):
    active = active_speakers(seg)
    valid = ~np.any(np.isnan(embeddings), axis=2)
    min_frames = round(min_frames_ratio * seg.shape[1])
    keep = active * valid * single_speaker_frame_mask(seg, min_frames)
    ci, si = np.where(keep)
    if len(ci) < 2:                       # too short / fully overlapped input: relax the frame rule
        ci, si = np.where(active * valid * single_speaker_frame_mask(seg, 0))
    n = len(ci)
    if n > max_num_embeddings:
        idx = list(range(n))
        random.shuffle(idx)
        idx = sorted(idx[: int(max_num_embeddings)])
        ci, si = ci[idx], si[idx]
    return embeddings[ci, si], ci, si


def constrained_argmax(soft: np.ndarray) -> np.ndarray:
    """per window: Hungarian assignment of local speakers to clusters (maximise similarity)."""
    soft = np.nan_to_num(soft, nan=np.nanmin(soft))
    C, S, K = soft.shape
    hard = -2 * np.ones((C, S), dtype=np.int8)
    todo = np.arange(C)
    if K >= S and C > 64:
        # Vectorised exact shortcut (row f1: 17 991 windows at 4 h): when every local speaker's best cluster is
        # STRICTLY best and the S best clusters are distinct, that assignment attains the upper bound sum_s max_k and is
        # the unique optimum, so the Hungarian solver must return it.  Everything else (ties, conflicts — e.g. several
        # inactive speakers sharing the bias embedding) goes through scipy as in the reference.
        best = soft.argmax(axis=2)                                            # [C, S]
        part = np.partition(soft, K - 2, axis=2) if K > 1 else None
        strict = (part[:, :, K - 1] > part[:, :, K - 2]).all(axis=1) if K > 1 else np.ones(C, bool)
        srt = np.sort(best, axis=1)
        distinct = (srt[:, 1:] != srt[:, :-1]).all(axis=1) if S > 1 else np.ones(C, bool)
        easy = strict & distinct
        hard[easy] = best[easy].astype(np.int8)
        todo = np.nonzero(~easy)[0]
    for c in todo:
        spk, clu = linear_sum_assignment(soft[c], maximize=True)
        hard[c, spk] = clu
    return hard


HIP_CDIST_MIN = 8192     # rows; below this scipy's single-core loop is faster than upload + launch


def _soft_clusters(embeddings: np.ndarray, centroids: np.ndarray, metric: str, backend: str = "auto",
                   device: int = -1, active: Optional[np.ndarray] = None) -> np.ndarray:
    """2 - cdist(embeddings, centroids) (PA/pipelines/clustering.py:207-216).  From HIP_CDIST_MIN rows up, float32
    embeddings and the cosine metric go through csrc/linkage.hip's dzn_cdist_cosine (float64, same formula, in-order
    sums: agrees with scipy to 2e-15 and keeps identical rows identical, tests/test_ops_gpu.py); anything else is
    scipy's own call."""
    C, S, D = embeddings.shape
    flat = embeddings.reshape(C * S, D)
    if (metric == "cosine" and backend != "scipy" and flat.dtype == np.float32
            and (backend == "hip" or (C * S >= HIP_CDIST_MIN and _hip_ready()))):
        from . import ops
        from ._lib import DznError
        try:
            soft = 2 - ops.cdist_cosine(flat, centroids, device=device).reshape(C, S, -1)
        except (DznError, MemoryError):
            if backend == "hip":
                raise
        else:
            _exact_scores_for_tied_rows(embeddings, centroids, metric, soft, active)
            return soft
    return 2 - cdist(flat, centroids, metric=metric).reshape(C, S, -1)


def _exact_scores_for_tied_rows(embeddings: np.ndarray, centroids: np.ndarray, metric: str, soft: np.ndarray,
                                active: Optional[np.ndarray] = None) -> None:
    """(r5) Local speakers of ONE window with bit-identical embeddings — every inactive speaker and every speaker whose mask
    samples to all-zero pooling weights gets seg_1's bias, and two speakers that are only ever active together
    share a mask — make the window's constrained assignment an exact tie, which `linear_sum_assignment` breaks by the LAST BIT
    of the scores (tests/golden/host30.npz: a 1e-15 perturbation of scipy's own scores moves ~330 active assignments, and an
    active few-frame speaker that ties with an inactive one then lands in another cluster, which the RTTM shows).  The
    reference's bits are scipy.cdist's, and scipy.cdist is row-independent (bit-identical for a row whatever else is in the
    call), so the rows that have an identical twin in their window get scipy's own scores, computed once per distinct vector:
    a few hundred rows at 30 min.  Windows in which a tied row belongs to an ACTIVE speaker (`active` [C, S]: the tie decides
    where real activity goes) take scipy's scores for ALL their rows, so that the assignment solver sees exactly the
    reference's matrix there (16 % of the rows of the 30-min fixture).  Every other row keeps the device's float64 scores
    (2e-15 from scipy's, against margins of 1e-6).  In place."""
    C, S, D = embeddings.shape
    twin = np.zeros((C, S), dtype=bool)
    for i in range(S):
        for j in range(i + 1, S):
            eq = (embeddings[:, i] == embeddings[:, j]).all(axis=-1)
            twin[:, i] |= eq
            twin[:, j] |= eq
    ci, si = np.nonzero(twin)
    if not len(ci):
        return
    rows = np.ascontiguousarray(embeddings[ci, si])
    keys = rows.view(np.dtype((np.void, rows.dtype.itemsize * D))).ravel()
    _, first, inv = np.unique(keys, return_index=True, return_inverse=True)
    exact = 2 - cdist(rows[first], centroids, metric=metric)
    soft[ci, si] = exact[inv]
    if active is not None:
        win = np.nonzero((twin & np.asarray(active, dtype=bool)).any(axis=1))[0]
        if len(win):
            soft[win] = 2 - cdist(embeddings[win].reshape(len(win) * S, D), centroids, metric=metric).reshape(len(win), S, -1)


def _set_num_clusters(n, num_clusters, min_clusters, max_clusters):
    lo = num_clusters or min_clusters or 1
    lo = max(1, min(n, lo))
    hi = num_clusters or max_clusters or n
    hi = max(1, min(n, hi))
    if lo > hi:
        raise ValueError(f"min_clusters must be smaller than (or equal to) max_clusters "
                         f"(here: min_clusters={lo:g} and max_clusters={hi:g}).")
    if lo == hi:
        num_clusters = lo
    return num_clusters, lo, hi


# --------------------------------------------------------------------------- AHC
class _DeviceBackends:
    """where the optional device accelerators of the host stage run; DiariZenPipeline sets `device` to its own HIP device.
    Each backend is "auto" (device from a size threshold up when a HIP device is present), "hip", or the host library."""
    device: int = -1              # HIP device ordinal for csrc/linkage.hip / vbx.hip (-1 = current)
    linkage_backend: str = "auto"  # "scipy" = scipy.cluster.hierarchy.linkage
    cdist_backend: str = "auto"    # "scipy" = scipy.spatial.distance.cdist
    vbx_backend: str = "auto"      # "numpy" = the reference-shaped loop in vb_gmm


class AgglomerativeClustering(_DeviceBackends):
    def __init__(self, metric: str = "cosine", max_num_embeddings: float = np.inf,
                 constrained_assignment: bool = True, method: str = "centroid", threshold: float = 0.6,
                 min_cluster_size: int = 13, linkage_backend: str = "auto"):
        self.linkage_backend = linkage_backend
        self.metric, self.max_num_embeddings = metric, max_num_embeddings
        self.constrained_assignment = constrained_assignment
        self.method, self.threshold, self.min_cluster_size = method, threshold, min_cluster_size

    def cluster(self, emb: np.ndarray, min_clusters: int, max_clusters: int,
                num_clusters: Optional[int] = None) -> np.ndarray:
        n = emb.shape[0]
        min_size = min(self.min_cluster_size, max(1, round(0.1 * n)))
        if n == 1:
            return np.zeros((1,), dtype=np.uint8)
        if self.metric == "cosine" and self.method in ("centroid", "median", "ward"):
            with np.errstate(divide="ignore", invalid="ignore"):
                emb /= np.linalg.norm(emb, axis=-1, keepdims=True)      # in place, like the reference
            dendro = (centroid_linkage(emb, self.linkage_backend, self.device) if self.method == "centroid"
                      else linkage(emb, method=self.method, metric="euclidean"))
        else:
            dendro = linkage(emb, method=self.method, metric=self.metric)
        clusters = _fcluster_distance(dendro, self.threshold) - 1

        def large_of(cl):
            ids, cnt = np.unique(cl, return_counts=True)
            return ids, cnt, ids[cnt >= min_size]

        ids, cnt, large = large_of(clusters)
        n_large = len(large)
        if n_large < min_clusters:
            num_clusters = min_clusters
        elif n_large > max_clusters:
            num_clusters = max_clusters
        if num_clusters is not None and n_large != num_clusters:
            # walk the dendrogram away from the threshold until the number of large clusters fits
            by_iter = np.copy(dendro)
            by_iter[:, 2] = np.arange(n - 1)
            best_it, best_n = n - 1, 1
            for it in np.argsort(np.abs(dendro[:, 2] - self.threshold)):
                if by_iter[it, 3] < min_size:
                    continue
                clusters = fcluster(by_iter, it, criterion="distance") - 1
                ids, cnt, large = large_of(clusters)
                n_large = len(large)
                if abs(n_large - num_clusters) < abs(best_n - num_clusters):
                    best_it, best_n = it, n_large
                if n_large == num_clusters:
                    break
            if best_n != num_clusters:
                clusters = fcluster(by_iter, best_it, criterion="distance") - 1
                ids, cnt, large = large_of(clusters)
                n_large = len(large)
        if n_large == 0:
            clusters[:] = 0
            return clusters
        small = ids[cnt < min_size]
        if len(small) == 0:
            return clusters
        big_c = np.vstack([np.mean(emb[clusters == k], axis=0) for k in large])
        small_c = np.vstack([np.mean(emb[clusters == k], axis=0) for k in small])
        nearest = np.argmin(cdist(big_c, small_c, metric=self.metric), axis=0)
        for j, b in enumerate(nearest):
            clusters[clusters == small[j]] = large[b]
        _, clusters = np.unique(clusters, return_inverse=True)
        return clusters

    def __call__(self, embeddings: np.ndarray, segmentations: np.ndarray,
                 num_clusters: Optional[int] = None, min_clusters: Optional[int] = None,
                 max_clusters: Optional[int] = None):
        train, ci, si = filter_embeddings(embeddings, segmentations,
                                          max_num_embeddings=self.max_num_embeddings)
        C, S, _ = embeddings.shape
        num_clusters, lo, hi = _set_num_clusters(train.shape[0], num_clusters, min_clusters, max_clusters)
        if hi < 2:
            return (np.zeros((C, S), dtype=np.int8), np.ones((C, S, 1)),
                    np.mean(train, axis=0, keepdims=True))
        train_clusters = self.cluster(train, lo, hi, num_clusters=num_clusters)
        K = int(np.max(train_clusters)) + 1
        train = embeddings[ci, si]
        centroids = np.vstack([np.mean(train[train_clusters == k], axis=0) for k in range(K)])
        soft = _soft_clusters(embeddings, centroids, self.metric, self.cdist_backend,
                              self.device, active=active_speakers(segmentations))
        hard = constrained_argmax(soft) if self.constrained_assignment else np.argmax(soft, axis=2)
        return hard, soft, centroids


# --------------------------------------------------------------------------- VBx
HIP_VBX_MIN = 4096       # embeddings; below this numpy's BLAS calls are faster than upload + launches


def vb_gmm(X: np.ndarray, Phi: np.ndarray, gamma: np.ndarray, Fa: float, Fb: float, max_iters: int,
           epsilon: float = 1e-4, backend: str = "auto", device: int = -1):
    """VBx with loopProb = 0 (the only branch the pipeline exercises, VBx.py:99-107): variational
    Bayes mixture over speakers in PLDA space; returns responsibilities and speaker priors.
    backend "numpy": the reference's expressions.  "hip" / "auto" (from HIP_VBX_MIN embeddings up): the two E-sized
    passes of every iteration (gamma^T rho and the E-step) run on the device in float64 (csrc/vbx.hip) while invL,
    alpha, the ELBO and the stopping rule stay the expressions below on the K x D statistics."""
    if backend not in ("auto", "numpy", "hip"):
        raise ValueError(f"unknown VBx backend {backend!r}")
    if backend == "hip" or (backend == "auto" and X.shape[0] >= HIP_VBX_MIN and _hip_ready()):
        from ._lib import DznError
        try:
            return _vb_gmm_hip(X, Phi, gamma, Fa, Fb, max_iters, epsilon, device)
        except (DznError, MemoryError):
            if backend == "hip":
                raise
    D = X.shape[1]
    pi = np.ones(gamma.shape[1]) / gamma.shape[1]
    G = -0.5 * (np.sum(X ** 2, axis=1, keepdims=True) + D * np.log(2 * np.pi))
    rho = X * np.sqrt(Phi)
    prev = None
    for it in range(max_iters):
        invL = 1.0 / (1 + Fa / Fb * gamma.sum(axis=0, keepdims=True).T * Phi)
        alpha = Fa / Fb * invL * gamma.T.dot(rho)
        log_p = Fa * (rho.dot(alpha.T) - 0.5 * (invL + alpha ** 2).dot(Phi) + G)
        lpi = np.log(pi + 1e-8)
        log_px = logsumexp(log_p + lpi, axis=-1)
        total = np.sum(log_px, axis=0)
        gamma = np.exp(log_p + lpi - log_px[:, None])
        pi = np.sum(gamma, axis=0)
        pi = pi / pi.sum()
        elbo = total + Fb * 0.5 * np.sum(np.log(invL) - invL - alpha ** 2 + 1)
        if it > 0 and elbo - prev < epsilon:
            break
        prev = elbo
    return gamma, pi


def _vb_gmm_hip(X, Phi, gamma, Fa, Fb, max_iters, epsilon, device):
    """the loop of vb_gmm with `gamma.T.dot(rho)`, `gamma.sum(0)` (VbxState.stats) and the E-step (VbxState.estep)
    on the device; everything K x D sized is computed here exactly as above."""
    from . import ops
    D = X.shape[1]
    Phi = np.asarray(Phi, dtype=np.float64)
    st = ops.VbxState(X, Phi, gamma, device)
    try:
        pi = np.ones(gamma.shape[1]) / gamma.shape[1]
        stats = st.stats()
        prev = None
        for it in range(max_iters):
            invL = 1.0 / (1 + Fa / Fb * stats[:, D:D + 1] * Phi)
            alpha = Fa / Fb * invL * stats[:, :D]
            ck = 0.5 * (invL + alpha ** 2).dot(Phi)
            total = st.estep(alpha, ck, np.log(pi + 1e-8), Fa)
            stats = st.stats()
            pi = stats[:, D].copy()
            pi = pi / pi.sum()
            elbo = total + Fb * 0.5 * np.sum(np.log(invL) - invL - alpha ** 2 + 1)
            if it > 0 and elbo - prev < epsilon:
                break
            prev = elbo
        return st.gamma(), pi
    finally:
        st.close()


def _l2n(x):
    return x / np.linalg.norm(x, axis=1, ord=2)[:, np.newaxis]


def load_plda(plda_dir: str):
    """vbx_setup (VBx.py:158-194): x-vector transform + PLDA simultaneous diagonalisation."""
    x = np.load(f"{plda_dir}/xvec_transform.npz")
    mean1, mean2, lda = x["mean1"], x["mean2"], x["lda"]
    p = np.load(f"{plda_dir}/plda.npz")
    mu, tr, psi = p["mu"], p["tr"], p["psi"]
    W = np.linalg.inv(tr.T.dot(tr))
    Bm = np.linalg.inv((tr.T / psi).dot(tr))
    acvar, wccn = eigh(Bm, W)
    psi = acvar[::-1]
    tr = wccn.T[::-1]

    def xvec_tf(v):
        return np.sqrt(lda.shape[1]) * _l2n(lda.T.dot(np.sqrt(lda.shape[0]) * _l2n(v - mean1).T).T - mean2)

    def plda_tf(v, lda_dim=lda.shape[1]):
        return (v - mu).dot(tr.T)[:, :lda_dim]

    return xvec_tf, plda_tf, psi


class VBxClustering(_DeviceBackends):
    def __init__(self, metric: str = "cosine", max_num_embeddings: float = np.inf,
                 constrained_assignment: bool = True, plda_dir: str = "", lda_dim: int = 128,
                 max_iters: int = 20, ahc_criterion: str = "distance", ahc_threshold: float = 0.6,
                 Fa: float = 0.07, Fb: float = 0.8, linkage_backend: str = "auto"):
        self.linkage_backend = linkage_backend
        self.metric, self.max_num_embeddings = metric, max_num_embeddings
        self.constrained_assignment = constrained_assignment
        self.plda_dir, self.lda_dim, self.max_iters = plda_dir, lda_dim, max_iters
        self.ahc_criterion, self.ahc_threshold, self.Fa, self.Fb = ahc_criterion, ahc_threshold, Fa, Fb
        self._plda = None

    def __call__(self, embeddings: np.ndarray, segmentations: np.ndarray, num_clusters=None,
                 min_clusters=None, max_clusters=None):
        train, _, _ = filter_embeddings(embeddings, segmentations, 0.1, self.max_num_embeddings)
        C, S, D = embeddings.shape
        if train.shape[0] < 2:
            return (np.zeros((C, S), dtype=np.int8), np.ones((C, S, 1)),
                    np.mean(train, axis=0, keepdims=True))
        normed = train / np.linalg.norm(train, axis=1, keepdims=True)
        dendro = centroid_linkage(normed, self.linkage_backend, self.device)
        ahc = fcluster(dendro, self.ahc_threshold, criterion=self.ahc_criterion) - 1
        _, ahc = np.unique(ahc, return_inverse=True)
        if self._plda is None:
            self._plda = load_plda(self.plda_dir)
        xvec_tf, plda_tf, psi = self._plda
        fea = plda_tf(xvec_tf(train), lda_dim=self.lda_dim)
        Phi = psi[: self.lda_dim]
        q0 = np.zeros((len(ahc), ahc.max() + 1))
        q0[range(len(ahc)), ahc.astype(int)] = 1.0
        q0 = softmax(q0 * 7.0, axis=1)                                  # init_smoothing = 7
        q, sp = vb_gmm(fea, Phi, q0, self.Fa, self.Fb, self.max_iters, backend=self.vbx_backend,
                       device=self.device)
        centroids = q[:, sp > 1e-7].T @ train.reshape(-1, D)            # unnormalised: cosine follows
        soft = _soft_clusters(embeddings, centroids, self.metric, self.cdist_backend,
                              self.device, active=active_speakers(segmentations))
        hard = constrained_argmax(soft) if self.constrained_assignment else np.argmax(soft, axis=2)
        _, hard = np.unique(hard, return_inverse=True)
        return hard.reshape(C, S), soft, centroids
