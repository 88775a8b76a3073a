"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product (diarizen_amd/).

Plain restatement of the reference's agglomerative clustering step for the boxes that have no /root/reference (the GPU box's
`cpu_baseline` leg of bench.py): `BaseClustering.filter_embeddings` (PA/pipelines/clustering.py:111-157 with
`filter_embeddings_by_frames`), `AgglomerativeClustering.cluster` (:394-513) WITHOUT the forced min / max / num_clusters
dendrogram walk (:434-482 — not needed at min_clusters = 1, max_clusters = 20; asked for, it raises), `assign_embeddings`
(:175-245) with `constrained_argmax` (:159-173).  Same scipy calls as the reference, in the same order.
Pinned: tests/test_oracle.py::test_clustering_port_equals_reference_run_at_30min_scale holds it to tests/golden/host30.npz
(hard clusters produced by the reference's own class on 8964 rows).
"""
from __future__ import annotations

import numpy as np
from scipy.cluster.hierarchy import fcluster, linkage
from scipy.optimize import linear_sum_assignment
from scipy.spatial.distance import cdist


def _single_speaker_frame_mask(seg: np.ndarray, min_frames: int) -> np.ndarray:
    """filter_embeddings_by_frames: speakers with at least `min_frames` frames in which they are the only active one"""
    single = (np.sum(seg, axis=2, keepdims=True) == 1)
    return np.sum(seg * single, axis=1) >= min_frames


def filter_embeddings(emb: np.ndarray, seg: np.ndarray, min_frames_ratio: float = 0.1):
    active = np.sum(seg, axis=1) > 0                                         # :133
    valid = ~np.any(np.isnan(emb), axis=2)                                   # :136
    min_frames = round(min_frames_ratio * seg.shape[1])                      # :139
    mask = _single_speaker_frame_mask(seg, min_frames)
    c, s = np.where(active * valid * mask)
    if len(c) < 2:                                                           # :143-146
        c, s = np.where(active * valid * _single_speaker_frame_mask(seg, 0))
    return emb[c, s], c, s


def cluster(train: np.ndarray, threshold: float, min_cluster_size: int, min_clusters: int, max_clusters: int) -> np.ndarray:
    n = len(train)
    mcs = min(min_cluster_size, max(1, round(0.1 * n)))                      # :398-400
    if n == 1:
        return np.zeros((1,), dtype=np.uint8)
    e = train / np.linalg.norm(train, axis=-1, keepdims=True)                # :409-411 (cosine + centroid)
    Z = linkage(e, method="centroid", metric="euclidean")
    clusters = fcluster(Z, threshold, criterion="distance") - 1              # :422
    uniq, counts = np.unique(clusters, return_counts=True)
    large = uniq[counts >= mcs]
    if len(large) < min_clusters or len(large) > max_clusters:
        raise NotImplementedError("the forced-cluster-count dendrogram walk (clustering.py:434-482) is not restated here")
    if len(large) == 0:                                                      # :484-486
        clusters[:] = 0
        return clusters
    small = uniq[counts < mcs]
    if len(small) == 0:
        return clusters
    lc = np.vstack([np.mean(e[clusters == k], axis=0) for k in large])       # :493-505
    sc = np.vstack([np.mean(e[clusters == k], axis=0) for k in small])
    d = cdist(lc, sc, metric="cosine")
    for si, li in enumerate(np.argmin(d, axis=0)):
        clusters[clusters == small[si]] = large[li]
    _, clusters = np.unique(clusters, return_inverse=True)                   # :511-512
    return clusters


def agglomerative(emb: np.ndarray, seg: np.ndarray, threshold: float, min_cluster_size: int, min_clusters: int = 1,
                  max_clusters: int = 20) -> np.ndarray:
    """emb [C, S, D] float, seg [C, L, S] {0, 1}  ->  hard clusters [C, S] int8 (BaseClustering.__call__, :247-322)"""
    segf = np.asarray(seg, dtype=np.float32)
    train, c, s = filter_embeddings(emb, segf)
    max_clusters = min(max_clusters, len(train))                             # set_num_clusters
    if max_clusters < 2:
        return np.zeros(emb.shape[:2], dtype=np.int8)
    tc = cluster(np.array(train, copy=True), threshold, min_cluster_size, min_clusters, max_clusters)
    K = int(np.max(tc)) + 1
    tr = emb[c, s]
    cent = np.vstack([np.mean(tr[tc == k], axis=0) for k in range(K)])      # :207-212
    C, S, D = emb.shape
    soft = 2 - cdist(emb.reshape(C * S, D), cent, metric="cosine").reshape(C, S, K)
    soft = np.nan_to_num(soft, nan=np.nanmin(soft))                          # :160
    hard = -2 * np.ones((C, S), dtype=np.int8)
    for ci, cost in enumerate(soft):                                         # :168-171
        sp, cl = linear_sum_assignment(cost, maximize=True)
        for a, b in zip(sp, cl):
            hard[ci, a] = b
    return hard
