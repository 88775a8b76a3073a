"""GPU-box diagnostic for tests/golden/host30.npz: which device backend of the host stage departs from the scipy path, and where.
    python scripts/diag_host30.py > gpurun_out/r5_diag_host30.txt"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from scipy.cluster.hierarchy import fcluster, linkage  # noqa: E402
from scipy.spatial.distance import cdist  # noqa: E402

from diarizen_amd import clustering as cl, ops  # noqa: E402

g = np.load(ROOT / "tests" / "golden" / "host30.npz")
seg, emb, ref = g["seg"], g["emb"], g["hard_clusters"].astype(np.int64)
train, ci, si = cl.filter_embeddings(emb, seg)
e = np.array(train, copy=True)
e /= np.linalg.norm(e, axis=-1, keepdims=True)
print("training rows", e.shape, e.dtype, "duplicate rows:", cl._has_duplicate_rows(e))
Zs = linkage(e, method="centroid", metric="euclidean")
Zh = ops.linkage_centroid(e, device=0)
same_ids = (Zs[:, :2].astype(np.int64) == Zh[:, :2].astype(np.int64)).all(axis=1)
print("merges with identical ids:", int(same_ids.sum()), "of", len(Zs), "; max |ddist| over them", float(np.abs(Zs[same_ids, 2] - Zh[same_ids, 2]).max()))
bad = np.nonzero(~same_ids)[0]
if len(bad):
    k = int(bad[0])
    print("first differing merge", k, "scipy", Zs[k], "hip", Zh[k])
    for j in range(max(0, k - 2), min(len(Zs), k + 4)):
        print("   ", j, "scipy", Zs[j, :3], "hip", Zh[j, :3], "ddist", Zs[j, 2] - Zh[j, 2])
    # inversions: centroid linkage is not monotone; how close are consecutive distances around k
    print("   gap of scipy distances around k:", np.diff(Zs[max(0, k - 2):k + 4, 2]))
for thr in (0.1,):
    cs, ch = fcluster(Zs, thr, criterion="distance"), fcluster(Zh, thr, criterion="distance")
    print("flat clusters at", thr, ": scipy", len(np.unique(cs)), "hip", len(np.unique(ch)), "rows in a different partition:",
          int((cs != ch).sum()))
res = {}
for lb in ("scipy", "hip"):
    for cb in ("scipy", "hip"):
        a = cl.AgglomerativeClustering(metric="cosine", method="centroid", min_cluster_size=13, threshold=0.1, linkage_backend=lb)
        a.cdist_backend, a.device = cb, 0
        hard, soft, cent = a(embeddings=emb, segmentations=seg, min_clusters=1, max_clusters=20)
        res[(lb, cb)] = (np.asarray(hard).astype(np.int64), soft)
        print(f"linkage={lb:5s} cdist={cb:5s}: clusters {int(np.max(hard)) + 1}, entries != reference golden: {int((hard != ref).sum())}")
s_s, s_h = res[("scipy", "scipy")][1], res[("scipy", "hip")][1]
print("soft scores scipy vs hip cdist (same centroids): max |d|", float(np.nanmax(np.abs(s_s - s_h))))
srt = np.sort(s_s.reshape(-1, s_s.shape[-1]), axis=1)
print("smallest top-2 margin of the scipy scores", float((srt[:, -1] - srt[:, -2]).min()))

# ---- where the host stage's time goes on this box (device backends on) ----
import cProfile
import pstats
import time
from diarizen_amd.core import SlidingWindow
from diarizen_amd.pipeline import run_host_stage

a = cl.AgglomerativeClustering(metric="cosine", method="centroid", min_cluster_size=13, threshold=0.1)
a.device = 0
dev = torch.device("cuda:0")


def once():
    ann = run_host_stage(seg, emb, chunks=SlidingWindow(start=0.0, duration=8.0, step=0.8), clustering=a, min_speakers=1,
                         max_speakers=20, sess_name="x", device=dev)
    return ann.to_rttm()


once()
for _ in range(3):
    t = time.perf_counter()
    once()
    print("host stage + RTTM text, s:", round(time.perf_counter() - t, 4))
pr = cProfile.Profile()
pr.enable()
once()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
st.sort_stats("cumulative").print_stats(30)
