import sys, time
from pathlib import Path
sys.path.insert(0, "/root/repo" if not Path("diarizen_amd").exists() else ".")
import numpy as np
from scipy.cluster.hierarchy import fcluster, linkage
from diarizen_amd import ops
from oracle.gen_golden import synth_host_case
for C in (100, 800, 2241, 4491):
    seg, emb = synth_host_case(3, C=C, L=99, n_spk=4)
    act = seg.sum(1) > 0
    e = emb[act].astype(np.float32)
    e /= np.linalg.norm(e, axis=-1, keepdims=True)
    t0 = time.perf_counter(); Zs = linkage(e, method="centroid", metric="euclidean"); ts = time.perf_counter() - t0
    t0 = time.perf_counter(); Zg = ops.linkage_centroid(e); tg = time.perf_counter() - t0
    same_pairs = np.array_equal(Zs[:, [0, 1, 3]], Zg[:, [0, 1, 3]])
    dd = np.abs(Zs[:, 2] - Zg[:, 2]).max()
    fs, fg = fcluster(Zs, 0.7, "distance"), fcluster(Zg, 0.7, "distance")
    print(f"n={len(e)} scipy {ts:.3f}s hip {tg:.3f}s pairs_equal={same_pairs} max|dist diff|={dd:.2e} flat_equal={np.array_equal(fs, fg)} k={fs.max()}", flush=True)
