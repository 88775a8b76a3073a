"""centroid linkage: scipy (host) vs csrc/linkage.hip at growing recording lengths."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from scipy.cluster.hierarchy import fcluster, linkage
from diarizen_amd import ops
from oracle.gen_golden import synth_host_case
for minutes, with_scipy in ((30, True), (60, True), (120, False), (240, False)):
    C = int((minutes * 60 - 8.0) / 0.8) + 1
    seg, emb = synth_host_case(3, C=C, L=99, n_spk=4)
    e = emb[seg.sum(1) > 0].astype(np.float32)
    e /= np.linalg.norm(e, axis=-1, keepdims=True)
    t0 = time.perf_counter(); Zg = ops.linkage_centroid(e); tg = time.perf_counter() - t0
    msg = f"{minutes} min: n={len(e)} hip {tg:.2f}s D={8*len(e)**2/1e9:.1f} GB"
    if with_scipy:
        t0 = time.perf_counter(); Zs = linkage(e, method="centroid", metric="euclidean"); ts = time.perf_counter() - t0
        msg += f" scipy {ts:.2f}s equal={np.array_equal(Zs, Zg)}"
    print(msg, "k=", fcluster(Zg, 0.7, "distance").max(), flush=True)
