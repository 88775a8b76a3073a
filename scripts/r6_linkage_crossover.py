"""Where the device centroid linkage overtakes scipy's host loop (clustering.HIP_LINKAGE_MIN), after r6's two-neighbour step loop:
python scripts/r6_linkage_crossover.py   -> ms per call at n = 256 .. 4096, dendrograms compared."""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from scipy.cluster.hierarchy import linkage
from diarizen_amd import ops
from oracle.gen_golden import linkage_scale_case
ops.linkage_centroid(linkage_scale_case(n=512, dim=256, K=6, seed=1))      # context + arena
for n in (256, 384, 512, 768, 1024, 1536, 2048, 3072, 4096):
    e = linkage_scale_case(n=n, dim=256, K=8, seed=n)
    td, ts = [], []
    for rep in range(3):
        t = time.perf_counter(); Zd = ops.linkage_centroid(e); td.append(time.perf_counter() - t)
        t = time.perf_counter(); Zs = linkage(e, method="centroid", metric="euclidean"); ts.append(time.perf_counter() - t)
    same = np.array_equal(Zd[:, [0, 1, 3]], Zs[:, [0, 1, 3]])
    print(f"n={n:5d} device {min(td) * 1e3:8.2f} ms  scipy {min(ts) * 1e3:8.2f} ms  same merges: {same}", flush=True)
