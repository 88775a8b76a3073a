"""Device centroid linkage, one launch per step (r3) against the two-kernel loop of r2 (DZN_LINKAGE_TWO_KERNEL=1):
python scripts/bench_linkage.py [n ...]   -> ms per call and us per merge, dendrograms compared."""
import os
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from diarizen_amd import ops
from oracle.gen_golden import linkage_scale_case

ns = [int(a) for a in sys.argv[1:]] or [5000, 20000, 35790]
for n in ns:
    e = linkage_scale_case(n=n, dim=256, K=12, seed=5)
    res = {}
    for mode in ("persist", "step", "top1", "two_kernel", "persist", "step", "top1"):     # persist = r6b, one persistent launch (opt-in); step = r6 default
        os.environ.pop("DZN_LINKAGE_TWO_KERNEL", None)
        os.environ.pop("DZN_LINKAGE_TOP1", None)
        os.environ.pop("DZN_LINKAGE_PERSIST", None)
        if mode == "persist":
            os.environ["DZN_LINKAGE_PERSIST"] = "1"
        if mode == "two_kernel":
            if n > 12000:
                continue
            os.environ["DZN_LINKAGE_TWO_KERNEL"] = "1"
        if mode == "top1":
            os.environ["DZN_LINKAGE_TOP1"] = "1"
        t = time.perf_counter()
        Z = ops.linkage_centroid(e)
        dt = time.perf_counter() - t
        res.setdefault(mode, []).append((dt, Z))
        print(f"n={n:6d} {mode:10s} {dt * 1e3:9.1f} ms  {dt / (n - 1) * 1e6:6.1f} us/merge", flush=True)
    a, b = res["step"][0][1], res["top1"][0][1]
    print(f"n={n:6d} dendrograms identical (persistent launch vs step loop): {np.array_equal(res['persist'][0][1], a)}")
    print(f"n={n:6d} dendrograms identical (two neighbours vs one): {np.array_equal(a, b)}"
          + (f", vs the two-kernel loop: {np.array_equal(a, res['two_kernel'][0][1])}" if "two_kernel" in res else ""))
