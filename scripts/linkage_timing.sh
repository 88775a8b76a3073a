# csrc/linkage.hip: scipy equality tests + device timing at 30 min / 4 h scale (synthetic embeddings, 4 speakers)
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_ops_gpu.py -q -k "linkage or backends" 2>&1 | tail -4
timeout 300 python - <<'PY'
import sys, time
sys.path.insert(0, '.')
import numpy as np
from scipy.cluster.hierarchy import linkage
from diarizen_amd import ops
from oracle.gen_golden import synth_host_case
for minutes in (30, 240):
    C = int((minutes * 60 - 8.0) / 0.8) + 1
    seg, emb = synth_host_case(3, C=C, L=99, n_spk=4)
    e = emb[seg.sum(1) > 0].astype(np.float32)
    e /= np.linalg.norm(e, axis=-1, keepdims=True)
    ops.linkage_centroid(e[:3000])
    t0 = time.perf_counter(); Zg = ops.linkage_centroid(e); tg = time.perf_counter() - t0
    print(f"LINKAGE {minutes} min: n={len(e)} device {tg:.3f} s", flush=True)
    if minutes == 30:
        Zs = linkage(e, method="centroid", metric="euclidean")
        print("  == scipy:", np.array_equal(Zs[:, [0, 1, 3]], Zg[:, [0, 1, 3]]), float(np.abs(Zs[:, 2] - Zg[:, 2]).max()))
PY
