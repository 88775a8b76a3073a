cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py -q -k "vbx or cdist or backends" 2>&1 | tail -15 | cut -c1-300
timeout 300 python - <<'PY'
import time, numpy as np, sys
sys.path.insert(0, '.')
from diarizen_amd import clustering as cl
from oracle.gen_golden import synth_vbx_case
for E, K in ((8000, 12), (60000, 13), (60000, 60)):
    X, Phi, q0 = synth_vbx_case(E, K)
    cl.vb_gmm(X[:4096], Phi, q0[:4096].copy(), 0.07, 0.8, 2, backend="hip")
    t = time.perf_counter(); a = cl.vb_gmm(X, Phi, q0.copy(), 0.07, 0.8, 20, backend="numpy"); tn = time.perf_counter() - t
    t = time.perf_counter(); b = cl.vb_gmm(X, Phi, q0.copy(), 0.07, 0.8, 20, backend="hip"); th = time.perf_counter() - t
    print(f"VBX_TIMING E={E} K={K}: numpy {tn:.3f} s, hip {th:.3f} s, max|dgamma| {np.abs(a[0]-b[0]).max():.2e}")
PY
