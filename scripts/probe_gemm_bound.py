"""What bounds the split contraction?  One process, one shape family, every kernel form (DZN_GEMM_CFG names):
    python scripts/probe_gemm_bound.py cfg1,cfg2 [quick]
variants: base (random data, residual) / zeroA (all-zero activations: same instructions, far less switching power) /
noR (no residual read) / noRC... / M/8 (operands L2 + MALL resident) / longK (K = 4096 at the same A bytes: the loop alone)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from diarizen_amd import _lib, ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
lib = _lib.load()
cfgs = sys.argv[1].split(",")
quick = len(sys.argv) > 2


def run(cfg, tag, M, N, K, zero=False, resid=True, it=8):
    A = torch.zeros(M, K, device=dev) if zero else torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) * 0.05
    R = torch.randn(M, N, device=dev) if resid else None
    W3 = ops.split_weights(W)
    W2h, cs = ops.split_weights_h2(W)
    am = ops.amax(A) if not zero else torch.ones(1, device=dev)
    out = torch.empty(M, N, device=dev)
    lib.dzn_op_set_gemm_cfg(cfg.encode())
    kw = {"W3": W3, "W2h": W2h, "col_scale": cs, "a_amax": am}
    for _ in range(2):
        ops.gemm(A, W, C_out=out, R=R, precision=3, **kw)
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(it):
        ops.gemm(A, W, C_out=out, R=R, precision=3, **kw)
    en.record(); torch.cuda.synchronize()
    dt = st.elapsed_time(en) / it * 1e-3
    gb = (M * K * 4 + M * N * 4 * (2 if resid else 1)) / 1e9
    print(f"cfg={cfg:10s} {tag:8s} M={M} N={N} K={K}: {dt*1e6:8.1f} us {2*M*N*K/dt/1e12:6.1f} TF/s  min-bytes {gb:5.2f} GB -> {gb/dt/1e3:5.2f} TB/s",
          flush=True)


for cfg in cfgs:
    run(cfg, "base", 149226, 1024, 1024)
    if quick:
        continue
    run(cfg, "zeroA", 149226, 1024, 1024, zero=True)
    run(cfg, "noR", 149226, 1024, 1024, resid=False)
    run(cfg, "M/8", 18653, 1024, 1024, it=32)
    run(cfg, "M/8noR", 18653, 1024, 1024, resid=False, it=32)
    run(cfg, "longK", 37306, 1024, 4096)
    run(cfg, "longKnoR", 37306, 1024, 4096, resid=False)
    run(cfg, "N128", 149226, 128, 1024)
lib.dzn_op_set_gemm_cfg(b"auto")
