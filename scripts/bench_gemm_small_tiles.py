"""Short-K contraction class (out_proj / FFN-out: N = 1024, K = 128 .. 480, residual epilogue) under a forced tile:
    DZN_GEMM_CFG=<cfg> python scripts/bench_gemm_small_tiles.py [ref.pt]
prints time / algorithmic TFLOP/s / algorithmic TB/s per shape; with a path, saves (or compares against) the outputs of the first run
— every tile shape walks K in the same order, so the results must be bit-identical."""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from diarizen_amd import ops
dev = torch.device("cuda:0")
cfg = os.environ.get("DZN_GEMM_CFG", "auto")
ref_path = sys.argv[1] if len(sys.argv) > 1 else None
ref = torch.load(ref_path) if ref_path and os.path.exists(ref_path) else {}
new = {}
M, N = 223839, 1024
for K in (128, 256, 384, 480):
    torch.manual_seed(K)
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) * 0.05
    R = torch.randn(M, N, device=dev)
    W2h, cs = ops.split_weights_h2(W)
    am = ops.amax(A)
    out = torch.empty(M, N, device=dev)
    kw = {"W2h": W2h, "col_scale": cs, "a_amax": am}
    for _ in range(3):
        ops.gemm(A, W, C_out=out, R=R, precision=3, **kw)
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 20
    st.record()
    for _ in range(it):
        ops.gemm(A, W, C_out=out, R=R, precision=3, **kw)
    en.record()
    torch.cuda.synchronize()
    dt = st.elapsed_time(en) / it * 1e-3
    alg_bytes = 4.0 * (M * K + 2 * M * N) + 4.0 * N * K
    same = ""
    key = f"K{K}"
    if key in ref:
        same = " bit-identical to the first run: " + str(bool(torch.equal(ref[key], out[:4096].cpu())))
    new[key] = out[:4096].cpu()
    print(f"[{cfg:10s}] M={M} N={N} K={K}: {dt * 1e6:7.1f} us  {2 * M * N * K / dt / 1e12:6.1f} TFLOP/s  {alg_bytes / dt / 1e12:5.2f} TB/s algorithmic{same}", flush=True)
if ref_path and not ref:
    torch.save(new, ref_path)
