"""fp32 MFMA kernel vs 3-way bf16 split kernel: accuracy against float64 and throughput.
python scripts/bench_gemm_split.py M,N,K [M,N,K ...]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from diarizen_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for spec in sys.argv[1:]:
    M, N, K = map(int, spec.split(","))
    A = torch.randn(M, K, device=dev) * torch.exp(torch.randn(M, K, device=dev))
    W = torch.randn(N, K, device=dev) * 0.05
    import os
    if os.environ.get("ZERO"):
        A.zero_(); W.zero_()
    W3 = ops.split_weights(W)
    ref = (A[:2048].double() @ W.double().T)
    scale = (A[:2048].double().abs() @ W.double().abs().T)
    res = {}
    A3 = ops.split_rows(A)
    for name, prec, kw in (("f32", 0, {}), ("f32s", 2, {"W3": W3}), ("f32sp", 2, {"W3": W3, "a_planes": A3})):
        out = torch.empty(M, N, device=dev)
        for _ in range(3):
            ops.gemm(A, W, C_out=out, precision=prec, **kw)
        torch.cuda.synchronize()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 10
        st.record()
        for _ in range(it):
            ops.gemm(A, W, C_out=out, precision=prec, **kw)
        en.record(); torch.cuda.synchronize()
        dt = st.elapsed_time(en) / it * 1e-3
        err = ((out[:2048].double() - ref).abs() / scale).max().item()
        rms = ((out[:2048].double() - ref) / scale).pow(2).mean().sqrt().item()
        res[name] = out
        print(f"{name:5s} M={M} N={N} K={K}: {dt*1e6:8.1f} us {2*M*N*K/dt/1e12:6.1f} TF/s  max|err|/sum|a||w| = {err:.3e} rms {rms:.3e}", flush=True)
    print("   max|f32s - f32| =", (res["f32"] - res["f32s"]).abs().max().item())
