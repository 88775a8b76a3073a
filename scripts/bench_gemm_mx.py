"""Contraction probe of the reduced mode: python scripts/bench_gemm_mx.py M,N,K [...]
times, on the same operands (planes packed beforehand, residual epilogue as in the encoder), the fp16 two-term kernel (f32h,
3 products), the single-term fp16 kernel (r2-r4's f16) and the MX kernel (fp16 hi*hi + fp8 cross terms, csrc/gemm_mx.hip) in its
shape-rule tile and in both forced tiles."""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from diarizen_amd import _lib, ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
lib = _lib.load()
for spec in sys.argv[1:]:
    M, N, K = map(int, spec.split(","))
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) * 0.05
    R = torch.randn(M, N, device=dev)
    W3 = ops.split_weights(W)
    W2h, cs = ops.split_weights_h2(W)
    Wmx, csm = ops.split_weights_mx(W)
    am = ops.amax(A)
    out = torch.empty(M, N, device=dev)
    base = {"W3": W3, "W2h": W2h, "col_scale": cs, "a_amax": am}
    runs = [("f32h", 3, base, None), ("f16x1", 4, base, None),
            ("mx auto", 4, dict(base, Wmx=Wmx, col_scale_mx=csm), "auto"),
            ("mx 128x128", 4, dict(base, Wmx=Wmx, col_scale_mx=csm), "128x128"),
            ("mx 128x64", 4, dict(base, Wmx=Wmx, col_scale_mx=csm), "128x64"),
            ("mx 128x64rpf", 4, dict(base, Wmx=Wmx, col_scale_mx=csm), "128x64rpf"),
            ("mx 256x128", 4, dict(base, Wmx=Wmx, col_scale_mx=csm), "256x128")]
    for name, prec, kw, force in runs:
        if force is not None:
            lib.dzn_op_set_gemm_mx_cfg(force.encode())
        for _ in range(3):
            ops.gemm(A, W, C_out=out, R=R, precision=prec, **kw)
        torch.cuda.synchronize()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 10
        st.record()
        for _ in range(it):
            ops.gemm(A, W, C_out=out, R=R, precision=prec, **kw)
        en.record(); torch.cuda.synchronize()
        dt = st.elapsed_time(en) / it * 1e-3
        print(f"[DZN_GEMM_RPF={os.environ.get('DZN_GEMM_RPF', 'default')}] {name:12s} M={M} N={N} K={K}: {dt*1e6:8.1f} us {2*M*N*K/dt/1e12:6.1f} TFLOP/s algorithmic", flush=True)
    lib.dzn_op_set_gemm_mx_cfg(b"auto")
