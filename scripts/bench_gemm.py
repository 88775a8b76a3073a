"""Micro-benchmark of the MFMA contraction kernel on transformer-layer shapes (GPU box)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from diarizen_amd import ops

dev = torch.device("cuda:0")
shapes = [(12768, 1024, 1024), (12768, 1770, 1024), (12768, 1024, 1792), (12768, 192, 1024),
          (12768, 96, 1024), (409568, 160, 1536), (2042880, 32, 288), (12768, 64, 8192)]
for prec in (0, 1):
    for (M, N, K) in shapes:
        A = torch.randn(M, K, device=dev)
        W = torch.randn(N, K, device=dev) * 0.05
        W16 = W.bfloat16()
        out = torch.empty(M, N, device=dev)
        for _ in range(2):
            ops.gemm(A, W, W16=W16, C_out=out, precision=prec)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        it = 5
        for _ in range(it):
            ops.gemm(A, W, W16=W16, C_out=out, precision=prec)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / it
        print(f"prec={'f32' if prec==0 else 'bf16'} M={M} N={N} K={K}: {dt*1e3:.3f} ms  {2*M*N*K/dt/1e12:.1f} TF/s", flush=True)
