"""GEMM shape probe: python scripts/bench_gemm_shapes.py prec M,N,K [M,N,K ...]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from diarizen_amd import ops
dev = torch.device("cuda:0")
prec = int(sys.argv[1])
for spec in sys.argv[2:]:
    M, N, K = map(int, spec.split(","))
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05; W16 = W.bfloat16()
    out = torch.empty(M, N, device=dev)
    for _ in range(3):
        ops.gemm(A, W, W16=W16, C_out=out, precision=prec)
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 10
    st.record()
    for _ in range(it):
        ops.gemm(A, W, W16=W16, C_out=out, precision=prec)
    en.record(); torch.cuda.synchronize()
    dt = st.elapsed_time(en) / it * 1e-3
    print(f"prec={prec} M={M} N={N} K={K}: {dt*1e6:.1f} us  {2*M*N*K/dt/1e12:.1f} TF/s", flush=True)
