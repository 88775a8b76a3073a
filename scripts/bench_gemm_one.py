"""One GEMM shape, repeated: target for rocprofv3 --pmc runs.  usage: bench_gemm_one.py M N K prec iters"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from diarizen_amd import ops
M, N, K, prec, it = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
dev = torch.device("cuda:0")
A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05; W16 = W.bfloat16()
out = torch.empty(M, N, device=dev)
for _ in range(it):
    ops.gemm(A, W, W16=W16, C_out=out, precision=prec)
torch.cuda.synchronize()
