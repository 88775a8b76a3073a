"""Split-GEMM tile probe: python scripts/bench_gemm_h2.py M,N,K [...]   (DZN_GEMM_CFG=128x192 etc. per process)
times the bf16 three-term (precision 2) and fp16 two-term (precision 3) kernels with operands split beforehand."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from diarizen_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg = os.environ.get("DZN_GEMM_CFG", "auto")
for spec in sys.argv[1:]:
    M, N, K = map(int, spec.split(","))
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) * 0.05
    R = torch.randn(M, N, device=dev)
    W3 = ops.split_weights(W)
    W2h, cs = ops.split_weights_h2(W)
    am = ops.amax(A)
    out = torch.empty(M, N, device=dev)
    for name, prec, kw in (("f32s", 2, {"W3": W3}), ("f32h", 3, {"W3": W3, "W2h": W2h, "col_scale": cs, "a_amax": am})):
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
        print(f"cfg={cfg:8s} {name} M={M} N={N} K={K}: {dt*1e6:8.1f} us {2*M*N*K/dt/1e12:6.1f} TF/s", flush=True)
