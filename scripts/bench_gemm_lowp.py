"""bf16 x bf16 GEMM shape probe: python scripts/bench_gemm_lowp.py M,N,K ..."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from diarizen_amd import ops
dev = torch.device("cuda:0")
for spec in sys.argv[1:]:
    M, N, K = map(int, spec.split(","))
    A = torch.randn(M, K, device=dev).bfloat16(); W16 = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    for odt in (torch.float32, torch.bfloat16):
        out = torch.empty(M, N, device=dev, dtype=odt)
        for _ in range(3):
            ops.gemm(A, None, N=N, K=K, ldw=K, W16=W16, C_out=out, precision=1)
        torch.cuda.synchronize()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(10):
            ops.gemm(A, None, N=N, K=K, ldw=K, W16=W16, C_out=out, precision=1)
        en.record(); torch.cuda.synchronize()
        dt = st.elapsed_time(en) / 10 * 1e-3
        print(f"bf16xbf16->{'f32' if odt==torch.float32 else 'bf16'} M={M} N={N} K={K}: {dt*1e6:.1f} us {2*M*N*K/dt/1e12:.1f} TF/s", flush=True)
