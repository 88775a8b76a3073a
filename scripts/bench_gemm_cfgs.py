"""Tile / pipeline-depth probe of csrc/gemm_split.hip in ONE process:
    python scripts/bench_gemm_cfgs.py cfg1,cfg2,... M,N,K [M,N,K ...]
times the fp16 two-term (precision 3, f32h) and single-term (precision 4, f16) kernels with pre-split weights
for every forced configuration (dzn_op_set_gemm_cfg) and checks each result against the "auto" one."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from diarizen_amd import _lib, ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
lib = _lib.load()
cfgs = sys.argv[1].split(",")
for spec in sys.argv[2:]:
    M, N, K = map(int, spec.split(","))
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) * 0.05
    R = torch.randn(M, N, device=dev)
    W3 = ops.split_weights(W)
    W2h, cs = ops.split_weights_h2(W)
    am = ops.amax(A)
    base = {}
    for cfg in ["auto"] + cfgs:
        lib.dzn_op_set_gemm_cfg(cfg.encode())
        for name, prec in (("f32h", 3), ("f16", 4)):
            out = torch.empty(M, N, device=dev)
            kw = {"W3": W3, "W2h": W2h, "col_scale": cs, "a_amax": am}
            try:
                for _ in range(2):
                    ops.gemm(A, W, C_out=out, R=R, precision=prec, **kw)
                torch.cuda.synchronize()
            except Exception as e:     # a configuration that cannot launch (LDS / registers)
                print(f"cfg={cfg:12s} {name} M={M} N={N} K={K}: FAILED {e}", flush=True)
                continue
            it, dt = 8, 1e9
            for rep in range(2):          # best of two batches: the first batch after a configuration switch runs colder
                st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                st.record()
                for _ in range(it):
                    ops.gemm(A, W, C_out=out, R=R, precision=prec, **kw)
                en.record(); torch.cuda.synchronize()
                dt = min(dt, st.elapsed_time(en) / it * 1e-3)
            if cfg == "auto":
                base[name] = out.clone()
                err = 0.0
            else:
                err = (out - base[name]).abs().max().item()
            print(f"cfg={cfg:12s} {name} M={M} N={N} K={K}: {dt*1e6:8.1f} us {2*M*N*K/dt/1e12:6.1f} TF/s  max|d vs auto|={err:.1e}",
                  flush=True)
lib.dzn_op_set_gemm_cfg(b"auto")
