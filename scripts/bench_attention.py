import sys, torch
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from diarizen_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, L, h, Htot = 256, 399, 12, 16
qkv = torch.randn(B * L, 3 * h * 64, device=dev)
gate = torch.rand(B * L, Htot, device=dev) * 2
table = torch.randn(Htot, 2 * L - 1, device=dev)
heads = torch.tensor([0, 1, 2, 4, 5, 7, 8, 9, 11, 12, 13, 15], dtype=torch.int32, device=dev)
for prec in (0, 2):
    for bias in (True, False):
        kw = dict(gate=gate, table=table, head_idx=heads, Htot=Htot) if bias else {}
        for _ in range(2): out = ops.attention(qkv, B, L, h, precision=prec, **kw)
        torch.cuda.synchronize()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(5): out = ops.attention(qkv, B, L, h, precision=prec, **kw)
        en.record(); torch.cuda.synchronize()
        dt = st.elapsed_time(en) / 5 * 1e-3
        print(f"prec={prec} bias={bias}: {dt*1e3:.3f} ms {4.0*B*h*L*L*64/dt/1e12:.1f} TF/s", flush=True)
