"""Both encoder-attention kernels on the bench's launch shape (the in-kernel split of attention_split.hip and the pre-split planes of
attention_planes.hip), a few launches each — run under rocprofv3 --pmc to see what the tile loop waits for:
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY ... --kernel-trace --output-format csv -d out -- python scripts/bench_attention_pmc.py
    python scripts/bench_attention_pmc.py --summarise out        # per-kernel sums of the counter CSVs under out/"""
import csv
import glob
import sys
from collections import defaultdict
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
    acc = defaultdict(lambda: defaultdict(float))
    n = defaultdict(int)
    for f in glob.glob(sys.argv[2] + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            full = r["Kernel_Name"]
            if "attn" not in full:
                continue
            k = full[full.index("attn"):][:48]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
    for k in acc:
        print(k)
        for c, v in sorted(acc[k].items()):
            print(f"   {c:32s} {v / max(n[(k, c)], 1):16.1f} per launch ({n[(k, c)]} launches)")
    sys.exit(0)

import torch
from diarizen_amd import _lib, ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, L, h, Htot = 256, 399, 5, 16
qkv = torch.randn(B * L, 3 * h * 64, device=dev)
gate = torch.rand(B * L, Htot, device=dev) * 2
table = torch.randn(Htot, 2 * L - 1, device=dev)
heads = torch.tensor([0, 3, 7, 12, 15], dtype=torch.int32, device=dev)
kw = dict(gate=gate, table=table, head_idx=heads, Htot=Htot)
for name, fn in (("split", lambda: ops.attention(qkv, B, L, h, precision=_lib.DZN_PREC_F32_H2, **kw)),
                 ("planes", lambda: ops.attention_planes(qkv, B, L, h, **kw))):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(5):
        fn()
    en.record()
    torch.cuda.synchronize()
    dt = st.elapsed_time(en) / 5 * 1e-3
    print(f"{name}: {dt * 1e3:.3f} ms per call (incl. the pack kernel for planes) {4.0 * B * h * L * L * 64 / dt / 1e12:.1f} TFLOP/s", flush=True)
