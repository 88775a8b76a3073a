"""Kernel classes of ONE 32-window batch of BASELINE configs[1] (wavlm_base_s80_md, 5 s windows, segmentation only) in situ:
python scripts/probe_config1.py [B]"""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from diarizen_amd import _lib
from diarizen_amd.configs import get_seg_config
from diarizen_amd.engine import Engine
from testkit.synth import synth_recording
from testkit.weights import turn_taking_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
cfg = get_seg_config("wavlm_base_s80_md")
N = 80000
eng = Engine(cfg, turn_taking_state_dict(cfg, 0), None, None, max_batch=B, max_samples=N, precision="f32h", device=dev)
wave = synth_recording(N + 1600 * (B - 1), seed=1).to(dev)
views = torch.as_strided(wave, (B, N), (1600, 1)).contiguous()
for it in range(12):
    if it == 2:
        torch.cuda.synchronize()
        _lib.profile_enable(True)
    eng.segment(views, want_logp=False)
torch.cuda.synchronize()
prof = _lib.profile_collect()
_lib.profile_enable(False)
tot = sum(p["ms"] for p in prof)
nl = 0
for p in sorted(prof, key=lambda p: -p["ms"]):
    tf = p["flops"] / (p["ms"] * 1e-3) / 1e12 if p["flops"] > 0 else 0.0
    gb = p["bytes"] / (p["ms"] * 1e-3) / 1e9 if p["bytes"] > 0 else 0.0
    nl += p["launches"]
    print(f"{p['name']:28s} launches={p['launches'] // 10:4d} us/launch={p['ms'] / p['launches'] * 1e3:8.1f} share={p['ms'] / tot:.3f} TF/s={tf:7.1f} GB/s={gb:7.1f}")
print(f"sum of kernel times {tot / 10:.3f} ms per batch of {B}, {nl // 10} launches")
torch.cuda.synchronize()
t0 = time.perf_counter()
for it in range(20):
    eng.segment(views, want_logp=False)
torch.cuda.synchronize()
print(f"wall {1e3 * (time.perf_counter() - t0) / 20:.3f} ms per batch (one stream, unprofiled)")
