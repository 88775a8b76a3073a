"""Time the kernel classes of one batch of windows in situ (HIP-event profiler):  python scripts/probe_kernel_class.py [B] [class-substring ...]
Used with ablation environment switches (e.g. DZN_CONV01_ABL=1/2: the fused frontend without its VALU / MFMA phase)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from diarizen_amd import _lib
from diarizen_amd.configs import RESNET34, get_seg_config
from diarizen_amd.engine import Engine
from testkit.synth import synth_recording
from testkit.weights import emb_state_dict, seg_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 374
want = sys.argv[2:]
dev = torch.device("cuda:0")
cfg = get_seg_config("wavlm_large_s80_md")
eng = Engine(cfg, seg_state_dict(cfg, 0), RESNET34, emb_state_dict(0), max_batch=B, max_samples=128000, precision="f32h", device=dev)
wave = synth_recording(128000 + 12800 * (B - 1), seed=1).to(dev)
views = torch.as_strided(wave, (B, 128000), (12800, 1)).contiguous()
for it in range(3):
    if it == 1:
        _lib.profile_enable(True)
    logp, ml = eng.segment(views, want_logp=False)
    masks = ml.float().permute(0, 2, 1).contiguous()
    eng.embed(views, masks)
torch.cuda.synchronize()
prof = _lib.profile_collect()
_lib.profile_enable(False)
tot = sum(p["ms"] for p in prof)
for p in sorted(prof, key=lambda p: -p["ms"]):
    if want and not any(w in p["name"] for w in want):
        continue
    tf = p["flops"] / (p["ms"] * 1e-3) / 1e12 if p["flops"] > 0 else 0.0
    gb = p["bytes"] / (p["ms"] * 1e-3) / 1e9 if p["bytes"] > 0 else 0.0
    print(f"{p['name']:28s} launches={p['launches']:4d} ms/launch={p['ms'] / p['launches']:8.3f} share={p['ms'] / tot:.3f} TF/s={tf:7.1f} GB/s={gb:7.1f}")
print(f"total {tot / 2:.2f} ms per batch of {B}")
