"""Which contractions of the DZN_PREC_F16 engine need their second fp16 term?  

    python scripts/f16_sensitivity.py [windows=64] [out.json]

Runs W non-degenerate 8 s windows of tests/golden/EN2002a_30s.wav (seeded turn-taking weights) through the f32h engine
(the fp32-grade reference of this probe: it sits 3.6e-4 from the CPU oracle, profiles/r3_decision_parity.json) and
through f16 engines created under different DZN_F16_KEEP2 masks (bit i = contraction class i of
csrc/engine.cpp:F16_CLASSES keeps two terms; bit 14 = ResNet stages 2-4) with and without the centred LayerNorm-folded
split (DZN_F16_CENTER).  Reports max |dlogp|, the argmax flip rate and the time per pass against SURVEY 8d's reduced bar
(max |dlogp| <= 5e-2, argmax >= 99.5 %), and for the embedding side the worst cosine against the f32h embeddings."""
from __future__ import annotations

import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

CLASSES = ["conv gemm", "feature projection", "pos conv", "qkv", "out_proj", "ffn1", "ffn2", "proj", "conf ffn w1",
           "conf ffn w2", "conf qkv", "conf out", "conf pw1", "conf pw2", "resnet stages 2-4"]


def main():
    from diarizen_amd.audio import first_channel_16k
    from diarizen_amd.configs import RESNET34, get_seg_config
    from diarizen_amd.engine import Engine
    from testkit.weights import emb_state_dict, turn_taking_state_dict
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    out_path = sys.argv[2] if len(sys.argv) > 2 else None
    dev = torch.device("cuda:0")
    cfg = get_seg_config("wavlm_large_s80_md")
    sd, esd = turn_taking_state_dict(cfg, 0), emb_state_dict(0)
    wave = torch.from_numpy(first_channel_16k(str(ROOT / "tests" / "golden" / "EN2002a_30s.wav")))
    N = 128000
    hop = (wave.numel() - N) // max(W - 1, 1)
    windows = torch.as_strided(wave, (W, N), (hop, 1)).contiguous().to(dev)

    def run(precision, env):
        for k in ("DZN_F16_KEEP2", "DZN_F16_CENTER"):
            os.environ.pop(k, None)
        os.environ.update(env)
        eng = Engine(cfg, sd, RESNET34, esd, max_batch=W, max_samples=N, precision=precision, device=dev)
        logp, ml = eng.segment(windows)
        _, masks = eng.prepare_masks(ml, 11, True, 2)
        emb = eng.embed(windows, masks)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            lp2, ml2 = eng.segment(windows)
            eng.embed(windows, masks)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 2
        eng.close()
        return logp, masks, emb, dt

    ref, ref_masks, ref_emb, ref_dt = run("f32h", {})
    am_ref = ref.argmax(-1)
    top2 = ref.topk(2, dim=-1).values
    rep = {"windows": W, "frames": int(am_ref.numel()), "f32h_ms": round(ref_dt * 1e3, 1),
           "min_top2_margin_ref": float((top2[..., 0] - top2[..., 1]).min()), "rows": []}

    def probe(label, mask, center):
        logp, _, _, dt = run("f16", {"DZN_F16_KEEP2": hex(mask), "DZN_F16_CENTER": str(int(center))})
        # embeddings on the REFERENCE masks, so the embedding error is not mixed with decision flips
        os.environ.update({"DZN_F16_KEEP2": hex(mask), "DZN_F16_CENTER": str(int(center))})
        eng = Engine(cfg, sd, RESNET34, esd, max_batch=W, max_samples=N, precision="f16", device=dev)
        emb = eng.embed(windows, ref_masks)
        torch.cuda.synchronize()
        eng.close()
        act = ref_masks.sum(-1) > 0
        cos = torch.nn.functional.cosine_similarity(emb[act], ref_emb[act], dim=-1)
        flips = (logp.argmax(-1) != am_ref).float().mean().item()
        row = {"label": label, "mask": hex(mask), "center": int(center), "max_abs_dlogp": float((logp - ref).abs().max()),
               "flip_rate": flips, "min_cos_emb": float(cos.min()), "ms": round(dt * 1e3, 1)}
        rep["rows"].append(row)
        print(json.dumps(row), flush=True)

    probe("r3 f16 (one term everywhere, folded LN)", 0, False)
    probe("one term everywhere, centred LN split", 0, True)
    for i, name in enumerate(CLASSES):
        probe(f"two terms: {name}", 1 << i, True)
    probe("two terms: qkv + ffn1 (folded LN, K = 1024), not centred", (1 << 3) | (1 << 5), False)
    probe("two terms: all conformer classes", 0x3F00, True)
    probe("two terms: all conformer + proj + feature projection", 0x3F82, True)
    probe("two terms: everything but ffn1/ffn2/qkv/out_proj/resnet", 0x3F87, True)
    probe("two terms everywhere (= f32h arithmetic)", 0x7FFF, True)
    if out_path:
        Path(out_path).write_text(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
