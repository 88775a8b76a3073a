"""Exactness of hard decisions (SURVEY.md §7 "Exactness of hard decisions"): the pipeline
consumes argmax'd powerset classes, so log-prob tolerance is not enough — report and bound the argmax-flip
rate of every engine mode against the oracle on MANY non-degenerate windows.

Input: tests/golden/EN2002a_30s.wav cut into 8 s windows at a fine hop (the seeded turn-taking weights were
calibrated on this file: 11 classes, ~15 transitions per window).  DZN_DECISION_WINDOWS (default 96; the
committed report profiles/r2_decision_parity.json was taken with 256) windows go through the oracle on the
host CPU and through the HIP engine in f32s / f32h / f32 / bf16.

Bars: fp32 modes (f32 = fp32 MFMA, f32s = bf16 three-term split, f32h = fp16 two-term split) — max |dlogp| <= 1e-3,
argmax agreement >= 99.9 % of frames, and EVERY flipped frame must be a near tie of the oracle itself (top-2
margin <= 2e-3, i.e. inside the stated log-prob tolerance).
bf16 (reduced precision, BASELINE configs[4] territory) is REPORTED, with a loose bound only: the seeded weights
keep 91 % of the head-feature energy in a constant component (calibrated classifier rows have norm ~20 on a
9 % time-varying part), so bf16 operand rounding (2^-9) is of the size of the signal itself — a stress case that
says nothing about trained weights, whose acceptance bar is the DER delta (scripts/der.py).
The report (flip rate, min top-2 margin, margin of the flipped frames) is written to gpurun_out/.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.dirname(os.path.dirname(__file__))


def decision_parity(gpu, n_windows: int, batch: int = 32):
    from diarizen_amd.audio import first_channel_16k
    from diarizen_amd.configs import get_seg_config
    from diarizen_amd.engine import Engine
    from testkit.weights import turn_taking_state_dict
    from oracle import seg_model
    cfg = get_seg_config("wavlm_large_s80_md")
    sd = turn_taking_state_dict(cfg, 0)
    wave = torch.from_numpy(first_channel_16k(os.path.join(GOLD, "EN2002a_30s.wav")))
    N = 128000
    hop = (wave.numel() - N) // max(n_windows - 1, 1)
    windows = torch.as_strided(wave, (n_windows, N), (hop, 1)).contiguous()
    ref = torch.cat([seg_model.seg_forward(sd, cfg, windows[b:b + 8]) for b in range(0, n_windows, 8)])
    top2 = ref.topk(2, dim=-1).values
    margin = top2[..., 0] - top2[..., 1]
    am_ref = ref.argmax(-1)
    hist = torch.bincount(am_ref.flatten(), minlength=cfg.n_classes)
    report = {"windows": n_windows, "frames": int(am_ref.numel()), "hop_samples": int(hop),
              "oracle": {"class_hist": hist.tolist(),
                         "transitions_per_window": float((am_ref[:, 1:] != am_ref[:, :-1]).sum(1).float().mean()),
                         "min_top2_margin": float(margin.min()),
                         "frames_with_margin_below_1e-3": int((margin < 1e-3).sum())},
              "modes": {}}
    from diarizen_amd import _lib
    has_bf16 = b"tuning build" in _lib.load().dzn_version()      # quarantined mode: DZN_TUNING=1 builds only
    for precision in ("f32s", "f32h", "f32", "f16") + (("bf16",) if has_bf16 else ()):
        eng = Engine(cfg, sd, max_batch=batch, max_samples=N, precision=precision, device=gpu)
        outs = []
        for b in range(0, n_windows, batch):
            lp, _ = eng.segment(windows[b:b + batch].to(gpu))
            outs.append(lp.cpu())
        logp = torch.cat(outs)
        flips = logp.argmax(-1) != am_ref
        report["modes"][precision] = {
            "max_abs_dlogp": float((logp - ref).abs().max()),
            "argmax_flips": int(flips.sum()), "flip_rate": float(flips.float().mean()),
            "max_oracle_margin_of_flipped_frames": float(margin[flips].max()) if flips.any() else 0.0}
        eng.close()
    return report


def test_argmax_flip_rate_vs_oracle(built_lib, gpu):
    n = int(os.environ.get("DZN_DECISION_WINDOWS", "96"))
    rep = decision_parity(gpu, n)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "decision_parity.json"), "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps(rep))
    assert sum(1 for c in rep["oracle"]["class_hist"] if c >= 0.05 * rep["frames"]) >= 6
    for p in ("f32s", "f32h", "f32"):
        m = rep["modes"][p]
        assert m["max_abs_dlogp"] <= 1e-3, (p, m)
        assert m["flip_rate"] <= 1e-3, (p, m)
        assert m["max_oracle_margin_of_flipped_frames"] <= 2e-3, (p, m)
    if "bf16" in rep["modes"]:
        m = rep["modes"]["bf16"]
        assert m["flip_rate"] <= 0.2 and np.isfinite(m["max_abs_dlogp"]), m
    m = rep["modes"]["f16"]                 # single-term fp16 (2^-11 per operand): reported, loose bound like bf16
    assert m["flip_rate"] <= 0.05 and np.isfinite(m["max_abs_dlogp"]), m
