"""ORACLE — TEST INFRASTRUCTURE ONLY.

Fits the classifier of the seeded "turn taking" weights (testkit/weights.py:turn_taking_state_dict)
so that the seeded model's hard decisions are NOT degenerate.  VERDICT r1 weak #1: with plain random
weights every frame of every fixture is one powerset class, so "bit-exact decisions" compared
constants and the overlap-exclusion rule only ever took its fallback branch.

Method (deterministic, no labels needed): run the oracle (oracle/seg_model.py) up to the classifier
input on calibration audio, take the within-window principal components of those head features
(within-window so that per-window offsets do not pick one class per window), and let the 11 logits
be seeded random combinations of the top-k whitened components: every class wins somewhere, the
argmax follows the (smooth) component dynamics.  The random draw is repeated until the fixture
criteria hold on the calibration audio: >= 6 classes with >= 5 % of the frames each, >= 5 class
transitions in every window, smallest top-2 logit margin >= 3e-4 (so fp32 re-association noise of
~1e-5 cannot flip a decision of the fixture itself).

    python oracle/calibrate.py                 # all configs -> testkit/data/cal_<config>_seed0.npz
    python oracle/calibrate.py --outlier       # the planted-massive-activation weights (testkit/weights.py:outlier_state_dict)
                                               # -> cal_<config>_outlier_seed0.npz: classifier + the LayerNorm sigma table

Calibration audio: tests/golden/EN2002a_30s.wav (the reference's example file) plus a few windows of
the synthetic bench recording, cut into windows of the config's fixture length.
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from oracle.wav import first_channel_pcm16 as first_channel_16k  # noqa: E402
from oracle.configs import get_seg_config  # noqa: E402
from testkit.synth import synth_recording  # noqa: E402
from testkit.weights import (CAL_DIR, outlier_ln_sites, outlier_plan, plant_outliers, seg_state_dict,  # noqa: E402
                             turn_taking_head)
from oracle import seg_model  # noqa: E402
from oracle.pipeline import slide_windows  # noqa: E402

WAV = ROOT / "tests" / "golden" / "EN2002a_30s.wav"
# config -> (window samples, step samples over the wav, number of synthetic windows)
PLAN = {
    "wavlm_large_s80_md": (128000, 12800, 8),
    "wavlm_base_s80_md": (80000, 16000, 8),
    "tiny_ln": (8000, 8000, 8),
    "tiny_gn": (8000, 8000, 8),
}


def calibration_windows(name: str):
    """-> (windows [C, W], number of leading windows that are real audio)"""
    window, step, nsyn = PLAN[name]
    wave = torch.from_numpy(first_channel_16k(str(WAV)))
    real = slide_windows(wave, window, step)
    syn = synth_recording(window + (nsyn - 1) * (window // 2), seed=3407)
    syn = syn.unfold(0, window, window // 2)
    return torch.cat([real, syn], dim=0), real.shape[0]


@torch.inference_mode()
def head_features(sd, cfg, windows: torch.Tensor, batch: int = 4) -> np.ndarray:
    out = []
    for c0 in range(0, windows.shape[0], batch):
        taps = {}
        seg_model.seg_forward(sd, cfg, windows[c0:c0 + batch], taps)
        out.append(taps[f"conf{cfg.conf_layers - 1}"].numpy())
    return np.concatenate(out).astype(np.float64)


def fit_classifier(F: np.ndarray, n_classes: int, n_real: int, k: int = 8, gain: float = 2.0, max_tries: int = 200):
    """F [C, L, A] head features -> (W [n_classes, A], b [n_classes], report)."""
    C, L, A = F.shape
    X = F.reshape(-1, A)
    mu = X.mean(0)
    Xw = (F - F.mean(1, keepdims=True)).reshape(-1, A)
    _, s, vt = np.linalg.svd(Xw, full_matrices=False)
    k = min(k, A, len(s))
    std = s[:k] / np.sqrt(len(X))
    best = None
    for trial in range(max_tries):
        r = np.random.default_rng(1000 + trial)
        G = r.normal(size=(n_classes, k)) * gain
        W = (G / std) @ vt[:k]
        b = -W @ mu + r.normal(size=n_classes) * 0.3
        logits = X @ W.T + b
        am = logits.argmax(1).reshape(C, L)
        hist = np.bincount(am.ravel(), minlength=n_classes) / am.size
        trans = (am[:n_real, 1:] != am[:n_real, :-1]).sum(1)      # transitions are asked of the real audio
        need_trans = min(5, max(1, L // 40))
        srt = np.sort(logits, 1)
        margin = float((srt[:, -1] - srt[:, -2]).min())
        rep = dict(trial=trial, classes_ge_5pct=int((hist >= 0.05).sum()), hist=hist.round(4).tolist(),
                   min_transitions=int(trans.min()), mean_transitions=float(trans.mean()), min_margin=margin,
                   w_row_norm=float(np.linalg.norm(W, axis=1).max()), k=int(k), gain=float(gain))
        score = (rep["classes_ge_5pct"] >= 6) + (rep["min_transitions"] >= need_trans) + (margin >= 3e-4)
        if best is None or score > best[0]:
            best = (score, W, b, rep)
        if score == 3:
            break
    return best[1], best[2], best[3]


def fit_ln_sigma(sd_base, cfg, windows: torch.Tensor, seed: int = 0) -> np.ndarray:
    """One oracle pass over `windows` with the planted weights in which every LayerNorm that reads the outlier-carrying stream
    measures the std of the TYPICAL channels of its input and rescales its gamma for them on the spot (sites are visited in
    forward order, so each measurement already sees the calibrated sites before it).  -> sigma per site of outlier_ln_sites(cfg),
    rounded to 4 significant digits so that the table, not a machine-dependent reduction order, defines the weights."""
    import torch.nn.functional as F
    chans, mags, rms, sigma = outlier_plan(cfg, seed)
    sd = plant_outliers(sd_base, cfg, seed, {})
    sites = outlier_ln_sites(cfg)
    by_ptr = {sd[k + ".weight"].data_ptr(): k for k in sites}
    typ = torch.ones(cfg.embed_dim, dtype=torch.bool)
    typ[chans] = False
    table = {}
    orig = F.layer_norm

    def measuring(x, shape, weight=None, bias=None, eps=1e-5):
        k = by_ptr.get(weight.data_ptr()) if weight is not None else None
        if k is not None and k not in table:
            s_ = float(f"{float(x[..., typ].std()):.4g}")
            table[k] = s_
            weight[typ] *= sigma / s_
        return orig(x, shape, weight, bias, eps)

    F.layer_norm = measuring
    try:
        with torch.inference_mode(False), torch.no_grad():
            seg_model.encoder(sd, cfg, seg_model.feature_extractor(sd, cfg, windows))
    finally:
        F.layer_norm = orig
    return np.array([table[k] for k in sites], dtype=np.float64)


def calibrate_outlier(name: str, seed: int = 0, verbose: bool = True):
    cfg = get_seg_config(name)
    base = turn_taking_head(seg_state_dict(cfg, seed), cfg, seed)
    windows, n_real = calibration_windows(name)
    ln_sigma = fit_ln_sigma(base, cfg, windows[:4], seed)
    sd = plant_outliers(base, cfg, seed, dict(zip(outlier_ln_sites(cfg), ln_sigma.tolist())))
    F = head_features(sd, cfg, windows)
    W, b, rep = fit_classifier(F, cfg.n_classes, n_real)
    path = CAL_DIR / f"cal_{name}_outlier_seed{seed}.npz"
    np.savez(path, W=W.astype(np.float32), b=b.astype(np.float32), ln_sigma=ln_sigma, report=np.array(repr(rep)))
    if verbose:
        print(name, "ln_sigma", ln_sigma.round(3).tolist())
        print(name, rep, "->", path.relative_to(ROOT))
    return path


def calibrate(name: str, seed: int = 0, verbose: bool = True):
    cfg = get_seg_config(name)
    sd = turn_taking_head(seg_state_dict(cfg, seed), cfg, seed)
    windows, n_real = calibration_windows(name)
    F = head_features(sd, cfg, windows)
    W, b, rep = fit_classifier(F, cfg.n_classes, n_real)
    CAL_DIR.mkdir(exist_ok=True)
    path = CAL_DIR / f"cal_{name}_seed{seed}.npz"
    np.savez(path, W=W.astype(np.float32), b=b.astype(np.float32), report=np.array(repr(rep)))
    if verbose:
        print(name, rep, "->", path.relative_to(ROOT))
    return path


if __name__ == "__main__":
    torch.set_num_threads(8)
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if "--outlier" in sys.argv:
        for nm in (args or ["wavlm_large_s80_md", "wavlm_base_s80_md", "tiny_ln", "tiny_gn"]):
            calibrate_outlier(nm)
    else:
        for nm in (args or list(PLAN)):
            calibrate(nm)
