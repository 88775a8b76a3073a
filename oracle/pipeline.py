"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product (diarizen_amd/).

CPU restatement of the device half of `DiariZenPipeline.__call__` exactly as the reference
executes it (diarizen/pipelines/inference.py:127-149):
  * Inference.slide windowing incl. the zero-padded last window (PA/core/inference.py:265-343),
  * hard powerset decisions per window (PA/core/inference.py:226),
  * scipy median filter (inference.py:131-132),
  * get_embeddings: overlap-excluded masks with the min_num_frames fallback and ONE embedding
    forward per (window, local speaker) (PA/pipelines/speaker_diarization.py:268-360).
The models are the oracle restatements (oracle/seg_model.py, oracle/emb_model.py), which are
pinned to the reference's modules.  Used to generate tests/golden/e2e_*.npz.
"""
from __future__ import annotations

import math

import numpy as np
import torch
from scipy.ndimage import median_filter

from oracle import emb_model, seg_model


def slide_windows(wave: torch.Tensor, window: int, step: int) -> torch.Tensor:
    """[N] -> [C, window] (PA/core/inference.py:282-299)."""
    n = wave.numel()
    chunks = []
    if n >= window:
        chunks.append(wave.unfold(0, window, step))
        num = chunks[0].shape[0]
        has_last = (n - window) % step > 0
    else:
        num, has_last = 0, True
    if has_last:
        last = wave[num * step:]
        chunks.append(torch.nn.functional.pad(last, (0, window - last.numel()))[None])
    return torch.cat(chunks, dim=0)


@torch.inference_mode()
def device_stage_reference(wave: torch.Tensor, cfg, sd, esd, duration: float, step_ratio: float = 0.1,
                           batch: int = 4, median: bool = True, sr: int = 16000, verbose: bool = False):
    window = int(math.floor(duration * sr))
    step = int(round(step_ratio * duration * sr))
    chunks = slide_windows(wave, window, step)
    C = chunks.shape[0]
    segs = []
    for c0 in range(0, C, batch):
        logp = seg_model.seg_forward(sd, cfg, chunks[c0:c0 + batch])
        segs.append(seg_model.to_multilabel(logp, cfg).numpy())
        if verbose:
            print(f"  seg {min(c0 + batch, C)}/{C}", flush=True)
    seg = np.vstack(segs).astype(np.float32)
    if median:
        seg = median_filter(seg, size=(1, 11, 1), mode="reflect")
    L = seg.shape[1]
    min_num_frames = math.ceil(L * 400 / window)                    # speaker_diarization.py:274-278
    clean = seg * (np.sum(seg, axis=2, keepdims=True) < 2)
    embs = np.zeros((C, seg.shape[2], 256), dtype=np.float32)
    for c in range(C):
        for s in range(seg.shape[2]):
            mask = clean[c, :, s] if np.sum(clean[c, :, s]) > min_num_frames else seg[c, :, s]
            e = emb_model.emb_forward(esd, chunks[c:c + 1], torch.from_numpy(mask)[None])
            embs[c, s] = e[0].numpy()
        if verbose:
            print(f"  emb {c + 1}/{C}", flush=True)
    return seg.astype(np.uint8), embs
