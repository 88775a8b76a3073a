"""Synthetic inputs: there is no network for datasets, so benchmarks and the large
parity runs use this seeded meeting-like signal."""
from __future__ import annotations

import torch


def synth_recording(num_samples: int, seed: int = 3407) -> torch.Tensor:
    """Meeting-like synthetic audio: 4 band-limited noise 'speakers' with random 1-8 s turns
    (<= 2 concurrent) over a noise floor, clipped to [-1, 1].  Seed 3407 is the
    reference's own (diarizen/utils.py:137)."""
    g = torch.Generator().manual_seed(seed)
    sr = 16000
    x = 0.01 * torch.randn(num_samples, generator=g)
    t = torch.arange(num_samples) / sr
    active = torch.zeros(4, num_samples)
    pos = 0
    while pos < num_samples:
        dur = int((1.0 + 7.0 * torch.rand(1, generator=g).item()) * sr)
        k = int(torch.randint(0, 3, (1,), generator=g).item())          # 0, 1 or 2 speakers
        spk = torch.randperm(4, generator=g)[:k]
        for s in spk.tolist():
            active[s, pos:pos + dur] = 1.0
        pos += dur
    for s in range(4):
        f0 = 110.0 + 45.0 * s
        voice = (torch.sin(2 * torch.pi * f0 * t) + 0.5 * torch.sin(2 * torch.pi * 2.7 * f0 * t)
                 + 0.3 * torch.randn(num_samples, generator=g))
        x += 0.08 * active[s] * voice
    return x.clamp_(-1.0, 1.0)


BLOCK_S = 60   # seconds per independently seeded block of synth_recording_range


def synth_recording_range(start: int, num: int, total: int, seed: int = 3407) -> torch.Tensor:
    """samples [start, start + num) of a `total`-sample recording built from independently seeded 60 s blocks
    (block b = synth_recording(60 s, seed + 1000 b)), zero beyond `total`: any rank can materialise just the slice
    its windows touch (strong scaling over one long recording, BASELINE configs[3]) and all ranks agree on the data."""
    out = torch.zeros(max(num, 0))
    if num <= 0:
        return out
    bl = BLOCK_S * 16000
    end = min(start + num, total)
    for b in range(start // bl, (max(end, start + 1) - 1) // bl + 1):
        lo, hi = max(start, b * bl), min(end, (b + 1) * bl)
        if hi <= lo:
            continue
        blk = synth_recording(min(bl, total - b * bl), seed=seed + 1000 * b)
        out[lo - start:hi - start] = blk[lo - b * bl:hi - b * bl]
    return out
