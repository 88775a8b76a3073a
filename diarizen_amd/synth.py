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
