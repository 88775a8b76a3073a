"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product (diarizen_amd/).

CPU fp32 restatement of the reference speaker-embedding forward
(PA/models/embedding/wespeaker/__init__.py:190-204: compute_fbank -> ResNet34 -> TSTP -> seg_1),
plain functional torch over a state_dict with the reference's key names ("resnet.conv1.weight"…).

Pinning:
  * ResNet34 trunk + TSTP/StatsPool + seg_1: PINNED against the reference's own
    wespeaker/resnet.py and blocks/pooling.py (loaded by file path in oracle/gen_golden.py ->
    tests/golden/emb_resnet.npz), and StatsPool against the reference's known-answer tests
    pyannote-audio/tests/test_stats_pool.py:28-131 (restated in tests/test_oracle.py).
  * kaldi fbank: the arithmetic lives in torchaudio.compliance.kaldi.fbank (third party, pinned
    torchaudio==2.1.1 in the reference README; NOT installed here, no fixture in the reference)
    => "parity unpinned" at that boundary.  The restatement follows Kaldi compute-fbank-feats
    semantics with torchaudio's defaults and is cross-checked against the independent
    transformers.audio_utils implementation (tests/test_oracle.py).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

EPS = torch.finfo(torch.float32).eps   # torchaudio _get_epsilon


# --------------------------------------------------------------------------- weights
from testkit.weights import emb_state_dict  # noqa: E402,F401


# --------------------------------------------------------------------------- kaldi fbank
def kaldi_mel_banks(num_bins: int = 80, n_fft: int = 512, sr: float = 16000.0, low: float = 20.0,
                    high: float = 0.0) -> torch.Tensor:
    """torchaudio.compliance.kaldi.get_mel_banks (vtln_warp 1.0) -> [num_bins, n_fft/2]."""
    nyq = 0.5 * sr
    if high <= 0.0:
        high += nyq
    bin_w = sr / n_fft
    mel_lo = 1127.0 * math.log(1.0 + low / 700.0)
    mel_hi = 1127.0 * math.log(1.0 + high / 700.0)
    delta = (mel_hi - mel_lo) / (num_bins + 1)
    b = torch.arange(num_bins).unsqueeze(1)
    left = mel_lo + b * delta
    center = mel_lo + (b + 1.0) * delta
    right = mel_lo + (b + 2.0) * delta
    mel = (1127.0 * (1.0 + bin_w * torch.arange(n_fft // 2) / 700.0).log()).unsqueeze(0)
    up = (mel - left) / (center - left)
    down = (right - mel) / (right - center)
    return torch.max(torch.zeros(1), torch.min(up, down))


def kaldi_fbank(wave: torch.Tensor, num_mel_bins: int = 80, frame_length: int = 400,
                frame_shift: int = 160, preemph: float = 0.97) -> torch.Tensor:
    """torchaudio.compliance.kaldi.fbank(num_mel_bins=80, frame_length=25, frame_shift=10,
    dither=0, sample_frequency=16000, window_type="hamming", use_energy=False) on a 1-D (already
    x 2^15) waveform: snip_edges framing, remove_dc_offset, pre-emphasis with replicate pad,
    Hamming, zero-pad to 512, |rfft|^2, mel, log(max(., eps)).  -> [T, 80]"""
    assert wave.dim() == 1 and wave.numel() >= frame_length
    frames = wave.unfold(0, frame_length, frame_shift)
    frames = frames - frames.mean(dim=1, keepdim=True)
    prev = torch.cat([frames[:, :1], frames[:, :-1]], dim=1)
    frames = frames - preemph * prev
    frames = frames * torch.hamming_window(frame_length, periodic=False, alpha=0.54, beta=0.46)
    n_fft = 512
    frames = F.pad(frames, (0, n_fft - frame_length))
    spec = torch.fft.rfft(frames).abs().pow(2.0)
    banks = F.pad(kaldi_mel_banks(num_mel_bins, n_fft), (0, 1))
    mel = spec @ banks.T
    return torch.max(mel, torch.tensor(EPS)).log()


def compute_fbank(waves: torch.Tensor) -> torch.Tensor:
    """wespeaker/__init__.py:80-103: x * (1<<15), per-item fbank (vmap), subtract the mean over
    frames.  waves [B, N] -> [B, T, 80]"""
    feats = torch.stack([kaldi_fbank(w * (1 << 15)) for w in waves])
    return feats - feats.mean(dim=1, keepdim=True)


# --------------------------------------------------------------------------- resnet + pooling
def _bn(sd, name, x):
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"],
                        sd[name + ".bias"], training=False, eps=1e-5)


def resnet_trunk(sd, fbank: torch.Tensor, num_blocks=(3, 4, 6, 3)) -> torch.Tensor:
    """resnet.py:358-365: [B, T, F] -> [B, 256, F/8, T/8]"""
    x = fbank.permute(0, 2, 1).unsqueeze(1)
    out = F.relu(_bn(sd, "resnet.bn1", F.conv2d(x, sd["resnet.conv1.weight"], padding=1)))
    for s, nb in enumerate(num_blocks):
        for j in range(nb):
            p = f"resnet.layer{s + 1}.{j}"
            stride = 2 if (j == 0 and s > 0) else 1
            y = F.relu(_bn(sd, p + ".bn1", F.conv2d(out, sd[p + ".conv1.weight"], stride=stride, padding=1)))
            y = _bn(sd, p + ".bn2", F.conv2d(y, sd[p + ".conv2.weight"], padding=1))
            if (p + ".shortcut.0.weight") in sd:
                scut = _bn(sd, p + ".shortcut.1", F.conv2d(out, sd[p + ".shortcut.0.weight"], stride=stride))
            else:
                scut = out
            out = F.relu(y + scut)                       # resnet.py:139-144
    return out


def stats_pool(seq: torch.Tensor, weights: Optional[torch.Tensor]) -> torch.Tensor:
    """PA/models/blocks/pooling.py:77-131.  seq [B, F, T]; weights None | [B, T'] | [B, S, T']."""
    if weights is None:
        return torch.cat([seq.mean(dim=-1), seq.std(dim=-1, correction=1)], dim=-1)
    squeeze = weights.dim() == 2
    if squeeze:
        weights = weights.unsqueeze(1)
    T = seq.shape[-1]
    if weights.shape[-1] != T:
        weights = F.interpolate(weights, size=T, mode="nearest")
    outs = []
    for s in range(weights.shape[1]):
        w = weights[:, s].unsqueeze(1)                      # _pool, pooling.py:63-75
        v1 = w.sum(dim=2) + 1e-8
        mean = torch.sum(seq * w, dim=2) / v1
        dx2 = torch.square(seq - mean.unsqueeze(2))
        v2 = torch.square(w).sum(dim=2)
        var = torch.sum(dx2 * w, dim=2) / (v1 - v2 / v1 + 1e-8)
        outs.append(torch.cat([mean, torch.sqrt(var)], dim=1))
    out = torch.stack(outs, dim=1)
    return out.squeeze(1) if squeeze else out


@torch.inference_mode()
def emb_forward(sd, waves: torch.Tensor, masks: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
    """waves [B, N], masks [B, S, L] (or [B, L]) -> [B, S, 256] (or [B, 256]).  The trunk is run
    once per window; pooling each of the S masks from it is bit-identical to S separate passes
    of the reference."""
    fb = compute_fbank(waves)
    if taps is not None:
        taps["fbank"] = fb
    out = resnet_trunk(sd, fb)
    B, Cc, Hh, Tt = out.shape
    seq = out.reshape(B, Cc * Hh, Tt)                       # rearrange "b d c f -> b (d c) f"
    stats = stats_pool(seq, masks)
    if taps is not None:
        taps["pool"] = stats
    return F.linear(stats, sd["resnet.seg_1.weight"], sd["resnet.seg_1.bias"])
