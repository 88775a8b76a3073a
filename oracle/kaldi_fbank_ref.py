"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product (diarizen_amd/).

A SECOND, independently structured restatement of the filter-bank front end of the WeSpeaker embedding model
(PA/models/embedding/wespeaker/__init__.py:80-103 -> torchaudio.compliance.kaldi.fbank, third party, pinned
torchaudio==2.1.1 by the reference's README, absent offline => the row stays "parity unpinned").

oracle/emb_model.py:kaldi_fbank follows torchaudio's vectorised call graph (unfold -> mean -> shifted copy -> rfft ->
dense [80, 257] mel matrix).  This file follows what torchaudio itself restates — Kaldi's C++ — step by step, in float64,
with Kaldi's own loop structure, so that a slip in one restatement is not silently shared by the other:

  feature-window.cc   NumFrames (snip_edges), ExtractWindow: copy, `window.Add(-window.Sum() / frame_length)`
                      (remove_dc_offset), Preemphasize — IN PLACE, from the last sample down, `w[0] -= coeff * w[0]` —
                      then `window.MulElements(window_function)`, zero padding to the padded window size
  feature-window.cc   FeatureWindowFunction "hamming": 0.54 - 0.46 cos(2 pi i / (N - 1))
  feature-fbank.cc    real FFT of the padded window, ComputePowerSpectrum (bin 0 = re^2, bins 1..N/2-1 = re^2 + im^2, the
                      Nyquist bin kept), MelBanks::Compute, ApplyFloor(FLT_EPSILON), ApplyLog
  mel-computations.cc MelBanks::MelBanks: 1127 ln(1 + f / 700), num_bins + 2 equally spaced mel points between low_freq and
                      Nyquist, per bin the SPARSE range of FFT bins strictly inside (left, right), triangular weights in mel

torchaudio's options for this call: dither 0, energy_floor 1, frame 25 ms / shift 10 ms at 16 kHz (400 / 160 samples),
preemphasis 0.97, remove_dc_offset, round_to_power_of_two (512), snip_edges, low_freq 20, high_freq 0 (-> Nyquist),
num_mel_bins 80, use_energy False, use_log_fbank, use_power, no VTLN, window "hamming".
"""
from __future__ import annotations

import math

import numpy as np

FLT_EPSILON = 1.1920928955078125e-07


def num_frames(num_samples: int, frame_length: int = 400, frame_shift: int = 160) -> int:
    """feature-window.cc NumFrames, snip_edges = true"""
    if num_samples < frame_length:
        return 0
    return 1 + (num_samples - frame_length) // frame_shift


def window_function(frame_length: int = 400) -> np.ndarray:
    a = 2.0 * math.pi / (frame_length - 1)
    return np.array([0.54 - 0.46 * math.cos(a * i) for i in range(frame_length)], dtype=np.float64)


def extract_window(wave: np.ndarray, f: int, frame_length: int = 400, frame_shift: int = 160, padded: int = 512,
                   preemph: float = 0.97, win: np.ndarray | None = None) -> np.ndarray:
    """feature-window.cc ExtractWindow + ProcessWindow for frame f (dither 0, no raw energy)"""
    start = f * frame_shift
    w = np.array(wave[start:start + frame_length], dtype=np.float64)
    w += -w.sum() / frame_length                       # remove_dc_offset
    for i in range(frame_length - 1, 0, -1):           # Preemphasize: in place, backwards
        w[i] -= preemph * w[i - 1]
    w[0] -= preemph * w[0]
    w *= win if win is not None else window_function(frame_length)
    out = np.zeros(padded, dtype=np.float64)
    out[:frame_length] = w
    return out


def mel_scale(f: float) -> float:
    return 1127.0 * math.log(1.0 + f / 700.0)


def mel_banks(num_bins: int = 80, padded: int = 512, sample_freq: float = 16000.0, low_freq: float = 20.0,
              high_freq: float = 0.0):
    """mel-computations.cc MelBanks::MelBanks (vtln_warp 1): list of (first FFT bin, weights) per mel bin"""
    nyquist = 0.5 * sample_freq
    if high_freq <= 0.0:
        high_freq += nyquist
    num_fft_bins = padded // 2
    fft_bin_width = sample_freq / padded
    mel_low, mel_high = mel_scale(low_freq), mel_scale(high_freq)
    delta = (mel_high - mel_low) / (num_bins + 1)
    bins = []
    for b in range(num_bins):
        left, center, right = mel_low + b * delta, mel_low + (b + 1) * delta, mel_low + (b + 2) * delta
        first, weights = -1, []
        for i in range(num_fft_bins):
            mel = mel_scale(fft_bin_width * i)
            if left < mel < right:
                weights.append((mel - left) / (center - left) if mel <= center else (right - mel) / (right - center))
                if first < 0:
                    first = i
        bins.append((first, np.array(weights, dtype=np.float64)))
    return bins


def power_spectrum(padded_window: np.ndarray) -> np.ndarray:
    """srfft + ComputePowerSpectrum: [N/2 + 1] energies"""
    spec = np.fft.rfft(padded_window)
    return spec.real ** 2 + spec.imag ** 2


def fbank(wave: np.ndarray, num_mel_bins: int = 80, frame_length: int = 400, frame_shift: int = 160) -> np.ndarray:
    """compute-fbank-feats on a 1-D waveform already scaled to int16 range  ->  float64 [T, num_mel_bins]"""
    padded = 1 << (frame_length - 1).bit_length()       # round_to_power_of_two
    T = num_frames(len(wave), frame_length, frame_shift)
    win = window_function(frame_length)
    banks = mel_banks(num_mel_bins, padded)
    out = np.empty((T, num_mel_bins), dtype=np.float64)
    for f in range(T):
        p = power_spectrum(extract_window(wave, f, frame_length, frame_shift, padded, 0.97, win))
        for b, (first, wts) in enumerate(banks):
            e = float(np.dot(wts, p[first:first + len(wts)]))
            out[f, b] = math.log(e if e > FLT_EPSILON else FLT_EPSILON)
    return out
