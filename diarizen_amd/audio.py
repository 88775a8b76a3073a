"""Audio ingest for the pipeline (replaces `torchaudio.load` at diarizen/pipelines/inference.py:127
and the `Audio` helper of PA/core/io.py for the one case the hot path needs: 16 kHz WAV, first
channel kept — "force to use the SDM data", inference.py:128).  torchaudio is not available in
this image; RIFF/WAVE PCM16/PCM32/float32 is decoded with the standard library + numpy.
"""
from __future__ import annotations

import io
import wave
from typing import BinaryIO, Tuple, Union

import numpy as np


def load_wav(src: Union[str, bytes, BinaryIO, io.BytesIO]) -> Tuple[np.ndarray, int]:
    """-> (float32 [channels, samples] in [-1, 1), sample_rate)  (torchaudio.load semantics)."""
    if isinstance(src, (bytes, bytearray)):
        src = io.BytesIO(src)
    try:
        with wave.open(src, "rb") as w:
            nch, width, sr, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
            raw = w.readframes(n)
        if width == 2:
            x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
        elif width == 4:
            x = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
        elif width == 1:
            x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
        else:
            raise ValueError(f"unsupported PCM width {width}")
    except wave.Error:
        x, sr, nch = _load_float_wav(src)
    return x.reshape(-1, nch).T.copy(), sr


def _load_float_wav(src):
    """IEEE-float WAVE (format tag 3), which the `wave` module refuses."""
    if hasattr(src, "seek"):
        src.seek(0)
        data = src.read()
    else:
        with open(src, "rb") as f:
            data = f.read()
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError("not a RIFF/WAVE file")
    pos, fmt, body = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], int.from_bytes(data[pos + 4:pos + 8], "little")
        if cid == b"fmt ":
            fmt = data[pos + 8:pos + 8 + size]
        elif cid == b"data":
            body = data[pos + 8:pos + 8 + size]
        pos += 8 + size + (size & 1)
    if fmt is None or body is None:
        raise ValueError("malformed WAVE file")
    tag = int.from_bytes(fmt[0:2], "little")
    nch = int.from_bytes(fmt[2:4], "little")
    sr = int.from_bytes(fmt[4:8], "little")
    bits = int.from_bytes(fmt[14:16], "little")
    if tag != 3 or bits != 32:
        raise ValueError(f"unsupported WAVE format tag {tag} / {bits} bit")
    return np.frombuffer(body, dtype="<f4").astype(np.float32), sr, nch


def first_channel_16k(src, expected_sr: int = 16000) -> np.ndarray:
    x, sr = load_wav(src)
    if sr != expected_sr:
        raise ValueError(f"expected {expected_sr} Hz audio, got {sr} Hz (resampling is host I/O, "
                         "out of scope of the engine: convert the file first)")
    return np.ascontiguousarray(x[0])
