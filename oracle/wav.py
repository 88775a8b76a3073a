"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product (diarizen_amd/).

The fixture generator's own WAV reader (python's `wave` module: 16-bit PCM only, which is what the reference's
example/EN2002a_30s.wav is) — channel 0 as float32 = int16 / 32768, the conversion torchaudio.load applies
(diarizen/pipelines/inference.py:127-128).  tests/test_oracle.py checks that the product's RIFF parser
(diarizen_amd/audio.py) returns the very same samples."""
import wave

import numpy as np


def first_channel_pcm16(path, expected_sr: int = 16000) -> np.ndarray:
    with wave.open(str(path)) as w:
        if w.getsampwidth() != 2 or w.getcomptype() != "NONE":
            raise ValueError(f"{path}: the oracle reads 16-bit PCM only")
        if w.getframerate() != expected_sr:
            raise ValueError(f"{path}: {w.getframerate()} Hz, expected {expected_sr}")
        raw = w.readframes(w.getnframes())
        return np.ascontiguousarray(np.frombuffer(raw, dtype="<i2").reshape(-1, w.getnchannels())[:, 0].astype(np.float32) / 32768.0)
