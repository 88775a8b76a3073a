"""Audio ingest for the pipeline (replaces `torchaudio.load` at diarizen/pipelines/inference.py:127
and the `Audio` helper of PA/core/io.py for what the hot path needs: WAV or (r5) FLAC in, first channel kept —
"force to use the SDM data", inference.py:128 — resampled to 16 kHz when the file is at another
rate, as `Audio.downmix_and_resample` does with `torchaudio.functional.resample`, PA/core/io.py:214-218).
torchaudio is not available in this image: RIFF/WAVE (PCM 8/16/24/32, float 32/64, WAVE_FORMAT_EXTENSIBLE) is parsed
directly with numpy, and the resampler restates torchaudio's default algorithm
(sinc_interp_hann, lowpass_filter_width 6, rolloff 0.99; torchaudio==2.1.1 functional/functional.py
`_get_sinc_resample_kernel` / `_apply_sinc_resample_kernel`).  Parity of the resampler is UNPINNED
(no torchaudio here, no fixture in the reference); tests pin its properties instead.
"""
from __future__ import annotations

import io
from typing import BinaryIO, Tuple, Union

import numpy as np


_PCM_GUID_TAIL = bytes.fromhex("000000001000800000aa00389b71")   # KSDATAFORMAT_SUBTYPE_*: tag + this tail


def load_wav(src: Union[str, bytes, BinaryIO, io.BytesIO]) -> Tuple[np.ndarray, int]:
    """-> (float32 [channels, samples] in [-1, 1), sample_rate)  (torchaudio.load semantics).
    RIFF/WAVE is parsed directly: PCM 8 / 16 / 24 / 32 bit, IEEE float 32 / 64, and WAVE_FORMAT_EXTENSIBLE
    (tag 0xFFFE: the real format is the first two bytes of the SubFormat GUID) — everything `torchaudio.load`
    reads from a .wav; the standard `wave` module refuses 24-bit extensible and float files."""
    if isinstance(src, (bytes, bytearray)):
        data = src
    elif hasattr(src, "read"):
        if hasattr(src, "seek"):
            src.seek(0)
        data = src.read()
    else:
        with open(src, "rb") as f:
            data = f.read()
    if bytes(data[:4]) not in (b"RIFF", b"RF64") or bytes(data[8:12]) != b"WAVE":
        raise ValueError("not a RIFF/WAVE file")
    view = memoryview(data)        # (slicing `bytes` copies: a 4 h file was copied twice before it was decoded)
    pos, fmt, body = 12, None, None
    while pos + 8 <= len(data):
        cid, size = bytes(view[pos:pos + 4]), int.from_bytes(view[pos + 4:pos + 8], "little")
        if cid == b"fmt ":
            fmt = bytes(view[pos + 8:pos + 8 + size])
        elif cid == b"data":
            body = view[pos + 8:] if size in (0, 0xFFFFFFFF) else view[pos + 8:pos + 8 + size]   # streamed files
            break
        pos += 8 + size + (size & 1)
    if fmt is None or body is None or len(fmt) < 16:
        raise ValueError("malformed WAVE file")
    tag = int.from_bytes(fmt[0:2], "little")
    nch = int.from_bytes(fmt[2:4], "little")
    sr = int.from_bytes(fmt[4:8], "little")
    block = int.from_bytes(fmt[12:14], "little")
    bits = int.from_bytes(fmt[14:16], "little")
    if tag == 0xFFFE:                                        # WAVE_FORMAT_EXTENSIBLE
        if len(fmt) < 40 or fmt[26:40] != _PCM_GUID_TAIL:
            raise ValueError("unsupported WAVE_FORMAT_EXTENSIBLE sub-format")
        tag = int.from_bytes(fmt[24:26], "little")
    if nch < 1 or sr < 1:
        raise ValueError("malformed WAVE fmt chunk")
    width = block // nch if block else (bits + 7) // 8      # container bytes per sample
    body = body[:len(body) - len(body) % (width * nch)]
    if tag == 1:                                             # integer PCM
        if width == 1:
            x = (np.frombuffer(body, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
        elif width == 2:
            x = np.frombuffer(body, dtype="<i2").astype(np.float32)
            x *= np.float32(1.0 / 32768.0)                   # in place; a power of two, so identical to the division
        elif width == 3:
            b3 = np.frombuffer(body, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
            v = b3[:, 0] | (b3[:, 1] << 8) | (b3[:, 2] << 16)
            v = np.where(v & 0x800000, v - 0x1000000, v)
            x = v.astype(np.float32) / 8388608.0
        elif width == 4:
            x = np.frombuffer(body, dtype="<i4").astype(np.float32)
            x *= np.float32(1.0 / 2147483648.0)
        else:
            raise ValueError(f"unsupported PCM width {width}")
    elif tag == 3:                                           # IEEE float
        if width == 4:
            x = np.frombuffer(body, dtype="<f4").astype(np.float32)
        elif width == 8:
            x = np.frombuffer(body, dtype="<f8").astype(np.float32)
        else:
            raise ValueError(f"unsupported float width {width}")
    else:
        raise ValueError(f"unsupported WAVE format tag {tag} ({bits} bit)")
    if nch == 1:
        return x.reshape(1, -1), sr                              # already [1, samples], contiguous
    return x.reshape(-1, nch).T.copy(), sr


class WavSource:
    """Lazy reader of ONE channel of a RIFF/WAVE file: parses the header, then decodes only the byte range a caller
    asks for — with N ranks sharding one recording (pipeline.device_stage) every rank reads 1/N of the file (+ one
    window of halo) instead of decoding all of it.  Same formats and the same float conversion as
    load_wav(); `needs_resampling` sources fall back to the full decode (the polyphase resampler needs context)."""

    def __init__(self, path, channel: int = 0):
        self.path = str(path)
        with open(self.path, "rb") as f:
            head = f.read(12)
            if head[:4] not in (b"RIFF", b"RF64") or head[8:12] != b"WAVE":
                raise ValueError("not a RIFF/WAVE file")
            fmt = None
            while True:
                ck = f.read(8)
                if len(ck) < 8:
                    raise ValueError("malformed WAVE file")
                cid, size = ck[:4], int.from_bytes(ck[4:8], "little")
                if cid == b"fmt ":
                    fmt = f.read(size)
                    if size & 1:
                        f.seek(1, 1)
                elif cid == b"data":
                    self._data_off = f.tell()
                    f.seek(0, 2)
                    avail = f.tell() - self._data_off
                    self._data_len = avail if size in (0, 0xFFFFFFFF) else min(size, avail)
                    break
                else:
                    f.seek(size + (size & 1), 1)
        if fmt is None or len(fmt) < 16:
            raise ValueError("malformed WAVE file")
        tag = int.from_bytes(fmt[0:2], "little")
        self.channels = int.from_bytes(fmt[2:4], "little")
        self.sample_rate = int.from_bytes(fmt[4:8], "little")
        block = int.from_bytes(fmt[12:14], "little")
        bits = int.from_bytes(fmt[14:16], "little")
        if tag == 0xFFFE:
            if len(fmt) < 40 or fmt[26:40] != _PCM_GUID_TAIL:
                raise ValueError("unsupported WAVE_FORMAT_EXTENSIBLE sub-format")
            tag = int.from_bytes(fmt[24:26], "little")
        if self.channels < 1 or self.sample_rate < 1 or channel >= self.channels:
            raise ValueError("malformed WAVE fmt chunk")
        self._width = block // self.channels if block else (bits + 7) // 8
        self._tag, self._channel = tag, channel
        if (tag, self._width) not in ((1, 1), (1, 2), (1, 3), (1, 4), (3, 4), (3, 8)):
            raise ValueError(f"unsupported WAVE format tag {tag} ({bits} bit)")
        self._frame = self._width * self.channels
        self.num_samples = self._data_len // self._frame

    def read(self, start: int, n: int) -> np.ndarray:
        """float32 [m] = samples start .. start + n of the channel (m < n at the end of the file)"""
        start = max(0, min(int(start), self.num_samples))
        n = max(0, min(int(n), self.num_samples - start))
        with open(self.path, "rb") as f:
            f.seek(self._data_off + start * self._frame)
            body = f.read(n * self._frame)
        w, c, nch = self._width, self._channel, self.channels
        if self._tag == 1:
            if w == 1:
                return (np.frombuffer(body, dtype=np.uint8).reshape(-1, nch)[:, c].astype(np.float32) - 128.0) / 128.0
            if w == 2:
                return np.frombuffer(body, dtype="<i2").reshape(-1, nch)[:, c].astype(np.float32) / 32768.0
            if w == 3:
                b3 = np.frombuffer(body, dtype=np.uint8).reshape(-1, nch, 3)[:, c].astype(np.int32)
                v = b3[:, 0] | (b3[:, 1] << 8) | (b3[:, 2] << 16)
                v = np.where(v & 0x800000, v - 0x1000000, v)
                return v.astype(np.float32) / 8388608.0
            return np.frombuffer(body, dtype="<i4").reshape(-1, nch)[:, c].astype(np.float32) / 2147483648.0
        dt = "<f4" if w == 4 else "<f8"
        return np.ascontiguousarray(np.frombuffer(body, dtype=dt).reshape(-1, nch)[:, c].astype(np.float32))


def resample(x: np.ndarray, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6,
             rolloff: float = 0.99) -> np.ndarray:
    """float32 [..., T] -> [..., ceil(T * new / orig)]: band-limited sinc interpolation with a Hann
    window, evaluated as one strided convolution with `new/gcd` polyphase filters (torchaudio's default
    `functional.resample`).  The filter bank is built in float64 and applied in float32, like torchaudio."""
    import math
    import torch
    orig_freq, new_freq = int(orig_freq), int(new_freq)
    if orig_freq == new_freq:
        return np.ascontiguousarray(x, dtype=np.float32)
    g = math.gcd(orig_freq, new_freq)
    o, n = orig_freq // g, new_freq // g
    base = min(o, n) * rolloff
    width = math.ceil(lowpass_filter_width * o / base)
    idx = torch.arange(-width, width + o, dtype=torch.float64)[None, None] / o
    t = torch.arange(0, -n, -1, dtype=torch.float64)[:, None, None] / n + idx
    t = (t * base).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kernels = torch.where(t == 0, torch.ones_like(t), t.sin() / t) * window * (base / o)
    kernels = kernels.to(torch.float32)                                  # [n, 1, 2 width + o]
    w = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    shape = w.shape
    w = w.reshape(-1, shape[-1])
    length = w.shape[-1]
    w = torch.nn.functional.pad(w, (width, width + o))
    y = torch.nn.functional.conv1d(w[:, None], kernels, stride=o)        # [B, n, frames]
    y = y.transpose(1, 2).reshape(w.shape[0], -1)
    target = math.ceil(n * length / o)
    return y[..., :target].reshape(shape[:-1] + (target,)).numpy()


def _read_all(src) -> bytes:
    if isinstance(src, (bytes, bytearray, memoryview)):
        return src
    if hasattr(src, "read"):
        if hasattr(src, "seek"):
            src.seek(0)
        return src.read()
    with open(src, "rb") as f:
        return f.read()


def load_flac(src: Union[str, bytes, BinaryIO, io.BytesIO], verify_md5: bool = True) -> Tuple[np.ndarray, int]:
    """-> (float32 [channels, samples] in [-1, 1), sample_rate) of a FLAC stream (torchaudio.load semantics: integer samples
    divided by 2^(bits - 1)).  Decoded by the native decoder of libdzn_hip.so (csrc/flac.cpp: host code, written from the
    format specification — libFLAC / libsndfile / torchaudio are not in this image); frame CRCs are verified inside, and the
    STREAMINFO MD5 of the decoded PCM here (a stream whose MD5 field is set cannot decode to wrong samples silently)."""
    import ctypes as C
    import hashlib
    from . import _lib
    data = bytes(_read_all(src))
    lib = _lib.load()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    sr, ch, bits, total = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
    md5 = (C.c_uint8 * 16)()
    rc = lib.dzn_flac_info(buf, len(data), C.byref(sr), C.byref(ch), C.byref(bits), C.byref(total), md5)
    if rc != 0:
        raise ValueError("not a FLAC stream (or a malformed STREAMINFO block)")
    # Output capacity.  A frame is at least 11 bytes (header 5 + one CONSTANT subframe + CRC-16) and carries at most 65535
    # samples per channel, so a stream of len(data) bytes cannot hold more than `bound` samples: a STREAMINFO total beyond it
    # is a lie (36-bit field: up to 2^36 samples = a multi-TB allocation before any frame is checked) and is refused.  A stream
    # of unknown length (total 0: piped encoders) starts from 4 samples per byte and doubles on DZN_E_NOMEM up to the bound -
    # digital silence codes 4096 samples in ~10 bytes, so a fixed guess would call a valid file corrupt.
    bound = (len(data) // 11 + 1) * 65535
    if total.value > bound:
        raise ValueError(f"FLAC STREAMINFO claims {total.value} samples, a {len(data)}-byte stream holds at most {bound}")
    cap = total.value if total.value > 0 else max(4096, len(data) * 4)
    done = C.c_int64()
    while True:
        out = np.empty((cap, ch.value), dtype=np.int32)
        rc = lib.dzn_flac_decode(buf, len(data), out.ctypes.data_as(C.c_void_p), cap, C.byref(done))
        if rc == -2 and total.value == 0 and cap < bound:      # DZN_E_NOMEM (include/dzn.h): capacity too small
            cap = min(cap * 2, bound)
            continue
        break
    if rc != 0:
        raise ValueError(f"FLAC decode failed (code {rc}): corrupt frame (CRC / syntax) or an unsupported stream")
    out = out[:done.value]
    if verify_md5 and any(md5):
        width = (bits.value + 7) // 8
        if width == 4:
            raw = out.astype("<i4").tobytes()
        else:
            raw = out.astype("<i4").view(np.uint8).reshape(-1, 4)[:, :width].tobytes()      # little-endian, sign already extended
        if hashlib.md5(raw).digest() != bytes(md5):
            raise ValueError("FLAC decode: MD5 of the decoded samples differs from the STREAMINFO signature")
    x = out.T.astype(np.float32) * np.float32(1.0 / float(1 << (bits.value - 1)))
    return np.ascontiguousarray(x), sr.value


_FORMATS = ((b"fLaC", "FLAC"), (b"OggS", "Ogg (Vorbis / Opus / FLAC-in-Ogg)"), (b"ID3", "MP3 (ID3 tag)"), (b"\xff\xfb", "MP3"),
            (b"\xff\xf3", "MP3"), (b"\xff\xf2", "MP3"), (b"FORM", "AIFF"), (b".snd", "Sun AU"), (b"\x1aE\xdf\xa3", "Matroska / WebM"),
            (b"NIST_1A", "NIST SPHERE"))


def load_audio(src) -> Tuple[np.ndarray, int]:
    """`torchaudio.load` for what this image can decode natively: RIFF/WAVE (load_wav) and FLAC (load_flac).  Anything else is
    refused BY NAME (the reference accepts whatever torchaudio's backend reads, diarizen/pipelines/inference.py:127; here an
    unsupported container must not surface as "not a RIFF/WAVE file")."""
    data = _read_all(src)
    head = bytes(data[:12])
    if head[:4] in (b"RIFF", b"RF64"):
        return load_wav(data)
    if head[:4] == b"fLaC":
        return load_flac(data)
    if len(head) >= 8 and head[4:8] == b"ftyp":
        raise ValueError("unsupported audio container: MP4 / M4A (AAC) — decodable formats: WAV, FLAC")
    for magic, name in _FORMATS:
        if head.startswith(magic):
            raise ValueError(f"unsupported audio format: {name} — decodable formats: WAV (PCM / float), FLAC")
    raise ValueError("unrecognised audio file (decodable formats: WAV (PCM / float), FLAC)")


def first_channel_16k(src, expected_sr: int = 16000) -> np.ndarray:
    """-> float32 [N] at `expected_sr`: channel 0 of the file (inference.py:128), resampled if needed."""
    x, sr = load_audio(src)
    x0 = np.ascontiguousarray(x[0])
    if sr != expected_sr:
        x0 = resample(x0, sr, expected_sr)
    return x0
