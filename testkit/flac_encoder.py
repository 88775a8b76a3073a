"""A small FLAC ENCODER — test infrastructure for diarizen_amd/csrc/flac.cpp (no libFLAC / flac binary in this image, so the
test streams are made here, from the format specification: RFC 9639).  Not a good compressor: it exists to put every construct
the decoder implements into a bit stream — CONSTANT / VERBATIM / FIXED order 0-4 / LPC subframes, Rice partitions with 4- and
5-bit parameters and escaped partitions, wasted bits, the three stereo decorrelations, explicit and table-coded block sizes /
sample rates / sample sizes, multi-byte frame numbers — with correct CRC-8 / CRC-16 and the STREAMINFO MD5 of the input.

    encode(samples int [n, channels], sample_rate, bits, frames=[FrameSpec, ...] | blocksize=...) -> bytes
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np


class BitWriter:
    def __init__(self):
        self.buf = bytearray()
        self.acc = 0
        self.n = 0

    def put(self, value: int, bits: int):
        if bits == 0:
            return
        value &= (1 << bits) - 1
        self.acc = (self.acc << bits) | value
        self.n += bits
        while self.n >= 8:
            self.n -= 8
            self.buf.append((self.acc >> self.n) & 0xff)
        self.acc &= (1 << self.n) - 1

    def unary(self, q: int):
        while q >= 32:
            self.put(0, 32)
            q -= 32
        self.put(1, q + 1)

    def align(self):
        if self.n:
            self.put(0, 8 - self.n)

    def bytes(self) -> bytes:
        assert self.n == 0
        return bytes(self.buf)


def crc8(data: bytes) -> int:
    c = 0
    for b in data:
        c ^= b
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xff if c & 0x80 else (c << 1) & 0xff
    return c


def crc16(data: bytes) -> int:
    c = 0
    for b in data:
        c ^= b << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x8005) & 0xffff if c & 0x8000 else (c << 1) & 0xffff
    return c


def utf8_number(v: int) -> bytes:
    """the extended UTF-8 coding of frame / sample numbers (up to 36 bits, 7 bytes)"""
    if v < 0x80:
        return bytes([v])
    for nbytes, limit in ((2, 1 << 11), (3, 1 << 16), (4, 1 << 21), (5, 1 << 26), (6, 1 << 31), (7, 1 << 36)):
        if v < limit:
            cont = [0x80 | ((v >> (6 * i)) & 0x3f) for i in range(nbytes - 1)][::-1]
            lead = ((0xff << (8 - nbytes)) & 0xff) | (v >> (6 * (nbytes - 1)))
            return bytes([lead] + cont)
    raise ValueError("number too large")


@dataclass
class SubSpec:
    kind: str = "fixed"            # "constant" | "verbatim" | "fixed" | "lpc"
    order: int = 2                 # fixed: 0-4, lpc: 1-32
    coefs: Optional[Sequence[int]] = None   # lpc: quantised coefficients
    precision: int = 12
    shift: int = 10
    partition_order: int = 0
    rice5: bool = False            # 5-bit Rice parameters (coding method 1)
    escape: bool = False           # write every partition as an escaped (raw) partition
    wasted: int = 0                # wasted bits (the samples must be multiples of 2^wasted)


@dataclass
class FrameSpec:
    blocksize: int = 4096
    stereo: str = "independent"    # "independent" | "left_side" | "side_right" | "mid_side"
    subs: List[SubSpec] = field(default_factory=lambda: [SubSpec()])
    explicit_rate: bool = False    # write the sample rate into the header instead of "from STREAMINFO"


BS_CODES = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12, 8192: 13, 16384: 14, 32768: 15}
SS_CODES = {8: 1, 12: 2, 16: 4, 20: 5, 24: 6, 32: 7}


def _zigzag(r: int) -> int:
    return (r << 1) if r >= 0 else ((-r) << 1) - 1


def _write_residual(bw: BitWriter, res: Sequence[int], blocksize: int, pred_order: int, spec: SubSpec):
    bw.put(1 if spec.rice5 else 0, 2)
    po = spec.partition_order
    assert blocksize % (1 << po) == 0 and (blocksize >> po) > pred_order
    bw.put(po, 4)
    pbits, esc = (5, 31) if spec.rice5 else (4, 15)
    i = 0
    for part in range(1 << po):
        cnt = (blocksize >> po) - (pred_order if part == 0 else 0)
        chunk = res[i:i + cnt]
        i += cnt
        if spec.escape:
            nb = max([1] + [int(abs(int(r))).bit_length() + 1 for r in chunk])
            bw.put(esc, pbits)
            bw.put(nb, 5)
            for r in chunk:
                bw.put(int(r), nb)
            continue
        zz = [_zigzag(int(r)) for r in chunk]
        best_k, best_bits = 0, None
        for k in range(0, esc):
            bits = sum((z >> k) + 1 + k for z in zz)
            if best_bits is None or bits < best_bits:
                best_k, best_bits = k, bits
        bw.put(best_k, pbits)
        for z in zz:
            bw.unary(z >> best_k)
            bw.put(z & ((1 << best_k) - 1), best_k)
    assert i == len(res)


def _write_subframe(bw: BitWriter, s: Sequence[int], bps: int, spec: SubSpec):
    n = len(s)
    w = spec.wasted
    if w:
        assert all(int(v) % (1 << w) == 0 for v in s), "wasted bits need samples that are multiples of 2^wasted"
        s = [int(v) >> w for v in s]
        bps -= w
    else:
        s = [int(v) for v in s]
    bw.put(0, 1)
    if spec.kind == "constant":
        assert all(v == s[0] for v in s)
        bw.put(0b000000, 6)
    elif spec.kind == "verbatim":
        bw.put(0b000001, 6)
    elif spec.kind == "fixed":
        bw.put(0b001000 | spec.order, 6)
    else:
        bw.put(0b100000 | (spec.order - 1), 6)
    if w:
        bw.put(1, 1)
        bw.unary(w - 1)
    else:
        bw.put(0, 1)
    if spec.kind == "constant":
        bw.put(s[0], bps)
        return
    if spec.kind == "verbatim":
        for v in s:
            bw.put(v, bps)
        return
    order = spec.order
    for v in s[:order]:
        bw.put(v, bps)
    if spec.kind == "fixed":
        co = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}[order]
        res = [s[i] - sum(c * s[i - 1 - j] for j, c in enumerate(co)) for i in range(order, n)]
    else:
        co = list(spec.coefs)
        assert len(co) == order
        bw.put(spec.precision - 1, 4)
        bw.put(spec.shift, 5)
        for c in co:
            bw.put(int(c), spec.precision)
        res = [s[i] - (sum(int(c) * s[i - 1 - j] for j, c in enumerate(co)) >> spec.shift) for i in range(order, n)]
    _write_residual(bw, res, n, order, spec)


def encode(samples: np.ndarray, sample_rate: int, bits: int, frames: Optional[List[FrameSpec]] = None, blocksize: int = 4096,
           md5: bool = True, total_in_header: bool = True) -> bytes:
    x = np.asarray(samples)
    if x.ndim == 1:
        x = x[:, None]
    n, C = x.shape
    if frames is None:
        frames = [FrameSpec(blocksize=blocksize)]
    # ---- STREAMINFO ----
    width = (bits + 7) // 8
    raw = x.astype("<i4").view(np.uint8).reshape(-1, 4)[:, :width].tobytes() if width < 4 else x.astype("<i4").tobytes()
    digest = hashlib.md5(raw).digest() if md5 else bytes(16)
    sizes = []
    pos, fi = 0, 0
    plan = []
    while pos < n:
        spec = frames[fi % len(frames)]
        bs = min(spec.blocksize, n - pos)
        plan.append((spec, pos, bs))
        sizes.append(bs)
        pos += bs
        fi += 1
    bw = BitWriter()
    bw.put(min(sizes[:-1] or sizes), 16)
    bw.put(max(sizes), 16)
    bw.put(0, 24)
    bw.put(0, 24)
    bw.put(sample_rate, 20)
    bw.put(C - 1, 3)
    bw.put(bits - 1, 5)
    bw.put(n if total_in_header else 0, 36)
    info = bw.bytes() + digest
    out = bytearray(b"fLaC")
    out += bytes([0x00]) + len(info).to_bytes(3, "big") + info                 # STREAMINFO, not last
    pad = bytes(7)
    out += bytes([0x80 | 1]) + len(pad).to_bytes(3, "big") + pad               # PADDING, last
    # ---- frames ----
    for fno, (spec, p0, bs) in enumerate(plan):
        blk = x[p0:p0 + bs].astype(np.int64)
        hw = BitWriter()
        hw.put(0b11111111111110, 14)
        hw.put(0, 1)
        hw.put(0, 1)                                       # fixed-blocksize stream: the coded number is the frame number
        bs_code = BS_CODES.get(bs, 6 if bs <= 256 else 7)
        hw.put(bs_code, 4)
        rate_code = {8000: 4, 16000: 5, 22050: 6, 24000: 7, 32000: 8, 44100: 9, 48000: 10, 96000: 11}.get(sample_rate, 13) if spec.explicit_rate else 0
        hw.put(rate_code, 4)
        if C == 2 and spec.stereo != "independent":
            hw.put({"left_side": 8, "side_right": 9, "mid_side": 10}[spec.stereo], 4)
        else:
            hw.put(C - 1, 4)
        hw.put(SS_CODES.get(bits, 0) if fno % 2 == 0 else 0, 3)     # alternate between the table code and "from STREAMINFO"
        hw.put(0, 1)
        for b in utf8_number(fno):
            hw.put(b, 8)
        if bs_code == 6:
            hw.put(bs - 1, 8)
        elif bs_code == 7:
            hw.put(bs - 1, 16)
        if rate_code == 13:
            hw.put(sample_rate, 16)
        head = hw.bytes()
        fw = BitWriter()
        for b in head:
            fw.put(b, 8)
        fw.put(crc8(head), 8)
        chans = [blk[:, c] for c in range(C)]
        bps = [bits] * C
        if C == 2 and spec.stereo != "independent":
            L, R = chans
            if spec.stereo == "left_side":
                chans, bps = [L, L - R], [bits, bits + 1]
            elif spec.stereo == "side_right":
                chans, bps = [L - R, R], [bits + 1, bits]
            else:
                chans, bps = [(L + R) >> 1, L - R], [bits, bits + 1]
        for c in range(C):
            _write_subframe(fw, [int(v) for v in chans[c]], bps[c], spec.subs[c % len(spec.subs)])
        fw.align()
        body = fw.bytes()
        out += body + crc16(body).to_bytes(2, "big")
    return bytes(out)
