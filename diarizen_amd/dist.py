"""Multi-GPU sharding of one recording's windows (one process per GPU, torch.distributed).

Windows are independent units for both device stages (PA/core/inference.py:316-381 batches
arbitrary windows; embeddings are per (window, speaker)), so rank r of G takes the contiguous
window range [r*ceil(C/G), (r+1)*ceil(C/G)) — contiguous so a rank touches a contiguous slice of
the waveform — and ONE collective follows: an all-gather of the per-window results
(u8 decisions [c, L, 4] + f32 embeddings [c, 4, 256], 5.7 KB per 8 s window, packed into one byte buffer) before the host
clustering, which needs every window (PA/pipelines/clustering.py:285-322).  On ROCm the "nccl"
backend is RCCL; on an 8-GPU xGMI node a 4 h recording moves ~13 MB per rank, i.e. microseconds
of wire time, so a single un-chunked all-gather per tensor is the right shape.  The reference has
no multi-GPU inference at all (device cuda:0 hard-coded, diarizen/pipelines/inference.py:57).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch


def _dist():
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def rank() -> int:
    d = _dist()
    return d.get_rank() if d else 0


def world_size() -> int:
    d = _dist()
    return d.get_world_size() if d else 1


def shard_range(num_windows: int, r: int, g: int) -> Tuple[int, int]:
    """contiguous block partition; trailing ranks may be empty"""
    per = -(-num_windows // g) if g > 0 else num_windows
    lo = min(r * per, num_windows)
    return lo, min(lo + per, num_windows)


def my_window_range(num_windows: int) -> Optional[Tuple[int, int]]:
    g = world_size()
    if g == 1:
        return None
    return shard_range(num_windows, rank(), g)


HEADER_BYTES = 16          # int64 rank, int64 window count: every rank's block says whose it is and how long


def _pack(seg: torch.Tensor, emb: Optional[torch.Tensor], cap: int, r: int) -> Tuple[torch.Tensor, int, int]:
    """one rank's block of the exchange as bytes: [header | cap rows of (u8 decisions, padded to 4 B | f32 embeddings)]"""
    c = seg.shape[0]
    seg_b = int(seg[0].numel()) if c else int(torch.tensor(seg.shape[1:]).prod())
    seg_bp = (seg_b + 3) & ~3
    emb_b = 0 if emb is None else 4 * (int(emb[0].numel()) if c else int(torch.tensor(emb.shape[1:]).prod()))
    row = seg_bp + emb_b
    buf = torch.zeros(HEADER_BYTES + cap * row, device=seg.device, dtype=torch.uint8)
    buf[:HEADER_BYTES] = torch.tensor([r, c], dtype=torch.int64).view(torch.uint8).to(seg.device, non_blocking=True)
    k = min(c, cap)        # a rank whose count is not its block's still takes part: its header says so and EVERY rank refuses
    if k:
        rows = buf[HEADER_BYTES:].view(cap, row)
        rows[:k, :seg_b] = seg[:k].reshape(k, seg_b)
        if emb is not None:
            rows[:k, seg_bp:] = emb[:k].contiguous().view(torch.uint8).reshape(k, emb_b)
    return buf, seg_bp, row


def gather_windows(seg: torch.Tensor, emb: Optional[torch.Tensor], expected_total: Optional[int] = None,
                   to_host: bool = False):
    """All-gather the per-window results of every rank, in window order: ONE collective on one packed byte buffer
    (`all_gather_into_tensor`; r5 ran three list all-gathers and read g counts back one by one).
    expected_total: the number of windows of the whole recording.  With it the per-rank counts are not exchanged at all - they
    ARE the block partition `shard_range(expected_total, r, g)` - and every block carries a header (rank, count) that is checked
    after the collective: every one of the world_size ranks must have contributed exactly its block (a rank that silently ran
    alone, e.g. a collective that fell back to a 1-rank group, or a rank that computed a different range, is an error, not a
    short result).  Without it the counts are exchanged first (one more small collective).
    to_host: return CPU tensors from ONE device-to-host copy of the gathered buffer (what the host stage wants anyway)."""
    d = _dist()
    if d is None or d.get_world_size() == 1:
        if expected_total is not None and seg.shape[0] != expected_total:
            raise RuntimeError(f"{seg.shape[0]} windows computed, {expected_total} expected")
        return (seg.cpu(), emb.cpu() if emb is not None else None) if to_host else (seg, emb)
    g, me = d.get_world_size(), d.get_rank()
    if d.get_backend() == "gloo" and seg.is_cuda:
        # gloo has no device all_gather: stage through the host (debug / single-GPU rehearsal of the N > 1 path;
        # production runs use the "nccl" backend = RCCL over xGMI, device to device)
        dev = seg.device
        s2, e2 = gather_windows(seg.cpu(), emb.cpu() if emb is not None else None, expected_total)
        return (s2, e2) if to_host else (s2.to(dev), (e2.to(dev) if e2 is not None else None))
    if expected_total is not None:
        counts = [hi - lo for lo, hi in (shard_range(expected_total, r, g) for r in range(g))]
    else:
        n = torch.tensor([seg.shape[0]], device=seg.device, dtype=torch.int64)
        allc = torch.zeros(g, device=seg.device, dtype=torch.int64)
        d.all_gather_into_tensor(allc, n)
        counts = allc.tolist()
        if counts[me] != seg.shape[0]:
            raise RuntimeError(f"all_gather of the window counts returned {counts} on rank {me} of {g}")
    cap = max(max(counts), 1)
    mine, seg_bp, row = _pack(seg, emb, cap, me)
    out = torch.empty(g * mine.numel(), device=seg.device, dtype=torch.uint8)
    d.all_gather_into_tensor(out, mine)
    blocks = out.view(g, mine.numel())
    if to_host:
        blocks = blocks.cpu()                                       # the one copy (synchronises the current stream)
        heads = blocks[:, :HEADER_BYTES].contiguous().view(torch.int64)
    else:
        heads = blocks[:, :HEADER_BYTES].contiguous().cpu().view(torch.int64)
    got = heads.tolist()
    want = [[r, c] for r, c in enumerate(counts)]
    if got != want:
        raise RuntimeError(f"ranks contributed (rank, windows) {got}, expected {want}"
                           + (f": the block partition of {expected_total} windows over {g} ranks" if expected_total is not None else ""))
    rows = blocks[:, HEADER_BYTES:].reshape(g, cap, row)
    keep = torch.cat([rows[r, :c] for r, c in enumerate(counts)], dim=0)           # [C, row] in window order
    C = keep.shape[0]
    seg_b = int(torch.tensor(seg.shape[1:]).prod())
    seg_g = keep[:, :seg_b].clone().view((C,) + tuple(seg.shape[1:]))
    emb_g = None
    if emb is not None:
        emb_g = keep[:, seg_bp:].clone().view(torch.float32).view((C,) + tuple(emb.shape[1:]))
    return seg_g, emb_g
