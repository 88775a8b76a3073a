"""Multi-GPU sharding of one recording's windows (one process per GPU, torch.distributed).

Windows are independent units for both device stages (PA/core/inference.py:316-381 batches
arbitrary windows; embeddings are per (window, speaker)), so rank r of G takes the contiguous
window range [r*ceil(C/G), (r+1)*ceil(C/G)) — contiguous so a rank touches a contiguous slice of
the waveform — and ONE collective follows: an all-gather of the per-window results
(u8 decisions [c, L, 4] + f32 embeddings [c, 4, 256], 5.7 KB per 8 s window) before the host
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


def gather_windows(seg: torch.Tensor, emb: Optional[torch.Tensor], expected_total: Optional[int] = None):
    """all-gather per-window results of every rank (padded to equal counts), in window order.
    expected_total: the number of windows of the whole recording — every one of the world_size ranks must have
    contributed its block and the blocks must add up (a rank that silently ran alone, e.g. a collective that fell back
    to a 1-rank group, is an error, not a short result)."""
    d = _dist()
    if d is None or d.get_world_size() == 1:
        if expected_total is not None and seg.shape[0] != expected_total:
            raise RuntimeError(f"{seg.shape[0]} windows computed, {expected_total} expected")
        return seg, emb
    g = d.get_world_size()
    if d.get_backend() == "gloo" and seg.is_cuda:
        # gloo has no device all_gather: stage through the host (debug / single-GPU rehearsal of the N > 1 path;
        # production runs use the "nccl" backend = RCCL over xGMI, device to device)
        dev = seg.device
        s2, e2 = gather_windows(seg.cpu(), emb.cpu() if emb is not None else None, expected_total)
        return s2.to(dev), (e2.to(dev) if e2 is not None else None)
    n = torch.tensor([seg.shape[0]], device=seg.device, dtype=torch.int64)
    counts = [torch.zeros_like(n) for _ in range(g)]
    d.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    if len(counts) != g or counts[d.get_rank()] != seg.shape[0]:
        raise RuntimeError(f"all_gather of the window counts returned {counts} on rank {d.get_rank()} of {g}")
    if expected_total is not None:
        blocks = [shard_range(expected_total, r, g) for r in range(g)]
        if counts != [hi - lo for lo, hi in blocks]:
            raise RuntimeError(f"ranks contributed {counts} windows, the block partition of {expected_total} windows over {g} "
                               f"ranks is {[hi - lo for lo, hi in blocks]}")
    cap = max(max(counts), 1)

    def gather(t: torch.Tensor) -> torch.Tensor:
        pad = t.new_zeros((cap,) + tuple(t.shape[1:]))
        pad[: t.shape[0]] = t
        outs = [torch.empty_like(pad) for _ in range(g)]
        d.all_gather(outs, pad.contiguous())
        return torch.cat([o[:c] for o, c in zip(outs, counts)], dim=0)

    return gather(seg), (gather(emb) if emb is not None else None)
