"""Sliding-window runtime: the device half of `Inference.slide` + `get_embeddings`.

Mirrors (reference, PA/ = pyannote-audio/pyannote/audio/):
  * PA/core/inference.py:237-409 `Inference.slide` with skip_aggregation=True — window
    arithmetic :265-299 (W = floor(dur*sr), S = round(step*sr), zero-padded last window),
    batching :316-343, powerset hard conversion :226;
  * diarizen/pipelines/inference.py:131-132 median filter;
  * PA/pipelines/speaker_diarization.py:268-360 `get_embeddings` (overlap-excluded masks, one
    embedding per (window, local speaker)).
Differences by design (MI355X-first): the recording is uploaded once and windows are strided
views of it in HBM; segmentation -> masks -> embedding run back-to-back on the device for each
batch of windows (no host round trip, no per-(window,speaker) crop loop), and the ResNet trunk is
computed once per window for its 4 speaker masks.  Only u8 decisions [C, L, 4] and f32 embeddings
[C, 4, 256] ever leave the device.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from contextlib import nullcontext as _nullcontext

from .engine import Engine


def window_plan(num_samples: int, window: int, step: int) -> Tuple[int, bool]:
    """(number of full windows, has_zero_padded_last)  — PA/core/inference.py:285-299."""
    if num_samples >= window:
        num_chunks = (num_samples - window) // step + 1
        has_last = (num_samples - window) % step > 0
    else:
        num_chunks = 0
        has_last = True
    return num_chunks, has_last


@dataclass
class SlideResult:
    segmentations: torch.Tensor            # u8 [C, L, S] (median-filtered hard decisions), device
    embeddings: Optional[torch.Tensor]     # f32 [C, S, dim] or None, device
    window: int
    step: int
    num_frames: int


class WindowRunner:
    def __init__(self, engine: Engine, duration: float, step_ratio: float = 0.1, batch_size: int = 32,
                 median_size: int = 11, exclude_overlap: bool = True, sample_rate: int = 16000,
                 extra_engines: Tuple[Engine, ...] = (), engine_factory=None, max_engines: int = 1):
        """extra_engines (r4): further handles with the SAME weights (one workspace each).  Consecutive batches then
        alternate over [engine, *extra_engines], each on its own HIP stream, so that independent batches overlap on the
        device — a batch's HBM-bound kernels run beside another batch's matrix-bound ones, launch tails fill, and at small
        batches (BASELINE configs[1]: 32 windows) the under-filled chip takes two launches at once.  Results are the same
        bits: a window's result does not depend on the batch or the handle it runs in.
        engine_factory / max_engines (r5): instead of handing in ready handles, let the runner CREATE further handles the first
        time a run has more than one batch — and only while the device keeps a reserve free afterwards (each handle is a full
        copy of the weights plus a workspace for `batch_size` windows: two of them created up front could take most of the
        then-free HBM before the embedding side, torch or RCCL had allocated anything).  A handle that cannot be
        created is simply not used."""
        self.engine = engine
        self.engines = (engine,) + tuple(extra_engines)
        self._factory = engine_factory
        self._max_engines = max(int(max_engines), len(self.engines))
        self._owned = []                 # handles this runner created (closed by close())
        self._streams = None
        self.sample_rate = sample_rate
        self.duration = duration
        self.window = int(math.floor(duration * sample_rate))               # inference.py:265
        self.step = int(round(step_ratio * duration * sample_rate))         # :266
        self.batch_size = min(batch_size, engine.max_batch)
        self.median_size = median_size
        self.exclude_overlap = exclude_overlap
        if self.window > engine.max_samples:
            raise ValueError("window longer than the engine's max_samples")
        self.num_frames = engine.num_frames(self.window)
        # minimum number of frames for the clean mask (speaker_diarization.py:274-278); the
        # embedding model needs one 400-sample fbank frame (speaker_verification.py:677-691)
        self.min_num_samples = 400
        self.min_num_frames = math.ceil(self.num_frames * self.min_num_samples / self.window)

    RESERVE_BYTES = 32 << 30             # HBM that must stay free after a further handle is created (or a quarter of the device)

    def _grow(self, device) -> None:
        """create further engine handles up to max_engines, each only if free HBM - its size stays above the reserve"""
        while self._factory is not None and len(self.engines) < self._max_engines:
            free, total = torch.cuda.mem_get_info(device)
            need = int(self.engine.workspace_bytes)
            if free - need < max(self.RESERVE_BYTES, total // 4):
                self._declined(f"a further engine handle needs {need >> 20} MiB, {free >> 20} MiB of {total >> 20} MiB are free and "
                               f"{max(self.RESERVE_BYTES, total // 4) >> 20} MiB must stay free")
                break
            try:
                e = self._factory()
            except Exception as ex:      # MemoryError / DznError: run on the handles that exist
                self._declined(f"creating a further engine handle failed: {ex}")
                break
            self._owned.append(e)
            self.engines = self.engines + (e,)
            self._streams = None

    grow_declined: Optional[str] = None      # why fewer handles than asked for are in use (None: all were created / none asked)

    def _declined(self, why: str) -> None:
        """num_streams is an upper bound; say ONCE when it is not reached (on a device with <= 64 GB the documented
        default of two streams would otherwise be silently one)"""
        import warnings
        asked = self._max_engines
        self._max_engines = len(self.engines)
        if self.grow_declined is None:
            self.grow_declined = why
            warnings.warn(f"diarizen_amd: running on {len(self.engines)} engine handle(s) / stream(s) instead of {asked}: {why}",
                          RuntimeWarning, stacklevel=3)

    def close(self) -> None:
        """release the handles this runner created (their weights and workspaces) — not the caller's engine"""
        for e in self._owned:
            e.close()
        self.engines = tuple(e for e in self.engines if e not in self._owned)
        self._owned = []
        self._streams = None

    def num_windows(self, num_samples: int) -> int:
        n, last = window_plan(num_samples, self.window, self.step)
        return n + int(last)

    def windows_view(self, wave: torch.Tensor) -> torch.Tensor:
        """[C, W] strided view over the zero-extended recording (no copy of the samples)."""
        assert wave.dim() == 1
        C = self.num_windows(wave.numel())
        need = (C - 1) * self.step + self.window
        if need > wave.numel():
            wave = torch.cat([wave, wave.new_zeros(need - wave.numel())])
        return torch.as_strided(wave, (C, self.window), (self.step, 1))

    def run(self, wave: torch.Tensor, with_embeddings: bool = True,
            window_range: Optional[Tuple[int, int]] = None, hook=None) -> SlideResult:
        """wave: f32 [N_total] on the device.  window_range=(c0, c1) restricts to a contiguous
        run of windows (multi-GPU sharding).  Enqueue only; results are device tensors.
        hook: the progress callback of `Inference.slide` (PA/core/inference.py:308-340), called as
        hook(completed=<windows done>, total=<windows>) before the first and after every batch; a hook makes the
        call wait for each batch (otherwise "completed" would only mean "enqueued")."""
        views = self.windows_view(wave)
        c0, c1 = window_range if window_range is not None else (0, views.shape[0])
        return self.run_views(views, c0, c1, with_embeddings=with_embeddings, hook=hook)

    def run_views(self, views: torch.Tensor, c0: int, c1: int, with_embeddings: bool = True, hook=None) -> SlideResult:
        """the batch loop of run() over rows c0 .. c1 of a [C, window] (strided) view of device samples — the streaming
        session hands in a view of its pre-zeroed device ring, so nothing is re-concatenated as audio arrives"""
        eng = self.engine
        wave = views
        C = max(c1 - c0, 0)
        S = eng.seg.max_speakers_per_chunk
        seg = torch.empty((C, self.num_frames, S), device=wave.device, dtype=torch.uint8)
        emb = (torch.empty((C, S, eng.emb.embed_dim), device=wave.device, dtype=torch.float32)
               if with_embeddings else None)
        # balanced batches: ceil(C / batch_size) launches of (almost) equal size instead of full batches plus a short
        # tail (2241 windows at batch 256: 9 x 249 rather than 8 x 256 + 193) — windows are independent and the
        # engines are batch-invariant, so only the tail efficiency changes
        nb = max(1, -(-C // self.batch_size))
        bs = max(1, -(-C // nb))
        if hook is not None:
            hook(completed=0, total=C)
        if nb > 1 and hook is None and len(self.engines) < self._max_engines:
            self._grow(wave.device)
        multi = len(self.engines) > 1 and hook is None and nb > 1
        cur = torch.cuda.current_stream(wave.device)
        if multi:
            if self._streams is None:
                self._streams = [torch.cuda.Stream(device=wave.device) for _ in self.engines]
            for st in self._streams:
                st.wait_stream(cur)                 # inputs / output buffers were produced on the caller's stream
        for bi, s0 in enumerate(range(c0, c1, bs)):
            s1 = min(s0 + bs, c1)
            k = bi % len(self.engines) if multi else 0
            e = self.engines[k]
            with torch.cuda.stream(self._streams[k]) if multi else _nullcontext():
                chunk = views[s0:s1].contiguous()
                _, ml = e.segment(chunk, want_logp=False)
                filt, masks = e.prepare_masks(ml, self.median_size, self.exclude_overlap,
                                              self.min_num_frames if self.exclude_overlap else -1,
                                              want_masks=with_embeddings)
                seg[s0 - c0:s1 - c0] = filt
                if with_embeddings:
                    emb[s0 - c0:s1 - c0] = e.embed(chunk, masks)
            if hook is not None:
                torch.cuda.current_stream(wave.device).synchronize()
                hook(completed=s1 - c0, total=C)
        if multi:
            for st in self._streams:
                cur.wait_stream(st)                 # the caller's stream sees every batch's results
        return SlideResult(seg, emb, self.window, self.step, self.num_frames)
