"""Streaming / online use of the hot path (row f3 of SURVEY §8f: "window scheduler": PA/core/inference.py:316-343 batches
windows of a file that is already complete; here windows are scheduled AS THE AUDIO ARRIVES).

    for t, annotation in pipeline.stream(chunks, sess_name="meeting"):      # chunks: iterable of float32 arrays
        ...                                                                 # provisional turns up to t seconds
    # the last item is the final annotation — byte-identical RTTM to pipeline(<the whole file>)

Design (MI355X-first, same kernels as the offline path):
  * ingest: every chunk lands in a slot of a PINNED host ring and is copied to the device on a dedicated copy stream
    (`non_blocking` H2D, one event per slot); the compute stream only waits for the event of the newest slot a window
    needs, so upload and compute overlap and the caller's buffer is free as soon as feed() returns;
  * the device holds the recording so far in ONE pre-zeroed buffer (4 h of 16 kHz mono = 0.92 GB of the 288 GB); windows are
    rows of a strided view over it, exactly as offline (inference.WindowRunner), so a window is computed once, when its last
    sample has arrived, and never again.  Samples past the end read as zeros: the reference's zero-padded last window
    (PA/core/inference.py:293-299) needs no special case at finish();
  * per-window results are batch-invariant (tests/test_properties_gpu.py), hence the streamed decisions / embeddings —
    and the final RTTM — equal the offline ones bit for bit;
  * provisional output: every `refresh_s` seconds of new audio the host stage (counting, clustering, reconstruction,
    Binarize) runs over the windows finished so far.  Speaker labels of provisional annotations are NOT stable across
    refreshes (each is a fresh clustering), the final one is the offline result.
"""
from __future__ import annotations

from typing import Iterable, Iterator, Optional, Tuple

import numpy as np
import torch

from .core import Annotation
from .inference import window_plan


def complete_windows(num_samples: int, window: int, step: int) -> int:
    """windows whose last sample has arrived (the zero-padded tail window only exists once the stream has ended)"""
    return 0 if num_samples < window else (num_samples - window) // step + 1


class StreamingSession:
    def __init__(self, pipeline, sess_name: Optional[str] = None, max_seconds: float = 4 * 3600.0, refresh_s: Optional[float] = 8.0,
                 slot_seconds: float = 10.0, slots: int = 4):
        self.pipe = pipeline
        self.sess_name = sess_name
        self.refresh_s = refresh_s
        r = pipeline._runner
        self.runner = r
        self.sr = r.sample_rate
        dev = pipeline.device
        self.capacity = int(max_seconds * self.sr) + r.window           # + one window of zeros behind the last sample
        self.dev_wave = torch.zeros(self.capacity, device=dev, dtype=torch.float32)
        self.views = torch.as_strided(self.dev_wave, ((self.capacity - r.window) // r.step + 1, r.window), (r.step, 1))
        self.slot_samples = int(slot_seconds * self.sr)
        self.ring = [torch.empty(self.slot_samples, dtype=torch.float32).pin_memory() for _ in range(slots)]
        self.slot_free = [None] * slots                                 # event: the slot's H2D copy has completed
        self.next_slot = 0
        self.copy_stream = torch.cuda.Stream(device=dev)
        # the zero fill of dev_wave above is queued on the CURRENT stream; the chunk copies run on copy_stream — without this
        # edge the first copies could land before the memset and be zeroed by it.  record_stream: the caching allocator
        # must not recycle dev_wave while copies on the other stream are pending.
        self.copy_stream.wait_stream(torch.cuda.current_stream(dev))
        self.dev_wave.record_stream(self.copy_stream)
        self.n = 0                                                      # samples received
        self.done = 0                                                   # windows computed
        self.seg = []                                                   # per batch: u8 [c, L, S] host arrays
        self.emb = []
        self.last_refresh_n = 0
        self.last_copy = None
        self.finished = False
        self.stats = {"uploads": 0, "launches": 0, "refreshes": 0}

    # ------------------------------------------------------------------ ingest
    def _upload(self, x: np.ndarray) -> None:
        off = 0
        while off < len(x):
            k = min(self.slot_samples, len(x) - off)
            i = self.next_slot
            if self.slot_free[i] is not None:
                self.slot_free[i].synchronize()                         # ring full: wait for the oldest copy only
            self.ring[i][:k].copy_(torch.from_numpy(x[off:off + k]))
            with torch.cuda.stream(self.copy_stream):
                self.dev_wave[self.n + off:self.n + off + k].copy_(self.ring[i][:k], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.copy_stream)
            self.slot_free[i] = ev
            self.last_copy = ev
            self.next_slot = (i + 1) % len(self.ring)
            self.stats["uploads"] += 1
            off += k

    def _compute(self, upto: int) -> None:
        """run windows done .. upto on the compute stream, behind the newest upload"""
        if upto <= self.done:
            return
        torch.cuda.current_stream(self.pipe.device).wait_event(self.last_copy)
        res = self.runner.run_views(self.views, self.done, upto, with_embeddings=True)
        self.seg.append(res.segmentations.cpu().numpy())               # 5.7 KB per window
        self.emb.append(res.embeddings.cpu().numpy())
        self.done = upto
        self.stats["launches"] += 1

    def feed(self, samples) -> Optional[Annotation]:
        """append float32 samples (16 kHz mono, the pipeline's rate); returns a provisional annotation when a refresh is
        due, else None"""
        if self.finished:
            raise RuntimeError("stream already finished")
        x = np.ascontiguousarray(np.asarray(samples, dtype=np.float32).reshape(-1))
        if self.n + len(x) + self.runner.window > self.capacity:
            raise MemoryError(f"stream longer than max_seconds = {(self.capacity - self.runner.window) / self.sr:.0f} s")
        if len(x) == 0:
            return None
        self._upload(x)
        self.n += len(x)
        self._compute(complete_windows(self.n, self.runner.window, self.runner.step))
        if (self.refresh_s is not None and self.done > 0
                and self.n - self.last_refresh_n >= self.refresh_s * self.sr):
            self.last_refresh_n = self.n
            self.stats["refreshes"] += 1
            return self._annotate()
        return None

    def _annotate(self) -> Annotation:
        seg, emb = np.concatenate(self.seg), np.concatenate(self.emb)
        return self.pipe.host_stage(seg, emb, self.sess_name)

    def finish(self) -> Annotation:
        """end of stream: the zero-padded tail window (if the reference would run one), then the final host stage"""
        r = self.runner
        n_full, has_last = window_plan(self.n, r.window, r.step)
        if self.n > 0:
            self._compute(n_full + int(has_last))
        self.finished = True
        if self.done == 0:
            return Annotation(uri=self.sess_name)
        ann = self._annotate()
        if self.pipe.rttm_out_dir is not None and self.sess_name is not None:
            import os
            with open(os.path.join(self.pipe.rttm_out_dir, self.sess_name + ".rttm"), "w") as f:
                f.write(ann.to_rttm())
        return ann

    @property
    def seconds(self) -> float:
        return self.n / self.sr


def stream(pipeline, chunks: Iterable, sess_name: Optional[str] = None, **kw) -> Iterator[Tuple[float, Annotation]]:
    """generator form: yields (seconds of audio received, provisional Annotation) at every refresh and finally
    (total seconds, final Annotation)"""
    sess = StreamingSession(pipeline, sess_name, **kw)
    for c in chunks:
        ann = sess.feed(c)
        if ann is not None:
            yield sess.seconds, ann
    yield sess.seconds, sess.finish()
