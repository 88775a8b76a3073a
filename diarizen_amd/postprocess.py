"""Host post-processing between the device stages and the RTTM: overlap-add aggregation,
instantaneous speaker count, reconstruction of the global diarization and binarisation.

Restates, with numpy vector ops instead of per-frame Python loops ("f2" row of SURVEY.md §8f):
  * Inference.aggregate                      PA/core/inference.py:544-666
  * SpeakerDiarizationMixin.speaker_count    PA/pipelines/utils/diarization.py:121-157
  * SpeakerDiarizationMixin.to_diarization   PA/pipelines/utils/diarization.py:192-239
  * SpeakerDiarization.reconstruct           PA/pipelines/speaker_diarization.py:377-425
  * Binarize.__call__ (onset = offset = 0.5) PA/utils/signal.py:254-317
The frame grid is SlidingWindow(start = chunks.start, duration = 0.025, step = 0.02): the
receptive-field START is discarded by aggregate() (inference.py:577-581) — kept as is.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

from .core import HAVE_PYANNOTE_CORE, Annotation, Segment, SlidingWindow, SlidingWindowFeature


def receptive_field(sample_rate: int = 16000) -> SlidingWindow:
    """Model._receptive_field (PA/core/model.py:180-195) for the 7-conv WavLM extractor
    (k = 10,3,3,3,3,2,2 ; s = 5,2,2,2,2,2,2): size 400 samples, step 320, centre of frame 0 at
    sample 79 -> start = (79 - 199.5)/sr = -0.00753125 s."""
    k = [10, 3, 3, 3, 3, 2, 2]
    s = [5, 2, 2, 2, 2, 2, 2]

    def size(n):
        for kk, ss in zip(reversed(k), reversed(s)):
            n = 1 + (kk - 1) + (n - 1) * ss
        return n

    def center(frame):       # PA/utils/receptive_field.py: frame*stride + (k-1)//2, innermost conv last
        c = int(frame)
        for kk, ss in zip(reversed(k), reversed(s)):
            c = c * ss + (kk - 1) // 2
        return c

    sz = size(1)
    st = size(2) - sz
    start = center(0) - (sz - 1) / 2
    return SlidingWindow(start=start / sample_rate, duration=sz / sample_rate, step=st / sample_rate)


def aggregate(scores: np.ndarray, chunks: SlidingWindow, frames: SlidingWindow, *, hamming: bool = False,
              missing: float = np.nan, skip_average: bool = False, epsilon: float = 1e-12
              ) -> SlidingWindowFeature:
    """scores [C, L, K] (NaN = missing) with window c covering [chunks.start + c*step, +duration).
    warm_up is (0, 0) on this path (diarizen/pipelines/inference.py:138-142)."""
    C, L, K = scores.shape
    frames = SlidingWindow(start=chunks.start, duration=frames.duration, step=frames.step)
    mask = (~np.isnan(scores)).astype(np.float32)
    data = np.nan_to_num(scores, nan=0.0).astype(np.float32)
    win = (np.hamming(L) if hamming else np.ones(L)).reshape(-1, 1).astype(np.float64)
    num_frames = frames.closest_frame(chunks.start + chunks.duration + (C - 1) * chunks.step
                                      + 0.5 * frames.duration) + 1
    out = np.zeros((num_frames, K), dtype=np.float32)
    cnt = np.zeros((num_frames, K), dtype=np.float32)
    seen = np.zeros((num_frames, K), dtype=np.float32)
    for c in range(C):
        s0 = frames.closest_frame(chunks.start + c * chunks.step + 0.5 * frames.duration)
        out[s0:s0 + L] += data[c] * mask[c] * win
        cnt[s0:s0 + L] += mask[c] * win
        np.maximum(seen[s0:s0 + L], mask[c], out=seen[s0:s0 + L])
    avg = out if skip_average else out / np.maximum(cnt, epsilon)
    avg[seen == 0.0] = missing
    return SlidingWindowFeature(avg, frames)


def speaker_count(segmentations: np.ndarray, chunks: SlidingWindow, frames: SlidingWindow
                  ) -> SlidingWindowFeature:
    """[C, L, S] {0,1} -> (num_frames, 1) uint8 instantaneous speaker count."""
    tot = np.sum(segmentations.astype(np.float32), axis=-1, keepdims=True)
    count = aggregate(tot, chunks, frames, hamming=False, missing=0.0, skip_average=False)
    count.data = np.rint(count.data).astype(np.uint8)
    return count


def to_diarization(clustered: np.ndarray, chunks: SlidingWindow, count: SlidingWindowFeature
                   ) -> Tuple[SlidingWindowFeature, SlidingWindowFeature]:
    act = aggregate(clustered, chunks, count.sliding_window, hamming=False, missing=0.0,
                    skip_average=True)
    return _select_top_count(act, count)


def reconstruct(segmentations: np.ndarray, chunks: SlidingWindow, hard_clusters: np.ndarray,
                count: SlidingWindowFeature) -> Tuple[SlidingWindowFeature, SlidingWindowFeature]:
    """segmentations [C, L, S], hard_clusters [C, S] (-2 = inactive) -> discrete diarization."""
    C, L, S = segmentations.shape
    K = int(np.max(hard_clusters)) + 1 if hard_clusters.size else 0
    K = max(K, 0)
    clustered = np.full((C, L, K), np.nan, dtype=np.float64)
    seg = segmentations.astype(np.float64)
    for k in range(K):
        sel = hard_clusters == k                                   # [C, S]
        has = sel.any(axis=1)
        if not has.any():
            continue
        vals = np.where(sel[:, None, :], seg, -np.inf).max(axis=2)  # max over local speakers -> [C, L]
        clustered[has, :, k] = vals[has]
    return to_diarization(clustered, chunks, count)


# ----------------------------------------------------------------------------- device versions (row f2)
def _frame_grid(C: int, L: int, chunks: SlidingWindow, frames: SlidingWindow):
    """(frame grid of aggregate(), start frame of every window, number of output frames) — the reference's float64
    closest_frame arithmetic (PA/core/inference.py:577-581, 611-620, 645) stays on the host."""
    grid = SlidingWindow(start=chunks.start, duration=frames.duration, step=frames.step)
    starts = np.array([grid.closest_frame(chunks.start + c * chunks.step + 0.5 * grid.duration) for c in range(C)],
                      dtype=np.int32)
    T = grid.closest_frame(chunks.start + chunks.duration + (C - 1) * chunks.step + 0.5 * grid.duration) + 1
    return grid, starts, int(T)


_HOST_STREAMS = {}


def _host_stream(device):
    """one high-priority non-blocking stream per device for the host stage's aggregations (the C side keeps its own for the
    linkage / cdist calls, csrc/linkage.hip)"""
    import torch
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    st = _HOST_STREAMS.get(key)
    if st is None:
        st = _HOST_STREAMS[key] = torch.cuda.Stream(device=device, priority=-1)
    return st


class DevicePost:
    """speaker_count / reconstruct with their overlap-add aggregations on the HIP device (dzn_speaker_count,
    dzn_cluster_activations: integer atomics over the u8 decisions).  The decisions are uploaded once (3.6 MB per
    30 min); the top-`count` selection per frame keeps the reference's numpy call on the downloaded activations."""

    def __init__(self, segmentations: np.ndarray, chunks: SlidingWindow, frames: SlidingWindow, device):
        import ctypes as C_
        import torch
        from . import _lib
        self._C, self.torch, self.lib, self.check = C_, torch, _lib.load(), _lib.check
        self.device = torch.device(device)
        seg = np.ascontiguousarray(segmentations)
        if seg.dtype != np.uint8:
            if not np.array_equal(seg, seg.astype(np.uint8)):
                raise ValueError("DevicePost needs hard {0,1} decisions")
            seg = seg.astype(np.uint8)
        self.C, self.L, self.S = seg.shape
        self.chunks = chunks
        self.grid, starts, self.T = _frame_grid(self.C, self.L, chunks, frames)
        # (r5) the host stage's own stream: it may run in a second thread while the engine executes the next recording's
        # device stage (pipeline.diarize_many), and work queued on the default stream would wait behind the engine's
        self.stream = _host_stream(self.device)
        with torch.cuda.device(self.device), torch.cuda.stream(self.stream):
            self.seg = torch.from_numpy(seg).to(self.device)
            self.starts = torch.from_numpy(starts).to(self.device)

    def _p(self, t):
        return self._C.c_void_p(t.data_ptr())

    def speaker_count(self) -> SlidingWindowFeature:
        torch = self.torch
        with torch.cuda.device(self.device), torch.cuda.stream(self.stream):
            work = torch.empty(2 * self.T, device=self.device, dtype=torch.int32)
            out = torch.empty(self.T, device=self.device, dtype=torch.uint8)
            st = self._C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            self.check(self.lib.dzn_speaker_count(self._p(self.seg), self.C, self.L, self.S, self._p(self.starts), self.T,
                                                  self._p(work), self._p(out), st), None, "dzn_speaker_count")
            return SlidingWindowFeature(out.cpu().numpy().reshape(-1, 1), self.grid)

    def reconstruct(self, hard_clusters: np.ndarray, count: SlidingWindowFeature):
        torch = self.torch
        K = int(np.max(hard_clusters)) + 1 if hard_clusters.size else 0
        if K < 1 or K > 32:
            return None                                     # caller falls back to the numpy path
        with torch.cuda.device(self.device), torch.cuda.stream(self.stream):
            hard = torch.from_numpy(np.ascontiguousarray(hard_clusters, dtype=np.int8)).to(self.device)
            act = torch.empty((self.T, K), device=self.device, dtype=torch.int32)
            st = self._C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            self.check(self.lib.dzn_cluster_activations(self._p(self.seg), self._p(hard), self.C, self.L, self.S,
                                                        self._p(self.starts), self.T, K, self._p(act), st), None,
                       "dzn_cluster_activations")
            a = act.cpu().numpy().astype(np.float32)
        return _select_top_count(SlidingWindowFeature(a, count.sliding_window), count)


def _select_top_count(act: SlidingWindowFeature, count: SlidingWindowFeature):
    """tail of to_diarization (PA/pipelines/utils/diarization.py:222-239) on aggregated activations"""
    K = act.data.shape[1]
    max_per_frame = int(np.max(count.data)) if count.data.size else 0
    if K < max_per_frame:
        act.data = np.pad(act.data, ((0, 0), (0, max_per_frame - K)))
    n = min(len(act.data), len(count.data))      # identical grids: extent & extent keeps all frames
    a = act.data[:n]
    c = count.data[:n].reshape(-1).astype(np.int64)
    binary = _top_count_mask(a, c)
    sw = SlidingWindow(start=act.sliding_window.start, duration=act.sliding_window.duration,
                       step=act.sliding_window.step)
    return SlidingWindowFeature(binary, sw), SlidingWindowFeature(a, sw)


def _top_count_mask(a: np.ndarray, c: np.ndarray) -> np.ndarray:
    """binary[i, k] = 1 for the c[i] largest activations of frame i — the reference sorts every frame
    (`np.argsort(-activations, axis=-1)` + a loop, PA/pipelines/utils/diarization.py:228-236), which is 0.07 s (AVX-512 host)
    to 0.6 s (without) of the 4 h host stage.  The selected SET does not depend on the sort when the c-th and (c+1)-th
    largest values of a frame differ: it is `a >= (c-th largest)`.  Only frames with a tie AT that boundary take the
    reference's own call, so its (numpy-build-dependent) tie order is reproduced, not re-invented."""
    n, K = a.shape
    binary = np.zeros_like(a)
    maxc = int(c.max()) if n else 0
    if maxc <= 0 or K == 0:
        return binary
    if maxc >= K:                                   # (count is capped at the number of columns by the padding above)
        maxc = K
    top = np.partition(a, K - maxc, axis=1)[:, K - maxc:]        # the maxc largest of every frame, unordered
    top = -np.sort(-top, axis=1)                                  # descending: top[:, j] = (j + 1)-th largest
    cc = np.minimum(c, K)
    kth = np.where(cc > 0, top[np.arange(n), np.maximum(cc, 1) - 1], np.inf)
    ge = a >= kth[:, None]
    clean = ge.sum(axis=1) == cc                    # no tie at the boundary: the set is unique
    binary[ge & clean[:, None]] = 1
    tied = np.nonzero(~clean & (cc > 0))[0]
    if len(tied):
        at = a[tied]
        order = np.argsort(-at, axis=-1)            # same call as the reference (ties: numpy's order)
        sel = (np.arange(K)[None, :] < cc[tied][:, None]).astype(a.dtype)
        bt = np.zeros_like(at)
        np.put_along_axis(bt, order, sel, axis=-1)
        binary[tied] = bt
    return binary


def binarize(diar: SlidingWindowFeature, onset: float = 0.5, offset: Optional[float] = None,
             uri: Optional[str] = None) -> Annotation:
    """Binarize(onset=0.5, offset=0.5, min_duration_on=0, min_duration_off=0): regions run from the
    MIDDLE of the first active frame to the middle of the first inactive frame (or of the last frame)."""
    offset = onset if offset is None else offset
    data = diar.data
    n, K = data.shape
    fr = diar.sliding_window
    # frames[i].middle with pyannote.core's exact float64 op order (the .3f RTTM rounding of the
    # x.xxx5 timestamps depends on it): s = start + i*step ; e = s + duration ; middle = .5*(s + e)
    s_ = fr.start + np.arange(n) * fr.step
    ts = 0.5 * (s_ + (s_ + fr.duration))
    ann = Annotation(uri=uri)
    if n < 2:
        return ann
    fast = not HAVE_PYANNOTE_CORE and type(ann).__setitem__ is Annotation.__setitem__ and hasattr(ann, "_tracks")
    for k in range(K):
        y = data[:, k]
        # hysteresis state machine; with onset == offset on {0,1} data it is a plain threshold
        if onset == offset:
            on = y > onset
            act = on.copy()
            # frames exactly equal to the threshold keep the previous state
            eq = y == onset
            if eq.any():
                act = _hysteresis(y, onset, offset)
        else:
            act = _hysteresis(y, onset, offset)
        d = np.diff(act.astype(np.int8))
        starts = np.nonzero(d == 1)[0] + 1
        ends = np.nonzero(d == -1)[0] + 1
        if act[0]:
            starts = np.concatenate([[0], starts])
        if act[-1]:
            ends = np.concatenate([ends, [n - 1]])
        # 30 k turns at 4 h: the python objects are the cost (r6 profile: 75 ms); timestamps leave numpy in one tolist() each and
        # the stand-in Annotation (core.py) is filled through its dictionary - same entries as `ann[Segment(s, e), k] = k`
        t0, t1 = ts[starts].tolist(), ts[ends].tolist()
        if fast:
            tracks = ann._tracks
            for a, b in zip(t0, t1):
                if b > a:                     # pyannote ignores empty segments
                    tracks.setdefault(Segment(a, b), {})[k] = k
        else:
            for a, b in zip(t0, t1):
                ann[Segment(a, b), k] = k
    return ann


def _hysteresis(y: np.ndarray, onset: float, offset: float) -> np.ndarray:
    act = np.zeros(len(y), dtype=bool)
    state = y[0] > onset
    act[0] = state
    for i in range(1, len(y)):
        if state:
            if y[i] < offset:
                state = False
        elif y[i] > onset:
            state = True
        act[i] = state
    return act
