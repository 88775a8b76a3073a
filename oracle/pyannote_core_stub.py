"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product (diarizen_amd/).

Stand-in for the subset of **pyannote.core 5.0.0** (third party; pinned in
pyannote-audio/requirements.txt; absent from this image and from /root/reference) that the REFERENCE's own
host-stage code touches when oracle/ref_host.py runs it by path:

    Inference.aggregate / trim                 PA/core/inference.py:544-714
    SpeakerDiarizationMixin.speaker_count      PA/pipelines/utils/diarization.py:121-157
    SpeakerDiarizationMixin.to_diarization     PA/pipelines/utils/diarization.py:192-239
    SpeakerDiarization.reconstruct             PA/pipelines/speaker_diarization.py:377-425
    Binarize.__call__                          PA/utils/signal.py:254-317

Restated from pyannote.core's published source (segment.py, feature.py, annotation.py), independently of
diarizen_amd/core.py (the product's stand-in).  **parity unpinned** at this boundary: no fixture of the reference
pins pyannote.core itself; what the goldens generated through this file DO pin is every loop, slice, rounding
and ordering decision of the five reference functions above.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterator, List, Optional, Tuple

import numpy as np

SEGMENT_PRECISION = 1e-6


@dataclass(frozen=True, order=True)
class Segment:
    """segment.py: ordered by (start, end); empty when end - start <= SEGMENT_PRECISION."""
    start: float = 0.0
    end: float = 0.0

    def __bool__(self):
        return bool((self.end - self.start) > SEGMENT_PRECISION)

    @property
    def duration(self) -> float:
        return self.end - self.start if self else 0.0

    @property
    def middle(self) -> float:
        return 0.5 * (self.start + self.end)

    def __and__(self, other: "Segment") -> "Segment":
        return Segment(start=max(self.start, other.start), end=min(self.end, other.end))


class SlidingWindow:
    """segment.py: window i = [start + i step, start + i step + duration)."""

    def __init__(self, duration=0.030, step=0.010, start=0.000, end=None):
        if duration <= 0:
            raise ValueError("'duration' must be a float > 0.")
        if step <= 0:
            raise ValueError("'step' must be a float > 0.")
        self.__duration, self.__step, self.__start = duration, step, start
        self.__end = np.inf if end is None else end
        self.__i = -1

    start = property(lambda self: self.__start)
    end = property(lambda self: self.__end)
    step = property(lambda self: self.__step)
    duration = property(lambda self: self.__duration)

    def closest_frame(self, t: float) -> int:
        return int(np.rint((t - self.__start - 0.5 * self.__duration) / self.__step))

    def range_to_segment(self, i0: int, n: int) -> Segment:
        start = self.__start + (i0 - 0.5) * self.__step + 0.5 * self.__duration
        end = start + n * self.__step
        if i0 == 0:
            start = self.start
        return Segment(start, end)

    def crop(self, focus: Segment, mode: str = "loose", fixed=None, return_ranges: bool = False):
        if not isinstance(focus, Segment) or fixed is not None or mode != "loose":
            raise NotImplementedError("stub: Segment focus, mode='loose', fixed=None only")
        i = int(np.ceil((focus.start - self.duration - self.start) / self.step))
        j = int(np.floor((focus.end - self.start) / self.step))
        rng = (i, j + 1)
        if return_ranges:
            return [list(rng)]
        return np.array(range(*rng), dtype=np.int64)

    def __getitem__(self, i: int) -> Optional[Segment]:
        start = self.__start + i * self.__step
        if start >= self.__end:
            return None
        return Segment(start=start, end=start + self.__duration)


class SlidingWindowFeature(np.lib.mixins.NDArrayOperatorsMixin):
    """feature.py: ndarray + the sliding window of its first axis; numpy ufuncs return the same wrapper."""
    _HANDLED_TYPES = (np.ndarray, int, float, np.number)

    def __init__(self, data: np.ndarray, sliding_window: SlidingWindow, labels: List[str] = None):
        self.sliding_window = sliding_window
        self.data = data
        self.labels = labels
        self.__i = -1

    def __len__(self):
        return self.data.shape[0]

    @property
    def extent(self) -> Segment:
        return self.sliding_window.range_to_segment(0, len(self))

    def __iter__(self) -> Iterator[Tuple[Segment, np.ndarray]]:
        self.__i = -1
        return self

    def __next__(self):
        self.__i += 1
        try:
            return self.sliding_window[self.__i], self.data[self.__i]
        except IndexError:
            raise StopIteration()

    def crop(self, focus: Segment, mode: str = "loose", fixed=None, return_data: bool = True):
        ranges = self.sliding_window.crop(focus, mode=mode, fixed=fixed, return_ranges=True)
        n_samples = self.data.shape[0]
        clipped = []
        for start, end in ranges:
            if end < 0 or start >= n_samples:
                continue
            clipped.append([max(start, 0), min(end, n_samples)])
        if clipped:
            data = np.vstack([self.data[s:e, :] for s, e in clipped])
        else:
            data = np.empty((0,) + self.data.shape[1:])
        if return_data:
            return data
        sw = SlidingWindow(start=self.sliding_window[clipped[0][0]].start, duration=self.sliding_window.duration,
                           step=self.sliding_window.step)
        return SlidingWindowFeature(data, sw, labels=self.labels)

    def __array__(self, dtype=None, copy=None) -> np.ndarray:
        return self.data if dtype is None else self.data.astype(dtype)

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        out = kwargs.get("out", ())
        for x in inputs + out:
            if not isinstance(x, self._HANDLED_TYPES + (SlidingWindowFeature,)):
                return NotImplemented
        inputs = tuple(x.data if isinstance(x, SlidingWindowFeature) else x for x in inputs)
        if out:
            kwargs["out"] = tuple(x.data if isinstance(x, SlidingWindowFeature) else x for x in out)
        data = getattr(ufunc, method)(*inputs, **kwargs)
        if type(data) is tuple:
            return tuple(type(self)(x, self.sliding_window) for x in data)
        elif method == "at":
            return None
        return type(self)(data, self.sliding_window)


class Timeline:            # imported by PA/utils/signal.py, not used on the Binarize path
    pass


class Annotation:
    """annotation.py: {segment: {track: label}} kept sorted by segment; RTTM writer of `_iter_rttm`."""

    def __init__(self, uri: Optional[str] = None, modality: Optional[str] = None):
        self.uri = uri
        self.modality = modality
        self._tracks: dict = {}

    def __setitem__(self, key, label):
        segment, track = key
        if not segment:                      # empty segments are silently ignored
            return
        self._tracks.setdefault(segment, {})[track] = label

    def __len__(self):
        return len(self._tracks)

    def itertracks(self, yield_label: bool = False):
        for segment in sorted(self._tracks):
            for track, label in sorted(self._tracks[segment].items(), key=lambda tl: (str(tl[0]), str(tl[1]))):
                yield (segment, track, label) if yield_label else (segment, track)

    def support(self, collar: float = 0.0):
        raise NotImplementedError("stub: Binarize is run with pad_* = min_duration_off = 0")

    def to_rttm(self) -> str:
        uri = self.uri if self.uri else "<NA>"
        if isinstance(uri, str) and " " in uri:
            raise ValueError("RTTM: URIs must not contain spaces")
        lines = []
        for segment, _, label in self.itertracks(yield_label=True):
            if isinstance(label, str) and " " in label:
                raise ValueError("RTTM: labels must not contain spaces")
            lines.append(f"SPEAKER {uri} 1 {segment.start:.3f} {segment.duration:.3f} <NA> <NA> {label} <NA> <NA>\n")
        return "".join(lines)
