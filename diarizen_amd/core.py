"""Minimal time-axis types used by the host side of the pipeline.

The reference returns `pyannote.core.Annotation` and passes `SlidingWindow(Feature)` objects
around (pyannote.core==5.0.0, a third-party dependency that is NOT installed in this image).
When pyannote.core is importable we use its classes, so the drop-in returns the very same
types; otherwise these small stand-ins provide the subset of behaviour the hot path relies on
(`closest_frame`, frame middles, `itertracks(yield_label=True)`, `to_rttm()`, `.uri`) with the
same arithmetic, so RTTM output is identical either way.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterator, List, Optional, Tuple

import numpy as np

try:  # pragma: no cover - not available offline
    from pyannote.core import Annotation, Segment, SlidingWindow, SlidingWindowFeature  # type: ignore
    HAVE_PYANNOTE_CORE = True
except Exception:  # ImportError and friends
    HAVE_PYANNOTE_CORE = False

    @dataclass(frozen=True, order=True)
    class Segment:
        start: float = 0.0
        end: float = 0.0

        @property
        def duration(self) -> float:
            return self.end - self.start if self.end > self.start else 0.0

        @property
        def middle(self) -> float:
            return 0.5 * (self.start + self.end)

    class SlidingWindow:
        """pyannote.core.SlidingWindow subset: frame i covers [start + i*step, start + i*step + duration)."""

        def __init__(self, duration: float = 0.030, step: float = 0.010, start: float = 0.0,
                     end: Optional[float] = None):
            self.duration, self.step, self.start = float(duration), float(step), float(start)
            self.end = end

        def closest_frame(self, t: float) -> int:
            return int(np.rint((t - self.start - 0.5 * self.duration) / self.step))

        def __getitem__(self, i: int) -> Segment:
            s = self.start + i * self.step
            return Segment(s, s + self.duration)

        def range_to_segment(self, i0: int, n: int) -> Segment:
            start = self.start + (i0 - 0.5) * self.step + 0.5 * self.duration
            end = start + n * self.step
            if i0 == 0:
                start = self.start
            return Segment(start, end)

    class SlidingWindowFeature:
        def __init__(self, data: np.ndarray, sliding_window: SlidingWindow, labels=None):
            self.data = data
            self.sliding_window = sliding_window
            self.labels = labels

        def __len__(self) -> int:
            return self.data.shape[0]

        def __iter__(self) -> Iterator[Tuple[Segment, np.ndarray]]:
            for i in range(len(self)):
                yield self.sliding_window[i], self.data[i]

        @property
        def extent(self) -> Segment:
            return self.sliding_window.range_to_segment(0, len(self))

    class Annotation:
        """pyannote.core.Annotation subset: (segment, track) -> label, iterated in segment order
        then by str(track); RTTM lines formatted exactly like pyannote.core's `_iter_rttm`."""

        def __init__(self, uri: Optional[str] = None, modality: Optional[str] = None):
            self.uri = uri
            self.modality = modality
            self._tracks: dict = {}

        def __setitem__(self, key, label) -> None:
            segment, track = key
            if segment.duration <= 0:      # pyannote ignores empty segments
                return
            self._tracks.setdefault(segment, {})[track] = label

        def __len__(self) -> int:
            return len(self._tracks)

        def __bool__(self) -> bool:
            return len(self._tracks) > 0

        def itertracks(self, yield_label: bool = False):
            for segment in sorted(self._tracks):
                tracks = self._tracks[segment]
                for track, label in sorted(tracks.items(), key=lambda tl: (str(tl[0]), str(tl[1]))):
                    yield (segment, track, label) if yield_label else (segment, track)

        def labels(self) -> List:
            return sorted({l for t in self._tracks.values() for l in t.values()}, key=str)

        def to_rttm(self) -> str:
            uri = self.uri if self.uri else "<NA>"
            return "".join(
                f"SPEAKER {uri} 1 {seg.start:.3f} {seg.duration:.3f} <NA> <NA> {label} <NA> <NA>\n"
                for seg, _, label in self.itertracks(yield_label=True))

        def __eq__(self, other) -> bool:
            return isinstance(other, Annotation) and list(self.itertracks(True)) == list(other.itertracks(True))
