"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product (diarizen_amd/).

Loop-for-loop restatement of the HOST half of `DiariZenPipeline.__call__`
(diarizen/pipelines/inference.py:137-185) with the reference's own per-chunk / per-frame Python
loops kept as loops (the product's diarizen_amd/postprocess.py is a vectorised re-design; this file
shares no code with it, nor with diarizen_amd/core.py):
  * Inference.trim / Inference.aggregate     PA/core/inference.py:668-714, 544-666
  * speaker_count                            PA/pipelines/utils/diarization.py:121-157
  * to_diarization                           PA/pipelines/utils/diarization.py:192-239
  * reconstruct                              PA/pipelines/speaker_diarization.py:377-425
  * Binarize.__call__ + Annotation.to_rttm   PA/utils/signal.py:254-317, pyannote.core 5.0.0
Clustering is NOT restated here: the caller passes `hard_clusters` (oracle/gen_golden.py computes
them with the reference's own PA/pipelines/clustering.py).

pyannote.core (third party, pinned 5.0.0 in pyannote-audio/requirements.txt, absent from this image
and from /root/reference) supplies SlidingWindow.closest_frame / __getitem__ / range_to_segment and
the RTTM line format; they are restated below from its published source -> **parity unpinned** at
that boundary (no fixture of the reference pins them).
"""
from __future__ import annotations

import numpy as np


class _SW:
    """pyannote.core.SlidingWindow: frame i = [start + i*step, start + i*step + duration)."""

    def __init__(self, start=0.0, duration=0.03, step=0.01):
        self.start, self.duration, self.step = float(start), float(duration), float(step)

    def closest_frame(self, t):                     # int(np.rint((t - start - .5*duration) / step))
        return int(np.rint((t - self.start - 0.5 * self.duration) / self.step))

    def frame_start(self, i):
        return self.start + i * self.step

    def frame_middle(self, i):                      # Segment.middle = .5 * (start + end)
        s = self.start + i * self.step
        return 0.5 * (s + (s + self.duration))


RECEPTIVE_FIELD = (-0.00753125, 0.025, 0.02)          # Model._receptive_field: start, duration, step (PA/core/model.py:180-195)


def aggregate(scores, chunks: _SW, frames: _SW, hamming=False, missing=np.nan, skip_average=False,
              warm_up=(0.0, 0.0), epsilon=1e-12):
    """PA/core/inference.py:574-666."""
    num_chunks, num_frames_per_chunk, num_classes = scores.shape
    frames = _SW(start=chunks.start, duration=frames.duration, step=frames.step)      # :577-581
    masks = 1 - np.isnan(scores)
    scores = np.nan_to_num(scores, copy=True, nan=0.0)
    hamming_window = (np.hamming(num_frames_per_chunk).reshape(-1, 1) if hamming
                      else np.ones((num_frames_per_chunk, 1)))
    warm_up_window = np.ones((num_frames_per_chunk, 1))
    warm_up_left = round(warm_up[0] / chunks.duration * num_frames_per_chunk)
    warm_up_window[:warm_up_left] = epsilon
    warm_up_right = round(warm_up[1] / chunks.duration * num_frames_per_chunk)
    warm_up_window[num_frames_per_chunk - warm_up_right:] = epsilon
    num_frames = frames.closest_frame(chunks.start + chunks.duration + (num_chunks - 1) * chunks.step
                                      + 0.5 * frames.duration) + 1
    aggregated_output = np.zeros((num_frames, num_classes), dtype=np.float32)
    overlapping_chunk_count = np.zeros((num_frames, num_classes), dtype=np.float32)
    aggregated_mask = np.zeros((num_frames, num_classes), dtype=np.float32)
    for c in range(num_chunks):
        score, mask = scores[c], masks[c]
        start_frame = frames.closest_frame(chunks.frame_start(c) + 0.5 * frames.duration)
        sl = slice(start_frame, start_frame + num_frames_per_chunk)
        n = aggregated_output[sl].shape[0]           # numpy slice-assignment clips at the end of the array
        aggregated_output[sl] += (score * mask * hamming_window * warm_up_window)[:n]
        overlapping_chunk_count[sl] += (mask * hamming_window * warm_up_window)[:n]
        aggregated_mask[sl] = np.maximum(aggregated_mask[sl], mask[:n])
    if skip_average:
        average = aggregated_output
    else:
        average = aggregated_output / np.maximum(overlapping_chunk_count, epsilon)
    average[aggregated_mask == 0.0] = missing
    return average, frames


def speaker_count(seg, chunks: _SW, frames: _SW, warm_up=(0.0, 0.0)):
    """PA/pipelines/utils/diarization.py:147-157 with Inference.trim (:668-714)."""
    _, num_frames, _ = seg.shape
    left = round(num_frames * warm_up[0])
    right = round(num_frames * warm_up[1])
    trimmed = seg[:, left:num_frames - right]
    tchunks = _SW(start=chunks.start + warm_up[0] * chunks.duration, step=chunks.step,
                  duration=(1 - warm_up[0] - warm_up[1]) * chunks.duration)
    count, fr = aggregate(np.sum(trimmed, axis=-1, keepdims=True), tchunks, frames, hamming=False,
                          missing=0.0, skip_average=False)
    return np.rint(count).astype(np.uint8), fr


def to_diarization(clustered, chunks: _SW, count, count_frames: _SW):
    """PA/pipelines/utils/diarization.py:213-239.  `activations` and `count` share one frame grid and
    one length here, so `extent & extent` + crop(mode="loose") keep every frame."""
    activations, _ = aggregate(clustered, chunks, count_frames, hamming=False, missing=0.0, skip_average=True)
    _, num_speakers = activations.shape
    max_speakers_per_frame = int(np.max(count))
    if num_speakers < max_speakers_per_frame:
        activations = np.pad(activations, ((0, 0), (0, max_speakers_per_frame - num_speakers)))
    n = min(len(activations), len(count))
    activations, count = activations[:n], count[:n]
    sorted_speakers = np.argsort(-activations, axis=-1)
    binary = np.zeros_like(activations)
    for t in range(n):
        for i in range(int(count[t, 0])):
            binary[t, sorted_speakers[t, i]] = 1.0
    return binary


def reconstruct(seg, chunks: _SW, hard_clusters, count, count_frames: _SW):
    """PA/pipelines/speaker_diarization.py:400-425."""
    num_chunks, num_frames, _ = seg.shape
    num_clusters = int(np.max(hard_clusters)) + 1
    clustered = np.nan * np.zeros((num_chunks, num_frames, num_clusters))
    for c in range(num_chunks):
        cluster, segmentation = hard_clusters[c], seg[c]
        for k in np.unique(cluster):
            if k == -2:
                continue
            clustered[c, :, k] = np.max(segmentation[:, cluster == k], axis=1)
    return to_diarization(clustered, chunks, count, count_frames)


def binarize_to_rttm(binary, frames: _SW, uri, onset=0.5, offset=0.5):
    """PA/utils/signal.py:268-304 (no padding / min durations) then pyannote.core's RTTM writer:
    tracks iterate in (segment, track) order, empty segments are dropped, line format
    'SPEAKER {uri} 1 {start:.3f} {duration:.3f} <NA> <NA> {label} <NA> <NA>'."""
    num_frames, num_classes = binary.shape
    timestamps = [frames.frame_middle(i) for i in range(num_frames)]
    regions = []                                            # (start, end, track k, label k)
    for k in range(num_classes):
        k_scores = binary[:, k]
        start = timestamps[0]
        is_active = k_scores[0] > onset
        t = timestamps[0]
        for t, y in zip(timestamps[1:], k_scores[1:]):
            if is_active:
                if y < offset:
                    regions.append((start, t, k))
                    start = t
                    is_active = False
            else:
                if y > onset:
                    start = t
                    is_active = True
        if is_active:
            regions.append((start, t, k))
    regions = [r for r in regions if r[1] - r[0] > 0]       # Annotation ignores empty segments
    regions.sort(key=lambda r: (r[0], r[1], str(r[2])))
    name = uri if uri else "<NA>"
    return "".join(f"SPEAKER {name} 1 {s:.3f} {e - s:.3f} <NA> <NA> {k} <NA> <NA>\n" for s, e, k in regions)


def host_stage(seg_u8, hard_clusters, duration, step_ratio, max_speakers, uri):
    """seg_u8 [C, L, S] median-filtered hard decisions, hard_clusters [C, S] from the clustering step
    (BEFORE the inactive -> -2 marking) -> RTTM text.  diarizen/pipelines/inference.py:137-185."""
    seg = seg_u8.astype(np.float32)
    chunks = _SW(start=0.0, duration=duration, step=step_ratio * duration)     # PA/core/inference.py:377-381
    frames = _SW(*[RECEPTIVE_FIELD[i] for i in (0, 1, 2)])
    count, count_frames = speaker_count(seg, chunks, frames, warm_up=(0.0, 0.0))
    count = np.minimum(count, max_speakers).astype(np.int8)
    inactive = np.sum(seg, axis=1) == 0
    hard = np.array(hard_clusters, copy=True)
    hard[inactive] = -2
    binary = reconstruct(seg, chunks, hard, count, count_frames)
    return binarize_to_rttm(binary, count_frames, uri)
