"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product (diarizen_amd/).

Runs the REFERENCE's OWN host-stage code, imported by path from /root/reference (build container only — the GPU
box has no reference tree; what travels are the fixtures oracle/gen_golden.py makes with this file):

    Inference.aggregate / Inference.trim        PA/core/inference.py:544-714
    SpeakerDiarizationMixin.speaker_count       PA/pipelines/utils/diarization.py:121-157
    SpeakerDiarizationMixin.to_diarization      PA/pipelines/utils/diarization.py:192-239
    SpeakerDiarization.reconstruct              PA/pipelines/speaker_diarization.py:377-425
    Binarize                                    PA/utils/signal.py:200-317

and strings them together exactly as `DiariZenPipeline.__call__` does (diarizen/pipelines/inference.py:137-185).
Nothing of those functions is restated here.  The third-party packages they import and this image lacks are
replaced by stand-ins: `pyannote.core` -> oracle/pyannote_core_stub.py (a restatement of 5.0.0: that boundary
stays **unpinned**), `pyannote.metrics` / `pyannote.pipeline` / `pytorch_lightning` / the `pyannote.audio`
package root / clustering / speaker_verification -> empty shells (imported by the modules, never reached by the
five functions).
"""
from __future__ import annotations

import importlib.util
import sys
import types
from pathlib import Path

import numpy as np

REF = Path("/root/reference")
PA = REF / "pyannote-audio" / "pyannote" / "audio"

_CACHE = {}


def available() -> bool:
    return PA.is_dir()


def load():
    """-> namespace(Inference, Mixin, SpeakerDiarization, Binarize, core): the reference classes + the pyannote.core stand-in."""
    if "ns" in _CACHE:
        return _CACHE["ns"]
    if not available():
        raise RuntimeError("/root/reference is not present: the reference host stage runs in the build container only")
    from oracle import pyannote_core_stub as core
    if not hasattr(np, "NaN"):
        np.NaN = np.nan                     # reference default arguments use the numpy < 2 spelling
    if not hasattr(np, "NAN"):
        np.NAN = np.nan
    saved = {k: v for k, v in sys.modules.items() if k == "pyannote" or k.startswith("pyannote.")
             or k.startswith("pytorch_lightning")}

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__path__ = []
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class _Shell:
        def __init__(self, *a, **k):
            pass

    def by_path(name, rel):
        spec = importlib.util.spec_from_file_location(name, PA / rel)
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        return m

    try:
        mod("pyannote")
        mod("pyannote.core", Segment=core.Segment, SlidingWindow=core.SlidingWindow,
            SlidingWindowFeature=core.SlidingWindowFeature, Annotation=core.Annotation, Timeline=core.Timeline)
        mod("pyannote.core.utils")
        mod("pyannote.core.utils.types", Label=object)
        mod("pyannote.core.utils.generators", pairwise=lambda it: zip(it[:-1], it[1:]))
        mod("pyannote.metrics")
        mod("pyannote.metrics.diarization", DiarizationErrorRate=_Shell, GreedyDiarizationErrorRate=_Shell)
        mod("pyannote.pipeline", Pipeline=_Shell)
        mod("pyannote.pipeline.parameter", ParamDict=_Shell, Uniform=_Shell, Categorical=_Shell, Integer=_Shell)
        mod("pytorch_lightning")
        mod("pytorch_lightning.utilities")
        mod("pytorch_lightning.utilities.memory", is_oom_error=lambda e: False)
        pa = mod("pyannote.audio", Audio=_Shell, Model=_Shell, Pipeline=_Shell)
        mod("pyannote.audio.core")
        mod("pyannote.audio.core.io", AudioFile=object)
        mod("pyannote.audio.core.model", Model=_Shell, Specifications=_Shell)
        mod("pyannote.audio.core.task", Resolution=_Shell, Specifications=_Shell, Problem=_Shell)
        mod("pyannote.audio.utils")
        mod("pyannote.audio.utils.reproducibility", fix_reproducibility=lambda *a, **k: None)
        by_path("pyannote.audio.utils.multi_task", "utils/multi_task.py")
        by_path("pyannote.audio.utils.powerset", "utils/powerset.py")
        inference = by_path("pyannote.audio.core.inference", "core/inference.py")
        pa.Inference = inference.Inference
        signal = by_path("pyannote.audio.utils.signal", "utils/signal.py")
        mod("pyannote.audio.pipelines")
        mod("pyannote.audio.pipelines.clustering", Clustering=_Shell)
        mod("pyannote.audio.pipelines.speaker_verification", PretrainedSpeakerEmbedding=_Shell)
        diar_utils = by_path("pyannote.audio.pipelines.utils.diarization", "pipelines/utils/diarization.py")
        mod("pyannote.audio.pipelines.utils", SpeakerDiarizationMixin=diar_utils.SpeakerDiarizationMixin,
            PipelineModel=object, get_model=None)
        sd = by_path("pyannote.audio.pipelines.speaker_diarization", "pipelines/speaker_diarization.py")
    finally:
        for k in [k for k in sys.modules if k == "pyannote" or k.startswith("pyannote.")
                  or k.startswith("pytorch_lightning")]:
            del sys.modules[k]
        sys.modules.update(saved)
    ns = types.SimpleNamespace(Inference=inference.Inference, Mixin=diar_utils.SpeakerDiarizationMixin,
                               SpeakerDiarization=sd.SpeakerDiarization, Binarize=signal.Binarize, core=core,
                               modules=types.SimpleNamespace(inference=inference, signal=signal, diarization=diar_utils,
                                                             speaker_diarization=sd))
    _CACHE["ns"] = ns
    return ns


RECEPTIVE_FIELD = dict(start=-0.00753125, duration=0.025, step=0.02)     # Model._receptive_field (PA/core/model.py:180-195)


def aggregate(scores: np.ndarray, chunk_start: float, chunk_duration: float, chunk_step: float, *, frame_duration=0.025,
              frame_step=0.02, frame_start=RECEPTIVE_FIELD["start"], **kw):
    """the reference's Inference.aggregate on a plain array -> (data, (start, duration, step) of the output frames)"""
    ns = load()
    swf = ns.core.SlidingWindowFeature(np.array(scores, copy=True),
                                       ns.core.SlidingWindow(start=chunk_start, duration=chunk_duration, step=chunk_step))
    frames = ns.core.SlidingWindow(start=frame_start, duration=frame_duration, step=frame_step)
    out = ns.Inference.aggregate(swf, frames, **kw)
    sw = out.sliding_window
    return out.data, (sw.start, sw.duration, sw.step)


def host_stage(seg: np.ndarray, hard_clusters: np.ndarray, duration: float, step_ratio: float, max_speakers: int, uri,
               return_parts: bool = False):
    """seg [C, L, S] median-filtered hard decisions, hard_clusters [C, S] as the clustering step returns them
    -> RTTM text, by the reference's own functions in the order of diarizen/pipelines/inference.py:137-185."""
    ns = load()
    core = ns.core
    chunks = core.SlidingWindow(start=0.0, duration=duration, step=step_ratio * duration)      # PA/core/inference.py:377-381
    segmentations = core.SlidingWindowFeature(seg.astype(np.float32), chunks)
    frames = core.SlidingWindow(**RECEPTIVE_FIELD)
    count = ns.Mixin.speaker_count(segmentations, frames, warm_up=(0.0, 0.0))                      # :137-142
    count.data = np.minimum(count.data, max_speakers).astype(np.int8)                               # :163
    inactive = np.sum(segmentations.data, axis=1) == 0                                              # :166
    hard = np.array(hard_clusters, copy=True)
    hard[inactive] = -2                                                                             # :170
    shell = types.SimpleNamespace(to_diarization=ns.Mixin.to_diarization)
    discrete, activations = ns.SpeakerDiarization.reconstruct(shell, segmentations, hard, count)   # :171-175
    result = ns.Binarize(onset=0.5, offset=0.5, min_duration_on=0.0, min_duration_off=0.0)(discrete)   # :178-184
    result.uri = uri
    rttm = result.to_rttm()
    if return_parts:
        sw = discrete.sliding_window
        return rttm, dict(count=count.data.copy(), binary=discrete.data.copy(), activations=np.asarray(activations.data).copy(),
                          frames=np.array([sw.start, sw.duration, sw.step]))
    return rttm
