"""Drop-in facades over the engine for the reference's two plugin points.

(1) `[model].path` plugin (PA/core/model.py:360-369 + diarizen/utils.py:79-134): the reference
    instantiates any importable class with `[model.args]` and hands it the checkpoint through
    `load_state_dict`.  `WavLMConformer` takes the same kwargs as
    diarizen/models/eend/model_wavlm_conformer.py:26-45 `Model.__init__`, so a hub `config.toml`
    works unchanged (the reference class path is aliased to this class by `instantiate`).
(2) `PretrainedSpeakerEmbedding` (PA/pipelines/speaker_verification.py:612-705): `SpeakerEmbedding`
    has the same properties (`sample_rate`, `dimension`, `metric`, `min_num_samples`) and the same
    `__call__(waveforms[B,1,N], masks[B,L]) -> np.ndarray[B,256]`.
Neither has a CPU path: `.to()` must receive a HIP device and the HIP extension must be built.
"""
from __future__ import annotations

import importlib
import os
from dataclasses import dataclass
from enum import Enum
from functools import cached_property
from typing import Any, Dict, Mapping, Optional

import numpy as np
import torch

from .configs import RESNET34, EmbConfig, SegConfig, get_seg_config
from .core import SlidingWindow
from .engine import Engine
from .postprocess import receptive_field

ALIASES = {
    "diarizen.models.eend.model_wavlm_conformer.Model": "diarizen_amd.models.WavLMConformer",
}


def instantiate(path: str, args: Optional[Dict[str, Any]] = None, initialize: bool = True):
    """Mirror of diarizen/utils.py:79-134 `instantiate`: "pkg.mod.Class" -> Class(**args)."""
    path = ALIASES.get(path, path)
    module_path, _, class_name = path.rpartition(".")
    cls = getattr(importlib.import_module(module_path), class_name)
    if not initialize:
        return cls
    return cls(**(args or {}))


class Resolution(Enum):      # PA/core/task.py Resolution
    FRAME = 1
    CHUNK = 2


@dataclass
class Specifications:
    """PA/core/task.py:80-136 `Specifications` as far as Inference / the pipelines read it (PA/core/inference.py:
    116-163, PA/core/model.py:159-170): iterable of itself (`for s in specifications`, `next(iter(...))`), frame
    resolution, powerset mono-label problem."""
    duration: float
    classes: tuple
    powerset_max_classes: int
    powerset: bool = True
    permutation_invariant: bool = True
    warm_up: tuple = (0.0, 0.0)
    resolution: Resolution = Resolution.FRAME
    min_duration: Optional[float] = None

    def __len__(self) -> int:
        return 1

    def __iter__(self):
        yield self

    @property
    def num_powerset_classes(self) -> int:
        from math import comb
        return sum(comb(len(self.classes), i) for i in range(self.powerset_max_classes + 1))


class WavLMConformer:
    """Segmentation model facade: forward(waveforms f32 [B, C, N]) -> log-probs f32 [B, L, 11]."""

    def __init__(self, wavlm_src: str = "wavlm_base", wavlm_layer_num: int = 13,
                 wavlm_feat_dim: int = 768, attention_in: int = 256, ffn_hidden: int = 1024,
                 num_head: int = 4, num_layer: int = 4, kernel_size: int = 31, dropout: float = 0.1,
                 use_posi: bool = False, output_activate_function=False,
                 max_speakers_per_chunk: int = 4, max_speakers_per_frame: int = 2,
                 chunk_size: int = 5, num_channels: int = 8, selected_channel: int = 0,
                 sample_rate: int = 16000, precision: str = "f32h", max_batch: int = 32):
        if use_posi or output_activate_function:
            raise NotImplementedError("use_posi / output activation are unused by the released confs")
        from dataclasses import replace
        if os.path.isfile(wavlm_src):
            # load_wavlm file branch (model_wavlm_conformer.py:209-221): {"config": ..., "state_dict": ...};
            # only the architecture is needed here, the weights arrive through load_state_dict()
            ckpt = torch.load(wavlm_src, map_location="cpu", weights_only=False)
            if "config" not in ckpt or "state_dict" not in ckpt:
                raise ValueError("Checkpoint must contain 'config' and 'state_dict'.")
            from .configs import seg_config_from_wavlm_kwargs
            cfg = seg_config_from_wavlm_kwargs(ckpt["config"], name=os.path.basename(wavlm_src))
        else:
            cfg = get_seg_config(wavlm_src)
        self.cfg: SegConfig = replace(cfg, attention_in=attention_in, ffn_hidden=ffn_hidden,
                                      conf_heads=num_head, conf_layers=num_layer,
                                      conf_kernel=kernel_size,
                                      max_speakers_per_chunk=max_speakers_per_chunk,
                                      max_speakers_per_frame=max_speakers_per_frame,
                                      sample_rate=sample_rate)
        assert wavlm_layer_num == self.cfg.wavlm_layer_num and wavlm_feat_dim == self.cfg.embed_dim
        self.chunk_size, self.sample_rate, self.selected_channel = chunk_size, sample_rate, selected_channel
        self.precision, self.max_batch = precision, max_batch
        self.specifications = Specifications(
            duration=chunk_size, classes=tuple(f"speaker#{i + 1}" for i in range(max_speakers_per_chunk)),
            powerset_max_classes=max_speakers_per_frame)
        from .compat import AudioLite
        self.audio = AudioLite(sample_rate, "downmix" if num_channels == 1 else None)    # PA/core/model.py:152-157
        self._state: Optional[Mapping[str, torch.Tensor]] = None
        self.engine: Optional[Engine] = None
        self.device = torch.device("cpu")

    # -- nn.Module-like surface used by Model.from_pretrained / Inference ---------------------------
    def load_state_dict(self, state_dict: Mapping[str, torch.Tensor], strict: bool = True):
        self._state = dict(state_dict)
        return self

    def eval(self):
        return self

    def to(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("diarizen_amd models run on a HIP device only (no CPU fallback)")
        self.device = device
        return self

    def bind(self, engine: Engine):
        """share an engine built elsewhere (the pipeline builds ONE engine for both models)"""
        self.engine, self.device = engine, engine.device
        return self

    def _ensure(self, num_samples: int):
        if self.engine is None:
            if self._state is None:
                raise RuntimeError("load_state_dict() must be called before forward()")
            self.engine = Engine(self.cfg, self._state, max_batch=self.max_batch,
                                 max_samples=max(num_samples, self.chunk_size * self.sample_rate),
                                 precision=self.precision, device=self.device)

    def num_frames(self, num_samples: int) -> int:
        return self.cfg.num_frames(num_samples)

    @cached_property
    def _receptive_field(self) -> SlidingWindow:
        return receptive_field(self.sample_rate)

    @property
    def dimension(self) -> int:
        return self.specifications.num_powerset_classes

    def forward(self, waveforms: torch.Tensor) -> torch.Tensor:
        assert waveforms.dim() == 3
        w = waveforms[:, self.selected_channel, :].to(self.device, torch.float32).contiguous()
        self._ensure(w.shape[1])
        logp, _ = self.engine.segment(w, want_logp=True, want_multilabel=False)
        return logp

    __call__ = forward


class SpeakerEmbedding:
    """WeSpeaker ResNet34 embedding facade (PyannoteAudioPretrainedSpeakerEmbedding surface)."""

    def __init__(self, state_dict: Optional[Mapping[str, torch.Tensor]] = None,
                 device: Optional[torch.device] = None, engine: Optional[Engine] = None,
                 cfg: EmbConfig = RESNET34, max_batch: int = 32, max_samples: int = 256000):
        self.cfg = cfg
        if engine is None:
            raise ValueError("SpeakerEmbedding needs engine= (the pipeline builds ONE engine that "
                             "holds both the segmentation and the embedding model)")
        self.engine = engine
        self.device = engine.device

    @property
    def sample_rate(self) -> int:
        return self.cfg.sample_rate

    @property
    def dimension(self) -> int:
        return self.cfg.embed_dim

    @property
    def metric(self) -> str:
        return "cosine"

    @cached_property
    def min_num_samples(self) -> int:
        """same bisection as speaker_verification.py:677-691: smallest length that does not raise"""
        lower, upper = 2, round(0.5 * self.sample_rate)
        middle = (lower + upper) // 2
        while lower + 1 < upper:
            try:
                self(torch.randn(1, 1, middle), None)
                upper = middle
            except Exception:
                lower = middle
            middle = (lower + upper) // 2
        return upper

    def __call__(self, waveforms: torch.Tensor, masks: Optional[torch.Tensor] = None) -> np.ndarray:
        w = waveforms[:, 0, :].to(self.device, torch.float32).contiguous()
        B, N = w.shape
        if masks is None:
            # un-weighted pooling == weights of ones (mean; unbiased std up to the 1e-8 guards)
            L = max(1, self.engine.num_frames(N)) if self.engine.seg is not None else 1
            m = torch.ones((B, 1, L), device=self.device)
        else:
            m = masks.to(self.device, torch.float32).reshape(B, 1, -1).contiguous()
        out = self.engine.embed(w, m)[:, 0]
        torch.cuda.synchronize(self.device)
        return out.cpu().numpy()
