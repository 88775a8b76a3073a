"""Make the engine acceptable to the REFERENCE's own `Inference` / pipeline objects.

`pyannote.audio.core.inference.Inference.__init__` (PA/core/inference.py:100-163) does
`model if isinstance(model, Model) else Model.from_pretrained(...)`, then reads `model.device`, calls
`model.eval()` / `model.to(device)`, iterates `model.specifications` (`.resolution`, `.duration`, `.warm_up`,
`.powerset`, `.classes`, `.powerset_max_classes`, `.permutation_invariant`; `isinstance(specifications,
Specifications)` inside `map_with_specifications`, PA/utils/multi_task.py:56-61); `slide` (:265-275) reads
`model.audio.get_num_samples(duration)` and `model._receptive_field`; `infer` (:215) calls `model(chunks)`.

So a plain class is not enough.  When `pyannote.audio` is importable — it is a dependency
of the reference, not of this package — `reference_model_class()` returns

    class WavLMConformerModel(pyannote.audio.core.model.Model)

whose constructor takes the kwargs of `diarizen.models.eend.model_wavlm_conformer.Model.__init__`
(model_wavlm_conformer.py:26-45), whose `specifications` / `audio` come from the reference base class itself
(PA/core/model.py:152-170) and whose `forward` is one `dzn_segment_forward` call.  `[model].path =
"diarizen_amd.compat.WavLMConformerModel"` in a hub `config.toml` therefore runs the reference pipeline unchanged
with the HIP engine underneath.  tests/test_host.py drives the reference's REAL `Inference.slide` through this class
(CPU, stub parents for the uninstalled third-party packages, a CPU stand-in for the engine).

Without pyannote.audio the duck-typed `diarizen_amd.models.WavLMConformer` is used by diarizen_amd's own pipeline;
its `Specifications` is iterable and carries `.resolution`, and it exposes `.audio`, so the same attribute reads work.
"""
from __future__ import annotations

from functools import cached_property
from typing import Mapping, Optional

import torch

from .models import WavLMConformer

_CLASS = None


def reference_model_class():
    """pyannote.audio.core.model.Model subclass over the engine (built on first use; ImportError without pyannote)."""
    global _CLASS
    if _CLASS is not None:
        return _CLASS
    from pyannote.audio.core.model import Model as RefModel   # the reference's base class (third-party parent)

    class WavLMConformerModel(RefModel):
        def __init__(self, wavlm_src: str = "wavlm_base", wavlm_layer_num: int = 13, wavlm_feat_dim: int = 768,
                     attention_in: int = 256, ffn_hidden: int = 1024, num_head: int = 4, num_layer: int = 4,
                     kernel_size: int = 31, dropout: float = 0.1, use_posi: bool = False,
                     output_activate_function=False, max_speakers_per_chunk: int = 4,
                     max_speakers_per_frame: int = 2, chunk_size: int = 5, num_channels: int = 8,
                     selected_channel: int = 0, sample_rate: int = 16000, precision: str = "f32h",
                     max_batch: int = 32):
            # model_wavlm_conformer.py:47-56: the base class builds `specifications`, `audio`, `powerset`
            super().__init__(num_channels=num_channels, duration=chunk_size,
                             max_speakers_per_chunk=max_speakers_per_chunk,
                             max_speakers_per_frame=max_speakers_per_frame)
            self._facade = WavLMConformer(
                wavlm_src=wavlm_src, wavlm_layer_num=wavlm_layer_num, wavlm_feat_dim=wavlm_feat_dim,
                attention_in=attention_in, ffn_hidden=ffn_hidden, num_head=num_head, num_layer=num_layer,
                kernel_size=kernel_size, dropout=dropout, use_posi=use_posi,
                output_activate_function=output_activate_function, max_speakers_per_chunk=max_speakers_per_chunk,
                max_speakers_per_frame=max_speakers_per_frame, chunk_size=chunk_size, num_channels=num_channels,
                selected_channel=selected_channel, sample_rate=sample_rate, precision=precision, max_batch=max_batch)
            self.chunk_size, self.selected_channel = chunk_size, selected_channel

        # --- what Model.from_pretrained / Inference call ------------------------------------------------
        def load_state_dict(self, state_dict: Mapping[str, torch.Tensor], strict: bool = True):
            self._facade.load_state_dict(state_dict, strict)
            return torch.nn.modules.module._IncompatibleKeys([], [])

        def state_dict(self, *a, **k):
            return dict(self._facade._state or {})

        def to(self, device=None, *a, **k):
            if device is not None and torch.device(device).type == "cuda":
                self._facade.to(device)
            return self

        @property
        def device(self) -> torch.device:
            return self._facade.device

        def bind(self, engine):
            self._facade.bind(engine)
            return self

        @property
        def dimension(self) -> int:
            return self._facade.dimension

        # --- receptive field (model_wavlm_conformer.py:98-190; PA/core/model.py:180-195 uses these) -----
        def num_frames(self, num_samples: int) -> int:
            return self._facade.num_frames(num_samples)

        def receptive_field_size(self, num_frames: int = 1) -> int:
            n = num_frames
            for k, s in zip(reversed(self._facade.cfg.conv_kernels), reversed(self._facade.cfg.conv_strides)):
                n = 1 + (k - 1) + (n - 1) * s
            return n

        def receptive_field_center(self, frame: int = 0) -> int:
            c = frame
            for k, s in zip(reversed(self._facade.cfg.conv_kernels), reversed(self._facade.cfg.conv_strides)):
                c = c * s + (k - 1) // 2
            return c

        def forward(self, waveforms: torch.Tensor, **kwargs) -> torch.Tensor:
            return self._facade(waveforms)

    _CLASS = WavLMConformerModel
    return _CLASS


def __getattr__(name: str):      # `diarizen_amd.compat.WavLMConformerModel` resolves lazily (needs pyannote.audio)
    if name == "WavLMConformerModel":
        return reference_model_class()
    raise AttributeError(name)


class AudioLite:
    """The two things Inference reads from `model.audio` on this path (PA/core/io.py: `Audio.get_num_samples`,
    `sample_rate`, `mono`) for the duck-typed facade."""

    def __init__(self, sample_rate: int = 16000, mono: Optional[str] = "downmix"):
        self.sample_rate, self.mono = sample_rate, mono

    def get_num_samples(self, duration: float, sample_rate: Optional[int] = None) -> int:
        import math
        return math.floor(duration * (sample_rate or self.sample_rate))       # PA/core/io.py:235-245
