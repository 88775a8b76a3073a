"""Architecture tables of the segmentation / embedding models served by the engine.

The numbers are the shape contract of the reference checkpoints
(diarizen/models/module/wavlm_config.py:38-239: the four `get_config` names, including the
irregular structured-pruned "s80" encoders) expressed as one dataclass that maps 1:1 onto
`dzn_config` of include/dzn.h.  Head (EEND-Conformer) defaults follow
diarizen/models/eend/model_wavlm_conformer.py:26-45 and the recipe confs
(attention_in 256, ffn_hidden 1024, 4 heads, 4 layers, kernel 31).
"""
from __future__ import annotations

from dataclasses import dataclass, field, replace
from typing import List, Tuple

# (k, s) of the 7 feature-extractor convs: shared by every config
_KS = [(10, 5), (3, 2), (3, 2), (3, 2), (3, 2), (2, 2), (2, 2)]


@dataclass(frozen=True)
class SegConfig:
    name: str
    extractor_layer_norm: bool          # "layer_norm" (large) vs "group_norm" (base)
    normalize_waveform: bool
    conv_channels: Tuple[int, ...]
    embed_dim: int
    total_heads: int
    layer_norm_first: bool              # encoder_layer_norm_first
    remaining_heads: Tuple[Tuple[int, ...], ...]   # () => layer has no attention
    ffn_dims: Tuple[int, ...]
    conv_kernels: Tuple[int, ...] = tuple(k for k, _ in _KS)
    conv_strides: Tuple[int, ...] = tuple(s for _, s in _KS)
    pos_conv_kernel: int = 128
    pos_conv_groups: int = 16
    num_buckets: int = 320
    max_distance: int = 800
    # EEND-Conformer head
    attention_in: int = 256
    ffn_hidden: int = 1024
    conf_heads: int = 4
    conf_layers: int = 4
    conf_kernel: int = 31
    max_speakers_per_chunk: int = 4
    max_speakers_per_frame: int = 2
    sample_rate: int = 16000

    @property
    def n_layers(self) -> int:
        return len(self.ffn_dims)

    @property
    def use_attention(self) -> Tuple[bool, ...]:
        return tuple(len(h) > 0 for h in self.remaining_heads)

    @property
    def wavlm_layer_num(self) -> int:
        return self.n_layers + 1

    @property
    def n_classes(self) -> int:
        """number of powerset classes: sum_{i<=max_per_frame} C(n, i) (PA/utils/powerset.py:58-66)"""
        from math import comb
        return sum(comb(self.max_speakers_per_chunk, i)
                   for i in range(self.max_speakers_per_frame + 1))

    def num_frames(self, num_samples: int) -> int:
        """model_wavlm_conformer.py:98-124 / PA/utils/receptive_field.py:26-53 (no padding)"""
        n = num_samples
        for k, s in zip(self.conv_kernels, self.conv_strides):
            n = (n - k) // s + 1
        return n


def seg_config_from_wavlm_kwargs(kw: dict, name: str = "custom") -> SegConfig:
    """Build a SegConfig from the kwargs dict of `wav2vec2_model(**config)` as stored in a WavLM
    checkpoint's "config" entry (load_wavlm file branch, model_wavlm_conformer.py:209-221) or listed
    in diarizen/models/module/wavlm_config.py.  Pruning must be disabled (:216-218)."""
    for k, v in kw.items():
        if "prune" in k and v is not False:
            raise ValueError(f"Pruning must be disabled. Found: {k}={v}")
    if kw.get("extractor_conv_bias", False):
        raise ValueError("extractor_conv_bias=True is not supported")
    if not all(kw.get("encoder_use_feed_forward", [True])):
        raise ValueError("layers without feed-forward are not supported")
    convs = kw["extractor_conv_layer_config"]
    heads_total = kw["encoder_total_num_heads"]
    if len(set(heads_total)) != 1:
        raise ValueError("encoder_total_num_heads must be uniform")
    use_attn = kw["encoder_use_attention"]
    remaining = tuple(tuple(h) if u else () for h, u in zip(kw["encoder_remaining_heads"], use_attn))
    return SegConfig(
        name=name, extractor_layer_norm=kw["extractor_mode"] == "layer_norm",
        normalize_waveform=bool(kw.get("normalize_waveform", False)),
        conv_channels=tuple(c for c, _, _ in convs), conv_kernels=tuple(k for _, k, _ in convs),
        conv_strides=tuple(s for _, _, s in convs), embed_dim=kw["encoder_embed_dim"],
        total_heads=heads_total[0], layer_norm_first=bool(kw["encoder_layer_norm_first"]),
        remaining_heads=remaining, ffn_dims=tuple(kw["encoder_ff_interm_features"]),
        pos_conv_kernel=kw["encoder_pos_conv_kernel"], pos_conv_groups=kw["encoder_pos_conv_groups"],
        num_buckets=kw.get("encoder_num_buckets", 320), max_distance=kw.get("encoder_max_distance", 800))


def _dense(n_layers: int, heads: int) -> Tuple[Tuple[int, ...], ...]:
    return tuple(tuple(range(heads)) for _ in range(n_layers))


WAVLM_BASE = SegConfig(
    name="wavlm_base", extractor_layer_norm=False, normalize_waveform=False,
    conv_channels=(512,) * 7, embed_dim=768, total_heads=12, layer_norm_first=False,
    remaining_heads=_dense(12, 12), ffn_dims=(3072,) * 12)

WAVLM_LARGE = SegConfig(
    name="wavlm_large", extractor_layer_norm=True, normalize_waveform=True,
    conv_channels=(512,) * 7, embed_dim=1024, total_heads=16, layer_norm_first=True,
    remaining_heads=_dense(24, 16), ffn_dims=(4096,) * 24)

WAVLM_BASE_S80_MD = SegConfig(
    name="wavlm_base_s80_md", extractor_layer_norm=False, normalize_waveform=False,
    conv_channels=(90, 161, 173, 181, 351, 155, 137), embed_dim=768, total_heads=12,
    layer_norm_first=False,
    remaining_heads=((1, 6), (5, 7, 8), (0, 3, 9), (0, 1, 4, 8, 11), (6, 8), (0,),
                     (7, 8, 10, 11), (0, 1, 4, 8), (), (), (4, 7), (5,)),
    ffn_dims=(666, 660, 649, 1080, 237, 299, 437, 573, 53, 80, 211, 334))

WAVLM_LARGE_S80_MD = SegConfig(
    name="wavlm_large_s80_md", extractor_layer_norm=True, normalize_waveform=True,
    conv_channels=(512, 153, 224, 255, 302, 368, 211), embed_dim=1024, total_heads=16,
    layer_norm_first=True,
    remaining_heads=((1, 2, 4, 5, 6), (9, 10, 14), (0, 1, 2, 4, 5, 7), (1, 4, 7, 12, 13, 14),
                     (0, 2, 3, 4, 13), (1, 7, 13, 14, 15), (11, 13, 15), (2, 3, 4, 8, 15),
                     (2, 5, 6, 15), (), (0, 1), (1, 3, 5, 12), (), (4, 7, 11), (6, 9), (11,),
                     (), (), (14,), (5, 15), (0, 2, 8, 11, 13, 15),
                     (0, 1, 3, 4, 5, 6, 7, 10, 13), (0, 1, 3, 6, 7, 9, 10, 11, 12, 14),
                     (1, 2, 3, 4, 7, 13, 14, 15)),
    ffn_dims=(1092, 925, 759, 646, 745, 615, 684, 958, 286, 294, 406, 377, 463, 542, 298, 236,
              96, 104, 134, 211, 473, 1011, 1770, 1316))

# A deliberately tiny pruned encoder in the large-s80 style (layer_norm extractor, pre-norm,
# irregular channels / heads / ffn, one attention-less layer).  Not a reference checkpoint:
# used by CPU-speed parity tests so the oracle finishes in well under a second.
TINY_LN = SegConfig(
    name="tiny_ln", extractor_layer_norm=True, normalize_waveform=True,
    conv_channels=(48, 37, 40, 45, 52, 33, 43), embed_dim=256, total_heads=4,
    layer_norm_first=True, remaining_heads=((1, 2), (), (0, 3), (2,)),
    ffn_dims=(100, 77, 40, 129), pos_conv_kernel=16, pos_conv_groups=4,
    attention_in=128, ffn_hidden=192, conf_heads=2, conf_layers=2, conf_kernel=7)

# Same idea in the base-s80 style (group_norm extractor, post-norm encoder).
TINY_GN = replace(TINY_LN, name="tiny_gn", extractor_layer_norm=False, normalize_waveform=False,
                  layer_norm_first=False)

_CONFIGS = {c.name: c for c in (WAVLM_BASE, WAVLM_LARGE, WAVLM_BASE_S80_MD, WAVLM_LARGE_S80_MD,
                                TINY_LN, TINY_GN)}


def get_seg_config(name: str) -> SegConfig:
    key = name.lower()
    if key not in _CONFIGS:
        raise ValueError(f"Unknown config name '{name}'. Available options: {', '.join(_CONFIGS)}.")
    return _CONFIGS[key]


@dataclass(frozen=True)
class EmbConfig:
    """WeSpeaker ResNet34 (PA/models/embedding/wespeaker/__init__.py:207-233, resnet.py:390-398)."""
    name: str = "wespeaker_resnet34"
    num_mel_bins: int = 80
    frame_length_ms: int = 25
    frame_shift_ms: int = 10
    sample_rate: int = 16000
    m_channels: int = 32
    num_blocks: Tuple[int, ...] = (3, 4, 6, 3)
    embed_dim: int = 256

    @property
    def frame_length(self) -> int:
        return self.sample_rate * self.frame_length_ms // 1000

    @property
    def frame_shift(self) -> int:
        return self.sample_rate * self.frame_shift_ms // 1000

    def num_fbank_frames(self, num_samples: int) -> int:
        if num_samples < self.frame_length:
            return 0
        return 1 + (num_samples - self.frame_length) // self.frame_shift


RESNET34 = EmbConfig()
