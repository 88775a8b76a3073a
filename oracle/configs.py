"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product (diarizen_amd/).

The oracle's OWN architecture tables.  tests/golden/wavlm_configs.json is a verbatim dump of the reference's
`diarizen/models/module/wavlm_config.py:get_config(name)` dictionaries for its four names, written by
`oracle/gen_golden.py configs` (build container); this module turns them into the attribute view the oracle's functional
models read — so the oracle no longer takes its shapes from diarizen_amd/configs.py.
tests/test_oracle.py asserts, field by field, that the product's hand-written table equals this reference-derived one, and
(when /root/reference is present) that the JSON still equals the reference.  The two `tiny_*` encoders are not reference
checkpoints: they are defined here (and, independently, in the product) for CPU-speed parity tests.
EEND-Conformer head defaults: diarizen/models/eend/model_wavlm_conformer.py:26-45 and the recipe confs.
"""
from __future__ import annotations

import json
from math import comb
from pathlib import Path

_JSON = Path(__file__).resolve().parents[1] / "tests" / "golden" / "wavlm_configs.json"

HEAD_DEFAULTS = dict(attention_in=256, ffn_hidden=1024, conf_heads=4, conf_layers=4, conf_kernel=31,
                     max_speakers_per_chunk=4, max_speakers_per_frame=2, sample_rate=16000)


class OracleSegConfig:
    """attribute view of one `wav2vec2_model(**config)` dictionary + the head's hyper-parameters"""

    def __init__(self, name: str, kw: dict, **head):
        self.name = name
        self.kwargs = kw
        convs = kw["extractor_conv_layer_config"]
        self.extractor_layer_norm = kw["extractor_mode"] == "layer_norm"
        self.normalize_waveform = bool(kw.get("normalize_waveform", False))
        self.conv_channels = tuple(int(c) for c, _, _ in convs)
        self.conv_kernels = tuple(int(k) for _, k, _ in convs)
        self.conv_strides = tuple(int(s) for _, _, s in convs)
        self.embed_dim = int(kw["encoder_embed_dim"])
        heads = set(kw["encoder_total_num_heads"])
        assert len(heads) == 1
        self.total_heads = int(heads.pop())
        self.layer_norm_first = bool(kw["encoder_layer_norm_first"])
        self.remaining_heads = tuple(tuple(int(x) for x in h) if u else ()
                                     for h, u in zip(kw["encoder_remaining_heads"], kw["encoder_use_attention"]))
        self.ffn_dims = tuple(int(f) for f in kw["encoder_ff_interm_features"])
        self.pos_conv_kernel = int(kw["encoder_pos_conv_kernel"])
        self.pos_conv_groups = int(kw["encoder_pos_conv_groups"])
        self.num_buckets = int(kw.get("encoder_num_buckets", 320))
        self.max_distance = int(kw.get("encoder_max_distance", 800))
        for k, v in {**HEAD_DEFAULTS, **head}.items():
            setattr(self, k, v)

    FIELDS = ("name", "extractor_layer_norm", "normalize_waveform", "conv_channels", "conv_kernels", "conv_strides", "embed_dim",
              "total_heads", "layer_norm_first", "remaining_heads", "ffn_dims", "pos_conv_kernel", "pos_conv_groups",
              "num_buckets", "max_distance", *HEAD_DEFAULTS)

    @property
    def n_layers(self) -> int:
        return len(self.ffn_dims)

    @property
    def use_attention(self):
        return tuple(len(h) > 0 for h in self.remaining_heads)

    @property
    def wavlm_layer_num(self) -> int:
        return self.n_layers + 1

    @property
    def n_classes(self) -> int:          # PA/utils/powerset.py:58-66
        return sum(comb(self.max_speakers_per_chunk, i) for i in range(self.max_speakers_per_frame + 1))

    def num_frames(self, num_samples: int) -> int:      # model_wavlm_conformer.py:98-124 (no padding)
        n = num_samples
        for k, s in zip(self.conv_kernels, self.conv_strides):
            n = (n - k) // s + 1
        return n


def _tiny_kwargs(layer_norm: bool) -> dict:
    """a deliberately tiny pruned encoder in the large-s80 (layer_norm extractor, pre-norm) / base-s80 (group_norm, post-norm)
    style: irregular channels / heads / ffn, one attention-less layer"""
    ch = (48, 37, 40, 45, 52, 33, 43)
    ks = ((10, 5), (3, 2), (3, 2), (3, 2), (3, 2), (2, 2), (2, 2))
    heads = [[1, 2], [], [0, 3], [2]]
    return {"extractor_mode": "layer_norm" if layer_norm else "group_norm",
            "extractor_conv_layer_config": [(c, k, s) for c, (k, s) in zip(ch, ks)], "extractor_conv_bias": False,
            "encoder_embed_dim": 256, "encoder_pos_conv_kernel": 16, "encoder_pos_conv_groups": 4, "encoder_num_layers": 4,
            "encoder_use_attention": [len(h) > 0 for h in heads], "encoder_use_feed_forward": [True] * 4,
            "encoder_total_num_heads": [4] * 4, "encoder_remaining_heads": heads, "encoder_num_buckets": 320,
            "encoder_max_distance": 800, "encoder_ff_interm_features": [100, 77, 40, 129],
            "encoder_layer_norm_first": layer_norm, "normalize_waveform": layer_norm}


_TINY_HEAD = dict(attention_in=128, ffn_hidden=192, conf_heads=2, conf_layers=2, conf_kernel=7)
_CACHE = {}


def reference_tables() -> dict:
    if "json" not in _CACHE:
        _CACHE["json"] = json.loads(_JSON.read_text())
    return _CACHE["json"]


def get_seg_config(name: str) -> OracleSegConfig:
    key = name.lower()
    if key not in _CACHE:
        if key == "tiny_ln":
            _CACHE[key] = OracleSegConfig(key, _tiny_kwargs(True), **_TINY_HEAD)
        elif key == "tiny_gn":
            _CACHE[key] = OracleSegConfig(key, _tiny_kwargs(False), **_TINY_HEAD)
        else:
            tables = reference_tables()
            if key not in tables:
                raise ValueError(f"Unknown config name '{name}'. Available options: {', '.join(tables)}.")
            _CACHE[key] = OracleSegConfig(key, tables[key])
    return _CACHE[key]
