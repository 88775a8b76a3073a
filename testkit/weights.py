"""Seeded random-initialised state_dicts with the reference checkpoints' exact key names/shapes.

No pretrained weights exist offline (BUT-FIT/diarizen-wavlm-*-s80-md, pyannote/wespeaker-voxceleb-
resnet34-LM), so benchmarks, smoke tests and parity fixtures use these.  Keys follow the reference
Model state_dict so the same ingest path serves real checkpoints.
"""
from __future__ import annotations

import math
from typing import Dict

import torch

SegConfig = object      # any object with the attributes of diarizen_amd.configs.SegConfig / oracle.configs.OracleSegConfig

P = "wavlm_model."


def seg_state_dict(cfg: SegConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded random state_dict with the exact keys/shapes of the reference Model
    (key list: SURVEY.md §8c).  BatchNorm running stats are randomised so that BN folding
    is exercised; LayerNorm affine params are non-trivial."""
    g = torch.Generator().manual_seed(seed)

    def rn(*shape, scale=1.0):
        return torch.randn(*shape, generator=g) * scale

    def lin(prefix, out_f, in_f, sd, bias=True, gain=1.0):
        sd[prefix + ".weight"] = rn(out_f, in_f, scale=gain / math.sqrt(in_f))
        if bias:
            sd[prefix + ".bias"] = rn(out_f, scale=0.1)

    def ln(prefix, n, sd):
        sd[prefix + ".weight"] = 1.0 + rn(n, scale=0.1)
        sd[prefix + ".bias"] = rn(n, scale=0.1)

    sd: Dict[str, torch.Tensor] = {}
    cin = 1
    for i, (c, k) in enumerate(zip(cfg.conv_channels, cfg.conv_kernels)):
        pre = f"{P}feature_extractor.conv_layers.{i}"
        sd[pre + ".conv.weight"] = rn(c, cin, k, scale=1.6 / math.sqrt(cin * k))
        if cfg.extractor_layer_norm or i == 0:
            ln(pre + ".layer_norm", c, sd)
        cin = c
    sd[P + "feature_extractor.dummy_weight"] = 1.0 + rn(cin, scale=0.05)
    D = cfg.embed_dim
    ln(P + "encoder.feature_projection.layer_norm", cin, sd)
    lin(P + "encoder.feature_projection.projection", D, cin, sd)
    cg = D // cfg.pos_conv_groups
    pc = P + "encoder.transformer.pos_conv_embed.conv"
    sd[pc + ".bias"] = rn(D, scale=0.1)
    sd[pc + ".parametrizations.weight.original0"] = 1.0 + 0.2 * torch.rand(1, 1, cfg.pos_conv_kernel, generator=g)
    sd[pc + ".parametrizations.weight.original1"] = rn(D, cg, cfg.pos_conv_kernel, scale=1.0)
    ln(P + "encoder.transformer.layer_norm", D, sd)
    for i in range(cfg.n_layers):
        lp = f"{P}encoder.transformer.layers.{i}"
        heads = cfg.remaining_heads[i]
        if heads:
            hd = len(heads) * 64
            for nm in ("q_proj", "k_proj", "v_proj"):
                lin(f"{lp}.attention.{nm}", hd, D, sd, gain=1.5)
            lin(f"{lp}.attention.out_proj", D, hd, sd, gain=0.7)
            lin(f"{lp}.attention.gru_rel_pos_linear", 8, 64, sd)
            sd[f"{lp}.attention.gru_rel_pos_const"] = 1.0 + rn(1, cfg.total_heads, 1, 1, scale=0.3)
            if i == 0:
                sd[f"{lp}.attention.rel_attn_embed.weight"] = rn(cfg.num_buckets, cfg.total_heads)
        ln(f"{lp}.layer_norm", D, sd)
        lin(f"{lp}.feed_forward.intermediate_dense", cfg.ffn_dims[i], D, sd)
        lin(f"{lp}.feed_forward.output_dense", D, cfg.ffn_dims[i], sd, gain=0.7)
        ln(f"{lp}.final_layer_norm", D, sd)
    # head
    sd["weight_sum.weight"] = rn(1, cfg.wavlm_layer_num, scale=1.0 / cfg.wavlm_layer_num) + 1.0 / cfg.wavlm_layer_num
    A = cfg.attention_in
    lin("proj", A, D, sd)
    ln("lnorm", A, sd)
    for i in range(cfg.conf_layers):
        cp = f"conformer.conformer_layer.{i}"
        for f_ in ("ffn1", "ffn2"):
            ln(f"{cp}.{f_}.ln_norm", A, sd)
            lin(f"{cp}.{f_}.w_1", cfg.ffn_hidden, A, sd)
            lin(f"{cp}.{f_}.w_2", A, cfg.ffn_hidden, sd)
        ln(f"{cp}.mha.ln_norm", A, sd)
        for nm in ("linearQ", "linearK", "linearV", "linearO"):
            lin(f"{cp}.mha.mha.{nm}", A, A, sd, gain=1.3)
        ln(f"{cp}.conv.ln_norm", A, sd)
        sd[f"{cp}.conv.pointwise_conv1.weight"] = rn(2 * A, A, 1, scale=1.0 / math.sqrt(A))
        sd[f"{cp}.conv.pointwise_conv1.bias"] = rn(2 * A, scale=0.1)
        sd[f"{cp}.conv.depthwise_conv.weight"] = rn(A, 1, cfg.conf_kernel, scale=1.0 / math.sqrt(cfg.conf_kernel))
        sd[f"{cp}.conv.depthwise_conv.bias"] = rn(A, scale=0.1)
        sd[f"{cp}.conv.bn_norm.weight"] = 1.0 + rn(A, scale=0.1)
        sd[f"{cp}.conv.bn_norm.bias"] = rn(A, scale=0.1)
        sd[f"{cp}.conv.bn_norm.running_mean"] = rn(A, scale=0.1)
        sd[f"{cp}.conv.bn_norm.running_var"] = 0.5 + torch.rand(A, generator=g)
        sd[f"{cp}.conv.bn_norm.num_batches_tracked"] = torch.tensor(7, dtype=torch.long)
        sd[f"{cp}.conv.pointwise_conv2.weight"] = rn(A, A, 1, scale=1.0 / math.sqrt(A))
        sd[f"{cp}.conv.pointwise_conv2.bias"] = rn(A, scale=0.1)
        ln(f"{cp}.ln_norm", A, sd)
    lin("classifier", cfg.n_classes, A, sd, gain=3.0)
    return sd


def emb_state_dict(seed: int = 0, m: int = 32, feat_dim: int = 80, embed_dim: int = 256,
                   num_blocks=(3, 4, 6, 3)) -> Dict[str, torch.Tensor]:
    """Seeded random WeSpeaker-ResNet34 state_dict (keys of resnet.py:ResNet/BasicBlock)."""
    g = torch.Generator().manual_seed(1000 + seed)

    def rn(*shape, scale=1.0):
        return torch.randn(*shape, generator=g) * scale

    sd: Dict[str, torch.Tensor] = {}

    def conv(name, co, ci, k):
        sd[name + ".weight"] = rn(co, ci, k, k, scale=math.sqrt(2.0 / (ci * k * k)))

    def bn(name, c):
        sd[name + ".weight"] = 1.0 + rn(c, scale=0.1)
        sd[name + ".bias"] = rn(c, scale=0.1)
        sd[name + ".running_mean"] = rn(c, scale=0.1)
        sd[name + ".running_var"] = 0.5 + torch.rand(c, generator=g)
        sd[name + ".num_batches_tracked"] = torch.tensor(3, dtype=torch.long)

    conv("resnet.conv1", m, 1, 3)
    bn("resnet.bn1", m)
    cin = m
    for s, nb in enumerate(num_blocks):
        cout = m << s
        for j in range(nb):
            p = f"resnet.layer{s + 1}.{j}"
            stride = 2 if (j == 0 and s > 0) else 1
            conv(p + ".conv1", cout, cin, 3)
            bn(p + ".bn1", cout)
            conv(p + ".conv2", cout, cout, 3)
            bn(p + ".bn2", cout)
            if stride != 1 or cin != cout:
                conv(p + ".shortcut.0", cout, cin, 1)
                bn(p + ".shortcut.1", cout)
            cin = cout
    stats_dim = (feat_dim // 8) * m * 8
    sd["resnet.seg_1.weight"] = rn(embed_dim, stats_dim * 2, scale=1.0 / math.sqrt(stats_dim * 2))
    sd["resnet.seg_1.bias"] = rn(embed_dim, scale=0.1)
    return sd


# ---------------------------------------------------------------------------- non-degenerate decisions
# A randomly initialised EEND head emits ONE powerset class for every frame (its features are a large
# constant plus ~5 % of frame-level noise), so hard decisions, the median filter, the overlap-excluded
# masks and the clustering would only ever be compared on constants.  `turn_taking_state_dict` returns
# seeded weights whose decisions look like turn taking on real audio:
#   * Conformer: the depthwise conv of every block becomes a positive Hann low-pass (per-channel
#     seeded gain) whose branch dominates the residual (pointwise_conv2 x 10) while the half-FFN / MHSA
#     branches are damped (x 0.1) -> the head features vary smoothly (lag-5 autocorrelation ~0.9);
#   * classifier: a calibration fitted ONCE by oracle/calibrate.py (within-window PCA of the oracle's head
#     features on tests/golden/EN2002a_30s.wav) and stored in testkit/data/cal_<config>.npz, so that
#     >= 6 powerset classes each take >= 5 % of the frames with ~15 transitions per 8 s window.
# Same keys / shapes as the reference checkpoint; everything else is `seg_state_dict(cfg, seed)`.
CAL_DIR = __import__("pathlib").Path(__file__).resolve().parent / "data"


def turn_taking_head(sd: Dict[str, torch.Tensor], cfg: SegConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(7000 + seed)
    sd = dict(sd)
    A, ks = cfg.attention_in, cfg.conf_kernel
    hann = torch.hann_window(ks + 2, periodic=False)[1:-1]
    hann = hann / hann.sum()
    for i in range(cfg.conf_layers):
        cp = f"conformer.conformer_layer.{i}"
        for nm in ("ffn1.w_2", "ffn2.w_2", "mha.mha.linearO"):
            sd[f"{cp}.{nm}.weight"] = sd[f"{cp}.{nm}.weight"] * 0.1
            sd[f"{cp}.{nm}.bias"] = sd[f"{cp}.{nm}.bias"] * 0.1
        gain = 4.0 * (1.0 + 0.2 * torch.randn(A, 1, 1, generator=g))
        sd[f"{cp}.conv.depthwise_conv.weight"] = (hann[None, None, :] * gain).contiguous()
        sd[f"{cp}.conv.pointwise_conv2.weight"] = sd[f"{cp}.conv.pointwise_conv2.weight"] * 10.0
    return sd


def turn_taking_state_dict(cfg: SegConfig, seed: int = 0, calibration=None) -> Dict[str, torch.Tensor]:
    """`calibration`: path of a cal_*.npz or a dict with W [n_classes, A], b [n_classes]; default = the
    file shipped for (cfg.name, seed)."""
    import numpy as np
    sd = turn_taking_head(seg_state_dict(cfg, seed), cfg, seed)
    if calibration is None:
        calibration = CAL_DIR / f"cal_{cfg.name}_seed{seed}.npz"
    if not isinstance(calibration, dict):
        if not __import__("os").path.exists(calibration):
            raise FileNotFoundError(f"{calibration}: run `python oracle/calibrate.py` (build container) first")
        calibration = dict(np.load(calibration))
    W = torch.as_tensor(np.asarray(calibration["W"], dtype=np.float32))
    b = torch.as_tensor(np.asarray(calibration["b"], dtype=np.float32))
    assert W.shape == sd["classifier.weight"].shape and b.shape == sd["classifier.bias"].shape
    sd["classifier.weight"], sd["classifier.bias"] = W.contiguous(), b.contiguous()
    return sd
