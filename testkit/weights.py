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


# ---------------------------------------------------------------------------- planted massive activations
# Trained WavLM / wav2vec 2.0 encoders carry "massive activations": a handful of residual-stream channels sit 10^3 - 10^4 x above
# the typical magnitude from an early layer onwards, almost constant over time, and every LayerNorm / linear layer downstream has
# learned to live with them (small gamma / small weight columns on those channels).  Seeded Gaussian weights have nothing of the kind,
# so every |max| tracker, power-of-two operand scale, fp8 block scale and LayerNorm statistic of the engine would only ever meet
# well-conditioned rows.  `outlier_state_dict` plants them (VERDICT r5 "next round" item 1):
#   * OUTLIER_CHANNELS residual channels become massive at the END of encoder layer 2 (1 for the tiny encoders): magnitudes 2^10 ... 2^13 x the
#     typical std (mixed signs), plus a token-varying part 2^6 x typical -
#       pre-norm encoders (large; W2V/components.py:920-935): bias + weight row of that layer's `feed_forward.output_dense`
#       (:805-813) - the residual stream is never normalised, so the channels stay massive to the last layer and in the layer sum;
#       post-norm encoders (base; :937-942): beta / gamma of that layer's `final_layer_norm`, and every later LayerNorm of the stream
#       re-emits them (gamma = the row rms, which an outlier-dominated row normalises to +-rms^-1 * M).
#   * every consumer is "trained" for them: LayerNorms that read the stream scale typical channels back to O(1) (gamma x rms / sigma)
#     and the massive ones down (pre-norm) ; linear layers that read the stream directly (post-norm q/k/v, FFN-in; `proj` after the
#     layer sum) get columns / M on those channels.  The relative-position gate (`gru_rel_pos_linear`, :702-710) is left alone: its
#     sigmoid saturates on the head that owns a massive channel, which is what a real checkpoint does to it too.
# The rest is `turn_taking_state_dict` with its own classifier calibration (testkit/data/cal_<config>_outlier_seed<seed>.npz).
OUTLIER_CHANNELS = 4
OUTLIER_MAGS = (2.0 ** 10, -2.0 ** 11, -1.25 * 2.0 ** 12, 2.0 ** 13)     # x sigma; their sum (2^11) shifts every row mean by 2-3 sigma


def outlier_layer(cfg: SegConfig) -> int:
    return min(2, cfg.n_layers // 2 - 1)


def outlier_plan(cfg: SegConfig, seed: int = 0):
    """-> (channels [4], signed magnitudes [4], rms of an outlier-dominated row, typical sigma of the stream)"""
    g = torch.Generator().manual_seed(9000 + seed)
    D = cfg.embed_dim
    chans = torch.randperm(D, generator=g)[:OUTLIER_CHANNELS]
    sigma = 2.0 if cfg.layer_norm_first else 1.0
    mags = sigma * torch.tensor(OUTLIER_MAGS)
    rms = float(torch.sqrt((mags ** 2).sum() / D))
    return chans, mags, rms, sigma


def outlier_ln_sites(cfg: SegConfig):
    """state_dict keys (without .weight) of the LayerNorms that read the stream once it carries the massive channels"""
    tp = f"{P}encoder.transformer.layers"
    sites = []
    for i in range(outlier_layer(cfg) + 1, cfg.n_layers):
        if cfg.remaining_heads[i] or not cfg.layer_norm_first:
            sites.append(f"{tp}.{i}.layer_norm")
        sites.append(f"{tp}.{i}.final_layer_norm")
    return sites


def plant_outliers(sd: Dict[str, torch.Tensor], cfg: SegConfig, seed: int = 0, ln_sigma=None) -> Dict[str, torch.Tensor]:
    """`ln_sigma`: {LayerNorm key: std of the TYPICAL channels of its input} - the calibration (oracle/calibrate.py --outlier
    measures it site by site in one pass and stores it next to the classifier); a site without an entry takes `sigma`."""
    sd = {k: v.clone() for k, v in sd.items()}
    ln_sigma = ln_sigma or {}
    chans, mags, rms, sigma = outlier_plan(cfg, seed)
    tp = f"{P}encoder.transformer.layers"
    p = outlier_layer(cfg)
    inv = sigma / mags.abs()
    if cfg.layer_norm_first:
        sd[f"{tp}.{p}.feed_forward.output_dense.weight"][chans] *= 64.0
        sd[f"{tp}.{p}.feed_forward.output_dense.bias"][chans] = mags
    else:
        sd[f"{tp}.{p}.final_layer_norm.weight"][chans] = 64.0
        sd[f"{tp}.{p}.final_layer_norm.bias"][chans] = mags
        for i in range(p + 1, cfg.n_layers):
            lp = f"{tp}.{i}"
            if cfg.remaining_heads[i]:
                for nm in ("q_proj", "k_proj", "v_proj"):
                    sd[f"{lp}.attention.{nm}.weight"][:, chans] *= inv
            sd[f"{lp}.feed_forward.intermediate_dense.weight"][:, chans] *= inv
    for key in outlier_ln_sites(cfg):
        w, b = sd[key + ".weight"], sd[key + ".bias"]
        g_c = w[chans].clone()
        w *= rms / float(ln_sigma.get(key, sigma))                   # typical channels: x / rms  ->  x / sigma_site
        if cfg.layer_norm_first:
            w[chans] = g_c * rms * inv                               # massive ones: M / rms  ->  O(sigma)
        else:
            w[chans] = rms                                            # +-M / rms  ->  +-M again
            b[chans] = 0.0
    sd["proj.weight"][:, chans] *= inv
    return sd


def outlier_state_dict(cfg: SegConfig, seed: int = 0, calibration=None) -> Dict[str, torch.Tensor]:
    """turn-taking weights + planted massive activations; classifier from cal_<config>_outlier_seed<seed>.npz"""
    import numpy as np
    if calibration is None:
        calibration = CAL_DIR / f"cal_{cfg.name}_outlier_seed{seed}.npz"
    if not isinstance(calibration, dict):
        if not __import__("os").path.exists(calibration):
            raise FileNotFoundError(f"{calibration}: run `python oracle/calibrate.py --outlier` (build container) first")
        calibration = dict(np.load(calibration))
    ln_sigma = dict(zip(outlier_ln_sites(cfg), np.asarray(calibration["ln_sigma"], dtype=np.float64).tolist()))
    sd = plant_outliers(turn_taking_head(seg_state_dict(cfg, seed), cfg, seed), cfg, seed, ln_sigma)
    sd["classifier.weight"] = torch.as_tensor(np.asarray(calibration["W"], dtype=np.float32)).contiguous()
    sd["classifier.bias"] = torch.as_tensor(np.asarray(calibration["b"], dtype=np.float32)).contiguous()
    return sd


def emb_outlier_state_dict(seed: int = 0) -> Dict[str, torch.Tensor]:
    """`emb_state_dict` with planted BatchNorm outliers: two channels of the 32-plane stream (from layer1.0 on: the fused
    BasicBlock kernels) and two of the 128-plane stream (from layer3.0 on: the generic contractions over NHWC images) sit 2^10 x
    above the typical activation to the end of their stage (identity shortcuts carry them), with a token-varying part 2^5 x
    typical; every convolution that reads them has its input-channel slice / 2^10, as a trained network's would be."""
    sd = {k: v.clone() for k, v in emb_state_dict(seed).items()}
    g = torch.Generator().manual_seed(9100 + seed)

    def plant(block, planes, typical, readers):
        ch = torch.randperm(planes, generator=g)[:2]
        sd[f"resnet.{block}.bn2.weight"][ch] *= 32.0
        sd[f"resnet.{block}.bn2.bias"][ch] = torch.tensor([1.0, 1.5]) * typical * 1024.0
        for r in readers:
            sd[f"resnet.{r}.weight"][:, ch] /= 1024.0
        return ch

    plant("layer1.0", 32, 2.0, ["layer1.1.conv1", "layer1.2.conv1", "layer2.0.conv1", "layer2.0.shortcut.0"])
    plant("layer3.0", 128, 32.0, [f"layer3.{j}.conv1" for j in range(1, 6)] + ["layer4.0.conv1", "layer4.0.shortcut.0"])
    return sd
