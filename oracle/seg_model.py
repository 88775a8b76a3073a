"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product (diarizen_amd/).

CPU fp32 restatement of the reference segmentation forward
(diarizen/models/eend/model_wavlm_conformer.py:238-264 Model.forward), written as plain
functional torch over a state_dict that uses the reference's own key names.  Each step cites
the reference lines it follows.  Parity is PINNED: oracle/gen_golden.py loads the very same
state_dict into the reference's nn.Modules (imported from /root/reference) and stores their
outputs under tests/golden/; tests/test_oracle.py checks this restatement against them.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from diarizen_amd.configs import SegConfig

P = "wavlm_model."


# --------------------------------------------------------------------------- weights
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


# --------------------------------------------------------------------------- pieces
def relpos_bucket(rel: torch.Tensor, num_buckets: int, max_distance: int) -> torch.Tensor:
    """W2V/components.py:629-666 (_relative_positions_bucket, bidirectional=True)."""
    nb = num_buckets // 2
    out = (rel > 0).to(torch.long) * nb
    a = rel.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(a.float() / max_exact) / math.log(max_distance / max_exact)
                         * (nb - max_exact)).to(torch.long)
    large = torch.minimum(large, torch.full_like(large, nb - 1))
    return out + torch.where(a < max_exact, a, large)


def position_bias(sd, cfg: SegConfig, L: int) -> torch.Tensor:
    """W2V/components.py:612-627 (compute_bias) -> [H, L, L]; layer 0's table feeds all layers."""
    ctx = torch.arange(L)[:, None]
    mem = torch.arange(L)[None, :]
    bucket = relpos_bucket(mem - ctx, cfg.num_buckets, cfg.max_distance)
    emb = sd[f"{P}encoder.transformer.layers.0.attention.rel_attn_embed.weight"]
    return emb[bucket].permute(2, 0, 1)


def wavlm_attention(sd, cfg: SegConfig, i: int, y: torch.Tensor, pbias: torch.Tensor) -> torch.Tensor:
    """W2V/components.py:690-725 + :453-486 for layer i with kept heads cfg.remaining_heads[i]."""
    lp = f"{P}encoder.transformer.layers.{i}.attention"
    B, L, D = y.shape
    H = cfg.total_heads
    heads = list(cfg.remaining_heads[i])
    h = len(heads)
    # gate from the attention INPUT split into the H original heads (:702-710)
    yh = y.view(B, L, H, D // H).permute(0, 2, 1, 3)
    t = F.linear(yh, sd[lp + ".gru_rel_pos_linear.weight"], sd[lp + ".gru_rel_pos_linear.bias"])
    t = t.view(B, H, L, 2, 4).sum(-1)
    ga, gb = torch.sigmoid(t).chunk(2, dim=-1)
    gate = ga * (gb * sd[lp + ".gru_rel_pos_const"] - 1.0) + 2.0       # [B,H,L,1]
    bias = (gate * pbias[None])[:, heads]                               # [B,h,L,L] (:713)
    q = F.linear(y, sd[lp + ".q_proj.weight"], sd[lp + ".q_proj.bias"]).view(B, L, h, 64).transpose(1, 2)
    k = F.linear(y, sd[lp + ".k_proj.weight"], sd[lp + ".k_proj.bias"]).view(B, L, h, 64).permute(0, 2, 3, 1)
    v = F.linear(y, sd[lp + ".v_proj.weight"], sd[lp + ".v_proj.bias"]).view(B, L, h, 64).transpose(1, 2)
    w = (64 ** -0.5 * q) @ k + bias
    w = w - w.max(dim=-1, keepdim=True)[0]
    w = torch.softmax(w, dim=-1)
    o = (w @ v).transpose(1, 2).reshape(B, L, h * 64)
    return F.linear(o, sd[lp + ".out_proj.weight"], sd[lp + ".out_proj.bias"])


def feature_extractor(sd, cfg: SegConfig, wave: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
    """W2V/model.py:113 + components.py:182-209, 106-132.  wave [B,N] -> [B,L,C6]."""
    x = wave
    if cfg.normalize_waveform:
        x = F.layer_norm(x, x.shape[-1:])                      # per-window, eps 1e-5, no affine
    x = x.unsqueeze(1)
    for i, s in enumerate(cfg.conv_strides):
        pre = f"{P}feature_extractor.conv_layers.{i}"
        x = F.conv1d(x, sd[pre + ".conv.weight"], stride=s)
        if cfg.extractor_layer_norm:                           # LN over channels per frame (:63-70)
            x = F.layer_norm(x.transpose(1, 2), (x.shape[1],), sd[pre + ".layer_norm.weight"],
                             sd[pre + ".layer_norm.bias"]).transpose(1, 2)
        elif i == 0:                                           # GroupNorm(C,C): per channel over time
            x = F.group_norm(x, x.shape[1], sd[pre + ".layer_norm.weight"], sd[pre + ".layer_norm.bias"])
        x = F.gelu(x)
        if taps is not None:
            taps[f"conv{i}"] = x.transpose(1, 2).contiguous()
    x = x.transpose(1, 2) * sd[P + "feature_extractor.dummy_weight"]
    return x


def encoder(sd, cfg: SegConfig, feats: torch.Tensor, taps: Optional[dict] = None) -> List[torch.Tensor]:
    """components.py:1151-1159 -> :1129 (feature projection) -> :1004-1024 (intermediates)."""
    tp = P + "encoder.transformer"
    D = cfg.embed_dim
    x = F.layer_norm(feats, (feats.shape[-1],), sd[P + "encoder.feature_projection.layer_norm.weight"],
                     sd[P + "encoder.feature_projection.layer_norm.bias"])
    x = F.linear(x, sd[P + "encoder.feature_projection.projection.weight"],
                 sd[P + "encoder.feature_projection.projection.bias"])
    if taps is not None:
        taps["featproj"] = x
    # positional conv with weight-norm over dims (0,1) per tap (components.py:344,366-380)
    g_ = sd[tp + ".pos_conv_embed.conv.parametrizations.weight.original0"]
    v_ = sd[tp + ".pos_conv_embed.conv.parametrizations.weight.original1"]
    w = g_ * v_ / v_.norm(dim=(0, 1), keepdim=True)
    pc = F.conv1d(x.transpose(1, 2), w, sd[tp + ".pos_conv_embed.conv.bias"],
                  padding=cfg.pos_conv_kernel // 2, groups=cfg.pos_conv_groups)
    if cfg.pos_conv_kernel % 2 == 0:
        pc = pc[..., :-1]
    x = x + F.gelu(pc).transpose(1, 2)
    if not cfg.layer_norm_first:      # Transformer.layer_norm_first == not encoder_layer_norm_first (:1595)
        x = F.layer_norm(x, (D,), sd[tp + ".layer_norm.weight"], sd[tp + ".layer_norm.bias"])
    reps = [x]
    pbias = position_bias(sd, cfg, x.shape[1])
    for i in range(cfg.n_layers):
        lp = f"{tp}.layers.{i}"
        ln1 = (sd[lp + ".layer_norm.weight"], sd[lp + ".layer_norm.bias"])
        ln2 = (sd[lp + ".final_layer_norm.weight"], sd[lp + ".final_layer_norm.bias"])

        def ffn(z):
            z = F.gelu(F.linear(z, sd[lp + ".feed_forward.intermediate_dense.weight"],
                                sd[lp + ".feed_forward.intermediate_dense.bias"]))
            return F.linear(z, sd[lp + ".feed_forward.output_dense.weight"],
                            sd[lp + ".feed_forward.output_dense.bias"])

        if cfg.remaining_heads[i]:
            r = x
            y = F.layer_norm(x, (D,), *ln1) if cfg.layer_norm_first else x
            x = r + wavlm_attention(sd, cfg, i, y, pbias)
        if cfg.layer_norm_first:
            x = x + ffn(F.layer_norm(x, (D,), *ln2))
        else:
            x = F.layer_norm(x, (D,), *ln1)
            x = x + ffn(x)
            x = F.layer_norm(x, (D,), *ln2)
        reps.append(x)
        if taps is not None:
            taps[f"layer{i}"] = x
    return reps


def conformer_block(sd, cfg: SegConfig, i: int, x: torch.Tensor) -> torch.Tensor:
    """diarizen/models/module/conformer.py:247-257."""
    cp = f"conformer.conformer_layer.{i}"
    A = cfg.attention_in
    B, T, _ = x.shape

    def half_ffn(z, name):                                         # :136-144
        r = z
        z = F.layer_norm(z, (A,), sd[f"{cp}.{name}.ln_norm.weight"], sd[f"{cp}.{name}.ln_norm.bias"])
        z = F.linear(z, sd[f"{cp}.{name}.w_1.weight"], sd[f"{cp}.{name}.w_1.bias"])
        z = z * torch.sigmoid(z)
        z = F.linear(z, sd[f"{cp}.{name}.w_2.weight"], sd[f"{cp}.{name}.w_2.bias"])
        return r + 0.5 * z

    x = half_ffn(x, "ffn1")
    # MHSA (:102-114, :47-71), no positional term
    r = x
    z = F.layer_norm(x, (A,), sd[f"{cp}.mha.ln_norm.weight"], sd[f"{cp}.mha.ln_norm.bias"])
    h, dk = cfg.conf_heads, A // cfg.conf_heads
    q, k, v = [F.linear(z, sd[f"{cp}.mha.mha.linear{n}.weight"], sd[f"{cp}.mha.mha.linear{n}.bias"])
               .view(B, T, h, dk).transpose(1, 2) for n in "QKV"]
    att = torch.softmax(q @ k.transpose(-2, -1) / math.sqrt(dk), dim=3)
    z = (att @ v).permute(0, 2, 1, 3).reshape(B, T, h * dk)
    x = r + F.linear(z, sd[f"{cp}.mha.mha.linearO.weight"], sd[f"{cp}.mha.mha.linearO.bias"])
    # conv module (:192-214)
    r = x
    z = F.layer_norm(x, (A,), sd[f"{cp}.conv.ln_norm.weight"], sd[f"{cp}.conv.ln_norm.bias"]).transpose(1, 2)
    z = F.conv1d(z, sd[f"{cp}.conv.pointwise_conv1.weight"], sd[f"{cp}.conv.pointwise_conv1.bias"])
    z = F.glu(z, dim=1)
    z = F.conv1d(z, sd[f"{cp}.conv.depthwise_conv.weight"], sd[f"{cp}.conv.depthwise_conv.bias"],
                 padding=(cfg.conf_kernel - 1) // 2, groups=A)
    z = F.batch_norm(z, sd[f"{cp}.conv.bn_norm.running_mean"], sd[f"{cp}.conv.bn_norm.running_var"],
                     sd[f"{cp}.conv.bn_norm.weight"], sd[f"{cp}.conv.bn_norm.bias"], training=False, eps=1e-5)
    z = z * torch.sigmoid(z)
    z = F.conv1d(z, sd[f"{cp}.conv.pointwise_conv2.weight"], sd[f"{cp}.conv.pointwise_conv2.bias"])
    x = r + z.transpose(1, 2)
    x = half_ffn(x, "ffn2")
    return F.layer_norm(x, (A,), sd[f"{cp}.ln_norm.weight"], sd[f"{cp}.ln_norm.bias"])


def powerset_mapping(num_classes: int, max_set_size: int) -> torch.Tensor:
    """PA/utils/powerset.py:68-97 (build_mapping): rows ordered by set size then combinations."""
    from itertools import combinations
    rows = []
    for size in range(max_set_size + 1):
        for comb in combinations(range(num_classes), size):
            r = torch.zeros(num_classes)
            r[list(comb)] = 1.0
            rows.append(r)
    return torch.stack(rows)


def to_multilabel(logp: torch.Tensor, cfg: SegConfig) -> torch.Tensor:
    """PA/utils/powerset.py:103-128, soft=False: one_hot(argmax) @ mapping."""
    mapping = powerset_mapping(cfg.max_speakers_per_chunk, cfg.max_speakers_per_frame)
    hard = F.one_hot(torch.argmax(logp, dim=-1), mapping.shape[0]).float()
    return hard @ mapping


@torch.inference_mode()
def seg_forward(sd, cfg: SegConfig, wave: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
    """wave [B, N] (channel already selected, model_wavlm_conformer.py:250) -> logp [B, L, n_classes]."""
    feats = feature_extractor(sd, cfg, wave, taps)
    if taps is not None:
        taps["features"] = feats
    reps = encoder(sd, cfg, feats, taps)
    # model_wavlm_conformer.py:231-236, 252-254: stack -> Linear(layers -> 1, no bias)
    ws = sd["weight_sum.weight"][0]
    z = torch.stack(reps, dim=-1) @ ws
    if taps is not None:
        taps["wsum"] = z
    z = F.layer_norm(F.linear(z, sd["proj.weight"], sd["proj.bias"]), (cfg.attention_in,),
                     sd["lnorm.weight"], sd["lnorm.bias"])          # :256-257
    if taps is not None:
        taps["head_in"] = z
    for i in range(cfg.conf_layers):
        z = conformer_block(sd, cfg, i, z)
        if taps is not None:
            taps[f"conf{i}"] = z
    logits = F.linear(z, sd["classifier.weight"], sd["classifier.bias"])   # :261
    return F.log_softmax(logits, dim=-1)                                   # :262, PA/core/model.py:223-224
