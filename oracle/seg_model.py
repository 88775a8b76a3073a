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

from oracle.configs import OracleSegConfig as SegConfig   # typing only: any object with these attributes works

P = "wavlm_model."


# --------------------------------------------------------------------------- weights
from testkit.weights import seg_state_dict  # noqa: E402,F401  (seeded random init lives in the product)


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
