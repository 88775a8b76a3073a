"""ORACLE — TEST INFRASTRUCTURE ONLY.

Generates the golden fixtures under tests/golden/ by running the REFERENCE's own modules
(imported from /root/reference, CPU fp32) on seeded inputs and seeded random weights.
Run in the build container only (the GPU box has no /root/reference):

    python oracle/gen_golden.py            # all fixtures
    python oracle/gen_golden.py seg emb    # a subset

What imports and what does not: the arithmetic modules import
(diarizen.models.module.wav2vec2, conformer, wespeaker/resnet.py, pooling.py, powerset.py,
VBx.py); the wrappers that need pyannote.core / lightning / torchaudio do not, so the few
glue lines around them (Model.forward, model_wavlm_conformer.py:250-262) are restated here
and cited.  The fixtures pin oracle/*.py; the HIP path is then checked against the oracle
and, at the fixture sizes, directly against these files.
"""
from __future__ import annotations

import importlib.util
import sys
import types
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
REF = Path("/root/reference")
GOLD = ROOT / "tests" / "golden"
sys.path.insert(0, str(ROOT))

from oracle.configs import get_seg_config  # noqa: E402
from oracle import seg_model  # noqa: E402


def _ref_path():
    if not REF.exists():
        raise SystemExit("/root/reference not present: golden fixtures can only be generated "
                         "in the build container")
    if str(REF) not in sys.path:
        sys.path.insert(0, str(REF))


def _load_by_path(name: str, path: Path, stubs: dict | None = None):
    """import a reference file by path with stub parent packages (pyannote.* is not installed)."""
    for k, v in (stubs or {}).items():
        sys.modules.setdefault(k, v)
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def synth_wave(B: int, N: int, seed: int) -> torch.Tensor:
    """deterministic speech-like test signal: amplitude-modulated noise + tones, |x| < 1"""
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(N) / 16000.0
    x = 0.05 * torch.randn(B, N, generator=g)
    for b in range(B):
        f0 = 90.0 + 40.0 * b
        env = (torch.sin(2 * np.pi * (0.7 + 0.3 * b) * t) > -0.2).float()
        x[b] += 0.2 * env * torch.sin(2 * np.pi * f0 * t) + 0.1 * env * torch.sin(2 * np.pi * 3.1 * f0 * t)
    return x.clamp(-1, 1)


# ------------------------------------------------------------------ segmentation model
def build_reference_seg(cfg, sd):
    """Reference modules wired as Model.__init__ does (model_wavlm_conformer.py:58-76)."""
    _ref_path()
    from diarizen.models.module.wav2vec2.model import wav2vec2_model
    from diarizen.models.module.wavlm_config import get_config
    from diarizen.models.module.conformer import ConformerEncoder
    import torch.nn as nn

    if cfg.name.startswith("tiny"):
        rc = dict(get_config("wavlm_large_s80_md" if cfg.extractor_layer_norm else "wavlm_base_s80_md"))
        rc["extractor_conv_layer_config"] = [(c, k, s) for c, k, s in
                                             zip(cfg.conv_channels, cfg.conv_kernels, cfg.conv_strides)]
        rc["encoder_embed_dim"] = cfg.embed_dim
        rc["encoder_pos_conv_kernel"] = cfg.pos_conv_kernel
        rc["encoder_pos_conv_groups"] = cfg.pos_conv_groups
        rc["encoder_num_layers"] = cfg.n_layers
        rc["encoder_use_attention"] = list(cfg.use_attention)
        rc["encoder_use_feed_forward"] = [True] * cfg.n_layers
        rc["encoder_total_num_heads"] = [cfg.total_heads] * cfg.n_layers
        rc["encoder_remaining_heads"] = [list(h) for h in cfg.remaining_heads]
        rc["encoder_ff_interm_features"] = list(cfg.ffn_dims)
    else:
        rc = get_config(cfg.name)
    wavlm = wav2vec2_model(**rc)
    conformer = ConformerEncoder(attention_in=cfg.attention_in, ffn_hidden=cfg.ffn_hidden,
                                 num_head=cfg.conf_heads, num_layer=cfg.conf_layers,
                                 kernel_size=cfg.conf_kernel, dropout=0.1, use_posi=False,
                                 output_activate_function=False)
    weight_sum = nn.Linear(cfg.wavlm_layer_num, 1, bias=False)
    proj = nn.Linear(cfg.embed_dim, cfg.attention_in)
    lnorm = nn.LayerNorm(cfg.attention_in)
    classifier = nn.Linear(cfg.attention_in, cfg.n_classes)

    def sub(prefix):
        return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}

    wavlm.load_state_dict(sub("wavlm_model."), strict=True)
    conformer.load_state_dict(sub("conformer."), strict=True)
    weight_sum.load_state_dict(sub("weight_sum."), strict=True)
    proj.load_state_dict(sub("proj."), strict=True)
    lnorm.load_state_dict(sub("lnorm."), strict=True)
    classifier.load_state_dict(sub("classifier."), strict=True)
    for m in (wavlm, conformer, weight_sum, proj, lnorm, classifier):
        m.eval()

    @torch.inference_mode()
    def forward(wave):  # model_wavlm_conformer.py:250-262, waveforms[:, selected_channel, :] already applied
        layer_reps, _ = wavlm.extract_features(wave)
        feat = torch.stack(layer_reps, dim=-1)
        feat = torch.squeeze(weight_sum(feat), -1)
        out = lnorm(proj(feat))
        out = conformer(out)
        out = classifier(out)
        return torch.log_softmax(out, dim=-1), layer_reps

    return forward


def gen_seg():
    cases = [("tiny_ln", 2, 8000, 0, 11), ("tiny_gn", 2, 8000, 0, 12),
             ("wavlm_large_s80_md", 1, 16000, 0, 13), ("wavlm_base_s80_md", 1, 16000, 0, 14)]
    for name, B, N, wseed, xseed in cases:
        cfg = get_seg_config(name)
        sd = seg_model.seg_state_dict(cfg, wseed)
        fwd = build_reference_seg(cfg, sd)
        wave = synth_wave(B, N, xseed)
        logp, reps = fwd(wave)
        np.savez_compressed(GOLD / f"seg_{name}.npz", B=B, N=N, weight_seed=wseed, wave_seed=xseed,
                            logp=logp.numpy(), rep0=reps[0].numpy(), rep_last=reps[-1].numpy())
        print(f"seg_{name}: logp {tuple(logp.shape)} argmax hist "
              f"{np.bincount(logp.argmax(-1).flatten().numpy(), minlength=cfg.n_classes).tolist()}")


def tt_windows(starts, N: int) -> torch.Tensor:
    """windows of tests/golden/EN2002a_30s.wav starting at `starts` (samples)"""
    from oracle.wav import first_channel_pcm16 as first_channel_16k
    wave = torch.from_numpy(first_channel_16k(str(GOLD / "EN2002a_30s.wav")))
    return torch.stack([wave[s:s + N] for s in starts])


# turn-taking (non-degenerate) fixtures: config -> (N, window starts).  Includes BASELINE configs[1] at full
# size (wavlm-base-s80, 5 s windows, batch 32) and the bench geometry (large-s80, 8 s windows).
TT_CASES = {
    "tiny_ln": (8000, [32000, 96000]),
    "tiny_gn": (8000, [48000, 160000]),
    "wavlm_large_s80_md": (128000, [0, 192000]),
    "wavlm_base_s80_md": (80000, [8000 * i for i in range(32)]),
}


def gen_seg_tt():
    """Reference modules (strict state_dict load) with the seeded turn-taking weights on real audio: logp
    goldens whose argmax is NOT constant (asserted), at the sizes BASELINE.json names."""
    from testkit.weights import turn_taking_state_dict
    for name, (N, starts) in TT_CASES.items():
        cfg = get_seg_config(name)
        sd = turn_taking_state_dict(cfg, 0)
        fwd = build_reference_seg(cfg, sd)
        wave = tt_windows(starts, N)
        outs = [fwd(wave[b0:b0 + 8])[0] for b0 in range(0, len(starts), 8)]
        logp = torch.cat(outs)
        am = logp.argmax(-1).numpy()
        hist = np.bincount(am.ravel(), minlength=cfg.n_classes)
        srt = np.sort(logp.numpy(), -1)
        margin = srt[..., -1] - srt[..., -2]
        print(f"seg_tt_{name}: logp {tuple(logp.shape)} argmax hist {hist.tolist()} "
              f"transitions/window {(am[:, 1:] != am[:, :-1]).sum(1).mean():.1f} min top-2 margin {margin.min():.2e}")
        assert (hist > 0).sum() >= 5, "degenerate fixture"
        np.savez_compressed(GOLD / f"seg_tt_{name}.npz", N=N, starts=np.array(starts), weight_seed=0,
                            logp=logp.numpy(), min_margin=margin.min())


# planted massive activations (testkit/weights.py:outlier_state_dict): config -> (N, window starts)
OUTLIER_CASES = {
    "tiny_ln": (8000, [32000, 96000]),
    "tiny_gn": (8000, [48000, 160000]),
    "wavlm_large_s80_md": (128000, [0, 192000]),
    "wavlm_base_s80_md": (80000, [40000 * i for i in range(8)]),
}


def gen_seg_outlier():
    """Reference modules (strict load) with the planted-outlier weights on real audio.  Also runs the SAME reference modules in
    float64 and stores that result: |fp32 - fp64| of the reference itself says how much of the 1e-3 bar the fixture leaves to
    an implementation whose fp32 operations are merely ordered differently (asserted <= 3e-4)."""
    from testkit.weights import outlier_plan, outlier_state_dict
    for name, (N, starts) in OUTLIER_CASES.items():
        cfg = get_seg_config(name)
        sd = outlier_state_dict(cfg, 0)
        fwd = build_reference_seg(cfg, sd)
        wave = tt_windows(starts, N)
        outs = [fwd(wave[b0:b0 + 8]) for b0 in range(0, len(starts), 8)]
        logp = torch.cat([o[0] for o in outs])
        rep_last = torch.cat([o[1][-1] for o in outs])
        chans, mags, rms, sigma = outlier_plan(cfg, 0)
        typ = torch.ones(cfg.embed_dim, dtype=torch.bool)
        typ[chans] = False
        ratio = float(rep_last[..., chans].abs().max() / rep_last[..., typ].std())
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        torch.set_default_dtype(torch.float64)
        try:
            fwd64 = build_reference_seg(cfg, sd64)
            logp64 = torch.cat([fwd64(wave[b0:b0 + 8].double())[0] for b0 in range(0, len(starts), 8)])
        finally:
            torch.set_default_dtype(torch.float32)
        ref_noise = float((logp.double() - logp64).abs().max())
        am = logp.argmax(-1).numpy()
        hist = np.bincount(am.ravel(), minlength=cfg.n_classes)
        srt = np.sort(logp.numpy(), -1)
        margin = srt[..., -1] - srt[..., -2]
        print(f"seg_outlier_{name}: logp {tuple(logp.shape)} argmax hist {hist.tolist()} transitions/window "
              f"{(am[:, 1:] != am[:, :-1]).sum(1).mean():.1f} min top-2 margin {margin.min():.2e}; last layer: massive / typical "
              f"= {ratio:.0f}; reference fp32 vs its own float64 run: {ref_noise:.2e}")
        assert (hist > 0).sum() >= 5, "degenerate fixture"
        assert ratio >= 256, "the outliers did not survive to the last layer"
        assert ref_noise <= 3e-4, "fixture too ill-conditioned for a 1e-3 bar"
        np.savez_compressed(GOLD / f"seg_outlier_{name}.npz", N=N, starts=np.array(starts), weight_seed=0,
                            logp=logp.numpy(), logp64=logp64.numpy(), min_margin=margin.min(), chans=chans.numpy(),
                            massive_over_typical=ratio, ref_fp32_vs_fp64=ref_noise,
                            rep_last_outlier=rep_last[..., chans].numpy())


# loudness extremes in ONE batch: per-window operand scales must not leak between windows.  window 0 as recorded, 1 near-silent
# (x 1e-4: below the eps of the waveform LayerNorm and of conv0's channel norm), 2 clipped (x 8, clamped to +-1), 3 digital silence
LOUD_CASES = {"tiny_ln": (8000, 32000), "tiny_gn": (8000, 48000), "wavlm_large_s80_md": (128000, 192000),
              "wavlm_base_s80_md": (80000, 120000)}


def loud_windows(N: int, start: int) -> torch.Tensor:
    w = tt_windows([start], N)[0]
    return torch.stack([w, w * 1e-4, (w * 8.0).clamp(-1.0, 1.0), torch.zeros_like(w)])


def gen_seg_loud():
    from testkit.weights import turn_taking_state_dict
    for name, (N, start) in LOUD_CASES.items():
        cfg = get_seg_config(name)
        sd = turn_taking_state_dict(cfg, 0)
        fwd = build_reference_seg(cfg, sd)
        wave = loud_windows(N, start)
        logp = fwd(wave)[0]
        alone = torch.cat([fwd(wave[b:b + 1])[0] for b in range(4)])
        print(f"seg_loud_{name}: logp {tuple(logp.shape)} classes per window "
              f"{[len(np.unique(logp[b].argmax(-1).numpy())) for b in range(4)]}; batch vs one-by-one in the reference "
              f"{(logp - alone).abs().max():.1e}; finite {bool(torch.isfinite(logp).all())}")
        assert torch.isfinite(logp).all()
        np.savez_compressed(GOLD / f"seg_loud_{name}.npz", N=N, start=start, weight_seed=0, logp=logp.numpy())


# ------------------------------------------------------------------ embedding model
def load_reference_resnet():
    """wespeaker/resnet.py + blocks/pooling.py + utils/receptive_field.py by file path, with
    stub parent packages (pyannote.* is not installed here)."""
    _ref_path()
    PA = REF / "pyannote-audio" / "pyannote" / "audio"
    for pkg in ("pyannote", "pyannote.audio", "pyannote.audio.models", "pyannote.audio.models.blocks",
                "pyannote.audio.utils", "pyannote.audio.models.embedding",
                "pyannote.audio.models.embedding.wespeaker"):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = []
            sys.modules[pkg] = m
    _load_by_path("pyannote.audio.utils.receptive_field", PA / "utils" / "receptive_field.py")
    pooling = _load_by_path("pyannote.audio.models.blocks.pooling", PA / "models" / "blocks" / "pooling.py")
    resnet = _load_by_path("pyannote.audio.models.embedding.wespeaker.resnet",
                           PA / "models" / "embedding" / "wespeaker" / "resnet.py")
    return resnet, pooling


def gen_emb():
    import warnings
    from oracle import emb_model
    resnet, pooling = load_reference_resnet()
    sd = emb_model.emb_state_dict(0)
    net = resnet.ResNet34(80, 256, pooling_func="TSTP", two_emb_layer=False)
    net.load_state_dict({k[len("resnet."):]: v for k, v in sd.items()}, strict=True)
    net.eval()
    B, N = 2, 24000
    wave = synth_wave(B, N, 31)
    fb = emb_model.compute_fbank(wave)              # fbank is third-party (torchaudio): oracle's own
    L = 74                                          # frames of a 1.5 s window in the seg model
    g = torch.Generator().manual_seed(5)
    masks = (torch.rand(B, 4, L, generator=g) > 0.5).float()
    masks[0, 2] = 0.0                               # inactive speaker -> embedding == seg_1.bias
    masks[1, 3, 5:] = 0.0                           # very few frames
    with torch.inference_mode(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        embs = torch.stack([net(fb.clone(), weights=masks[:, s])[1] for s in range(4)], dim=1)
        multi = net(fb.clone(), weights=masks)[1]  # (batch, speakers, frames) path of StatsPool
    np.savez_compressed(GOLD / "emb_resnet.npz", B=B, N=N, L=L, wave_seed=31, weight_seed=0,
                        masks=masks.numpy(), fbank=fb.numpy(), emb=embs.numpy(), emb_multi=multi.numpy())
    print("emb_resnet:", tuple(embs.shape), "multi-vs-single max diff",
          (embs - multi).abs().max().item(), "zero-mask == bias:",
          (embs[0, 2] - sd["resnet.seg_1.bias"]).abs().max().item())


def gen_emb_outlier():
    """The reference ResNet34 with planted BatchNorm outliers (testkit/weights.py:emb_outlier_state_dict) on the inputs of
    gen_emb; the float64 run of the same module says what the fixture leaves of the bar."""
    import warnings
    from oracle import emb_model
    from testkit.weights import emb_outlier_state_dict
    resnet, pooling = load_reference_resnet()
    sd = emb_outlier_state_dict(0)
    net = resnet.ResNet34(80, 256, pooling_func="TSTP", two_emb_layer=False)
    net.load_state_dict({k[len("resnet."):]: v for k, v in sd.items()}, strict=True)
    net.eval()
    B, N, L = 2, 24000, 74
    wave = synth_wave(B, N, 31)
    fb = emb_model.compute_fbank(wave)
    g = torch.Generator().manual_seed(5)
    masks = (torch.rand(B, 4, L, generator=g) > 0.5).float()
    masks[0, 2] = 0.0
    masks[1, 3, 5:] = 0.0
    with torch.inference_mode(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        embs = torch.stack([net(fb.clone(), weights=masks[:, s])[1] for s in range(4)], dim=1)
        net64 = net.double()
        embs64 = torch.stack([net64(fb.double(), weights=masks[:, s].double())[1] for s in range(4)], dim=1)
    rel = float((embs.double() - embs64).abs().max() / embs64.abs().max())
    taps = {}
    trunk = emb_model.resnet_trunk(sd, fb)
    print("emb_resnet_outlier:", tuple(embs.shape), "reference fp32 vs its own float64 run, rel:", f"{rel:.1e}",
          "|emb| max", float(embs.abs().max()), "trunk out max", float(trunk.abs().max()))
    assert rel < 2e-5
    np.savez_compressed(GOLD / "emb_resnet_outlier.npz", B=B, N=N, L=L, wave_seed=31, weight_seed=0,
                        masks=masks.numpy(), emb=embs.numpy(), emb64=embs64.numpy(), ref_fp32_vs_fp64_rel=rel)


def gen_statspool_powerset():
    """Known answers from the reference's OWN unit tests, evaluated through the reference's own
    modules: pyannote-audio/tests/test_stats_pool.py:28-131, tests/utils/test_powerset.py:29-76."""
    _, pooling = load_reference_resnet()
    PA = REF / "pyannote-audio" / "pyannote" / "audio"
    sp = pooling.StatsPool()
    x = torch.Tensor([[[2.0, 4.0], [2.0, 4.0]], [[1.0, 1.0], [1.0, 1.0]]])
    w1 = torch.Tensor([[0.5, 0.01], [0.2, 0.1]])
    w2 = torch.Tensor([[[0.1, 0.2], [0.2, 0.3]], [[0.001, 0.001], [0.2, 0.3]]])
    w3 = torch.Tensor([[[0.1, 0.2, 0.3], [0.2, 0.3, 0.4]], [[0.001, 0.001, 0.002], [0.2, 0.3, 0.4]]])  # frame mismatch
    w0 = torch.zeros(2, 2)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        outs = dict(x=x.numpy(), w1=w1.numpy(), w2=w2.numpy(), w3=w3.numpy(), w0=w0.numpy(),
                    y_none=sp(x).numpy(), y_w1=sp(x, weights=w1).numpy(), y_w2=sp(x, weights=w2).numpy(),
                    y_w3=sp(x, weights=w3).numpy(), y_w0=sp(x, weights=w0).numpy())
    # powerset mapping of the reference (scipy.special + itertools inside)
    stubs = {}
    pw = _load_by_path("pyannote.audio.utils.powerset", PA / "utils" / "powerset.py", stubs)
    for nc, ms in [(4, 2), (3, 2), (5, 3), (2, 1)]:
        outs[f"mapping_{nc}_{ms}"] = pw.Powerset(nc, ms).mapping.numpy()
    np.savez_compressed(GOLD / "statspool_powerset.npz", **outs)
    print("statspool/powerset known answers:", {k: v.shape for k, v in outs.items() if k.startswith("y_")})


# ------------------------------------------------------------------ host clustering
def synth_vbx_case(E, K, D=128, seed=4):
    """PLDA-space features with K - 1 real speakers, a diagonal Phi and an imperfect one-hot AHC initialisation smoothed
    as VBxClustering does (softmax(7 q)): input of the VBx mixture tests (tests/test_host.py, tests/test_ops_gpu.py)."""
    from scipy.special import softmax
    r = np.random.default_rng(seed)
    mu = 3.0 * r.standard_normal((K - 1, D))
    lab = r.integers(0, K - 1, E)
    X = mu[lab] + r.standard_normal((E, D))
    Phi = np.abs(r.standard_normal(D)) * 2.0 + 0.05
    q0 = np.zeros((E, K))
    noisy = np.where(r.random(E) < 0.15, r.integers(0, K, E), lab)       # an imperfect AHC initialisation
    q0[np.arange(E), noisy] = 1.0
    return X, Phi, softmax(q0 * 7.0, axis=1)


def synth_host_case(seed: int, C: int = 60, L: int = 99, S: int = 4, D: int = 256, n_spk: int = 3):
    """synthetic per-window decisions + embeddings with a known speaker structure"""
    g = np.random.default_rng(seed)
    protos = g.normal(size=(n_spk, D))
    seg = np.zeros((C, L, S), dtype=np.float32)
    emb = np.zeros((C, S, D), dtype=np.float32)
    for c in range(C):
        k = g.integers(1, min(n_spk, 3) + 1)
        who = g.permutation(n_spk)[:k]
        slots = g.permutation(S)[:k]
        for spk, slot in zip(who, slots):
            a = g.integers(0, L // 2)
            b = g.integers(a + 5, L)
            seg[c, a:b, slot] = 1.0
        for slot in range(S):
            if seg[c, :, slot].sum() > 0:
                spk = who[list(slots).index(slot)]
                emb[c, slot] = protos[spk] + 0.35 * g.normal(size=D)
            else:
                emb[c, slot] = 0.1 * g.normal(size=D)      # "bias-like" embedding of an inactive speaker
    return seg, emb


def load_reference_clustering():
    """PA/pipelines/clustering.py + diarizen/clustering/VBx.py with stubbed pyannote.* parents."""
    _ref_path()
    PA = REF / "pyannote-audio" / "pyannote" / "audio"

    def stub(name, **attrs):
        m = sys.modules.get(name)
        if m is None:
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
        for k, v in attrs.items():
            setattr(m, k, v)
        return m

    class _P:  # pyannote.pipeline.Pipeline / parameter stand-ins (hyper-parameters are set directly)
        def __init__(self, *a, **k):
            pass

    from oracle import pyannote_core_stub as mycore
    stub("pyannote")
    stub("pyannote.core", SlidingWindow=mycore.SlidingWindow, SlidingWindowFeature=mycore.SlidingWindowFeature,
         Segment=mycore.Segment, Annotation=mycore.Annotation)
    stub("pyannote.pipeline", Pipeline=_P)
    stub("pyannote.pipeline.parameter", Categorical=_P, Integer=_P, Uniform=_P)
    stub("pyannote.audio")
    stub("pyannote.audio.core")
    stub("pyannote.audio.core.io", AudioFile=object)
    stub("pyannote.audio.pipelines")
    stub("pyannote.audio.pipelines.utils", oracle_segmentation=None)
    stub("pyannote.audio.utils")
    stub("pyannote.audio.utils.permutation", permutate=None)
    return _load_by_path("pyannote.audio.pipelines.clustering", PA / "pipelines" / "clustering.py")


def gen_host():
    import tempfile
    cl = load_reference_clustering()
    out = {}
    for i, (seed, C, nspk, thr, mcs) in enumerate([(1, 60, 3, 0.6, 5), (2, 120, 4, 0.7, 13), (3, 12, 2, 0.6, 13),
                                                   (4, 200, 5, 0.5, 8)]):
        seg, emb = synth_host_case(seed, C=C, n_spk=nspk)
        ahc = cl.AgglomerativeClustering(metric="cosine")
        ahc.method, ahc.threshold, ahc.min_cluster_size = "centroid", thr, mcs
        swf = types.SimpleNamespace(data=seg)
        hard, soft, cent = ahc(embeddings=emb.copy(), segmentations=swf, min_clusters=1, max_clusters=20)
        out[f"ahc{i}_args"] = np.array([seed, C, nspk, thr, mcs], dtype=np.float64)
        out[f"ahc{i}_hard"] = hard
        out[f"ahc{i}_centroids"] = cent
    # VBx with a synthetic (seeded) PLDA model in the hub's file format (VBx.py:171-175)
    g = np.random.default_rng(7)
    D, Dl = 256, 128
    with tempfile.TemporaryDirectory() as td:
        a = g.normal(size=(Dl, Dl))
        np.savez(f"{td}/xvec_transform.npz", mean1=0.1 * g.normal(size=D), mean2=0.1 * g.normal(size=Dl),
                 lda=g.normal(size=(D, Dl)) / np.sqrt(D))
        np.savez(f"{td}/plda.npz", mu=0.1 * g.normal(size=Dl), tr=a / np.sqrt(Dl) + np.eye(Dl),
                 psi=np.sort(g.uniform(0.5, 30.0, size=Dl))[::-1].copy())
        for f in ("xvec_transform.npz", "plda.npz"):
            out["plda_" + f.replace(".npz", "")] = np.frombuffer(open(f"{td}/{f}", "rb").read(), dtype=np.uint8)
        for i, (seed, C, nspk) in enumerate([(11, 80, 3), (12, 150, 4)]):
            seg, emb = synth_host_case(seed, C=C, n_spk=nspk)
            vb = cl.VBxClustering(metric="cosine", plda_dir=td, lda_dim=128, maxIters=20)
            vb.ahc_criterion, vb.ahc_threshold, vb.Fa, vb.Fb = "distance", 0.6, 0.07, 0.8
            hard, soft, cent = vb(embeddings=emb.copy(), segmentations=types.SimpleNamespace(data=seg))
            out[f"vbx{i}_args"] = np.array([seed, C, nspk], dtype=np.float64)
            out[f"vbx{i}_hard"] = hard
            out[f"vbx{i}_centroids"] = cent
    np.savez_compressed(GOLD / "host_clustering.npz", **out)
    print("host_clustering:", {k: v.shape for k, v in out.items() if k.endswith("_hard")},
          "clusters:", [int(out[k].max()) + 1 for k in out if k.endswith("_hard")])


# ------------------------------------------------------------------ end-to-end (BASELINE configs[0])
E2E_CONFIG = {
    "model": {"path": "diarizen.models.eend.model_wavlm_conformer.Model",
              "args": {"wavlm_src": "wavlm_large_s80_md", "wavlm_layer_num": 25, "wavlm_feat_dim": 1024,
                       "attention_in": 256, "ffn_hidden": 1024, "num_head": 4, "num_layer": 4,
                       "kernel_size": 31, "chunk_size": 8, "max_speakers_per_chunk": 4,
                       "max_speakers_per_frame": 2}},
    "inference": {"args": {"seg_duration": 8, "segmentation_step": 0.1, "batch_size": 32,
                           "apply_median_filtering": True}},
    "clustering": {"args": {"method": "AgglomerativeClustering", "min_speakers": 1, "max_speakers": 20,
                            "ahc_criterion": "distance", "ahc_threshold": 0.1, "min_cluster_size": 3}},
}


def mask_branches(seg: np.ndarray, window: int):
    """(number of (window, speaker) pairs that use the overlap-excluded mask, number that fall back to
    the full mask although active) — PA/pipelines/speaker_diarization.py:268-322."""
    import math
    L = seg.shape[1]
    min_num_frames = math.ceil(L * 400 / window)
    segf = seg.astype(np.float32)
    clean = segf * (segf.sum(2, keepdims=True) < 2)
    use_clean = clean.sum(1) > min_num_frames
    active = segf.sum(1) > 0
    return int(use_clean.sum()), int((active & ~use_clean).sum())


def decision_stats(seg: np.ndarray) -> dict:
    """non-degeneracy report of hard decisions [C, L, S] (VERDICT r1 'next' #1)."""
    code = (seg.astype(np.int64) * (1 << np.arange(seg.shape[2]))).sum(2)       # one int per powerset class
    vals, cnt = np.unique(code, return_counts=True)
    frac = cnt / code.size
    trans = (code[:, 1:] != code[:, :-1]).sum(1)
    return {"classes": int(len(vals)), "classes_ge_5pct": int((frac >= 0.05).sum()),
            "min_transitions_per_window": int(trans.min()), "mean_transitions_per_window": float(trans.mean()),
            "overlap_frac": float((seg.sum(2) >= 2).mean()), "silence_frac": float((seg.sum(2) == 0).mean())}


E2E_VBX = {"ahc_threshold": 0.1, "Fa": 0.07, "Fb": 0.8, "lda_dim": 128, "max_iters": 20}


def gen_e2e():
    """example/EN2002a_30s.wav through the oracle device stage (reference execution order) with the
    seeded TURN-TAKING weights (testkit/weights.py:turn_taking_state_dict — plain random weights emit one
    class for every frame) -> per-window decisions + embeddings; then the host stage: the REFERENCE's own
    clustering module on those outputs + the loop-for-loop restatement oracle/host_stage.py -> golden RTTM.
    The wav itself (a data fixture of the reference, not source) sits next to the goldens so the GPU box
    can read it."""
    import copy
    import shutil
    from oracle.wav import first_channel_pcm16 as first_channel_16k
    from testkit.weights import emb_state_dict, turn_taking_state_dict
    from oracle import host_stage
    from oracle.pipeline import device_stage_reference
    _ref_path()
    src = REF / "example" / "EN2002a_30s.wav"
    dst = GOLD / "EN2002a_30s.wav"
    if not dst.exists():
        shutil.copyfile(src, dst)
    wave = torch.from_numpy(first_channel_16k(str(dst)))
    cfg = get_seg_config("wavlm_large_s80_md")
    sd = turn_taking_state_dict(cfg, 0)
    esd = emb_state_dict(0)
    seg, emb = device_stage_reference(wave, cfg, sd, esd, duration=8.0, verbose=True)
    st = decision_stats(seg)
    n_clean, n_fallback = mask_branches(seg, 128000)
    print("e2e decisions:", st, "mask branches clean/fallback:", n_clean, n_fallback)
    assert st["classes_ge_5pct"] >= 6 and st["min_transitions_per_window"] >= 5, "degenerate fixture"
    assert n_clean > 0 and n_fallback > 0, "both mask branches must occur in the fixture"
    # host stage: reference clustering (PA/pipelines/clustering.py) + restated loops
    cl = load_reference_clustering()
    clu = E2E_CONFIG["clustering"]["args"]
    ahc = cl.AgglomerativeClustering(metric="cosine")
    ahc.method, ahc.threshold, ahc.min_cluster_size = "centroid", clu["ahc_threshold"], clu["min_cluster_size"]
    swf = types.SimpleNamespace(data=seg.astype(np.float32))
    hard, _, _ = ahc(embeddings=emb.copy(), segmentations=swf, min_clusters=clu["min_speakers"],
                     max_clusters=clu["max_speakers"])
    # the seeded ResNet's embeddings carry little speaker structure (pairwise distances 0.1-0.2), hence the
    # low ahc_threshold of E2E_CONFIG; the fixture must not sit on a clustering near-tie: 100x the GPU's
    # embedding error must leave the hard clusters unchanged
    r = np.random.default_rng(0)
    for _ in range(8):
        pert = (emb * (1 + 1e-4 * r.normal(size=emb.shape))).astype(np.float32)
        h2, _, _ = ahc(embeddings=pert, segmentations=swf, min_clusters=clu["min_speakers"],
                       max_clusters=clu["max_speakers"])
        assert np.array_equal(h2, hard), "clustering of the fixture is not robust to 1e-4 perturbations"
    rttm = host_stage.host_stage(seg, hard, 8.0, 0.1, clu["max_speakers"], "EN2002a")
    speakers = sorted({ln.split()[7] for ln in rttm.splitlines()})
    print("e2e:", seg.shape, emb.shape, "clusters:", int(hard.max()) + 1, "RTTM speakers:", speakers,
          "lines:", len(rttm.splitlines()))
    assert len(speakers) >= 3, "fixture must have >= 3 speakers in the RTTM"
    # same outputs through the reference VBxClustering (diarizen/clustering/VBx.py) with the seeded PLDA model of
    # host_clustering.npz -> second golden RTTM (the from_pretrained / VBx test)
    import tempfile
    hc = np.load(GOLD / "host_clustering.npz")
    with tempfile.TemporaryDirectory() as td:
        for f in ("xvec_transform", "plda"):
            open(f"{td}/{f}.npz", "wb").write(hc["plda_" + f].tobytes())
        vb = cl.VBxClustering(metric="cosine", plda_dir=td, lda_dim=128, maxIters=20)
        vb.ahc_criterion, vb.ahc_threshold, vb.Fa, vb.Fb = "distance", E2E_VBX["ahc_threshold"], E2E_VBX["Fa"], E2E_VBX["Fb"]
        hard_vbx, _, _ = vb(embeddings=emb.copy(), segmentations=swf)
    rttm_vbx = host_stage.host_stage(seg, hard_vbx, 8.0, 0.1, clu["max_speakers"], "EN2002a")
    print("e2e VBx: clusters", int(hard_vbx.max()) + 1, "lines", len(rttm_vbx.splitlines()))
    (GOLD / "e2e_EN2002a_30s_vbx.rttm").write_text(rttm_vbx)
    np.savez_compressed(GOLD / "e2e_EN2002a_30s.npz", seg=seg, emb=emb, hard_clusters=hard, hard_clusters_vbx=hard_vbx,
                        weight_seed=0,
                        weights="turn_taking", stats=np.array(repr(st)), mask_branches=np.array([n_clean, n_fallback]))
    (GOLD / "e2e_EN2002a_30s.rttm").write_text(rttm)



# ------------------------------------------------------------------ host stage by the REFERENCE's own functions
HOST_REF_CASES = [
    # name, seed, C, L, S, window duration (s), step ratio, clusters, max_speakers
    ("w2s_c60", 21, 60, 99, 4, 2.0, 0.1, 3, 20),
    ("w2s_c200", 22, 200, 99, 4, 2.0, 0.1, 6, 20),
    ("w5s_c40_step50", 23, 40, 249, 4, 5.0, 0.5, 4, 20),
    ("w8s_c1", 24, 1, 399, 4, 8.0, 0.1, 2, 20),
    ("w8s_c7_cap1", 25, 7, 399, 4, 8.0, 0.1, 3, 1),          # count capped by max_speakers (inference.py:163)
    ("w2s_c30_pad", 26, 30, 99, 4, 2.0, 0.1, 1, 20),         # fewer clusters than the frame count: np.pad branch
    ("w2s_c25_s3", 27, 25, 99, 3, 2.0, 0.25, 5, 20),
]


def synth_decisions(seed, C, L, S, K):
    """hard decisions with overlap, silence and inactive local speakers + a cluster per (window, local speaker),
    several local speakers of one window sometimes sharing a cluster (the max branch of reconstruct)"""
    g = np.random.default_rng(seed)
    seg = np.zeros((C, L, S), dtype=np.uint8)
    for c in range(C):
        for s in range(S):
            if g.random() < 0.3:
                continue                                 # inactive local speaker
            for _ in range(g.integers(1, 4)):
                a = int(g.integers(0, L - 3))
                b = int(min(L, a + g.integers(2, max(3, L // 2))))
                seg[c, a:b, s] = 1
    hard = g.integers(0, K, size=(C, S)).astype(np.int8)
    return seg, hard


def gen_host_ref():
    """tests/golden/host_ref.npz: the reference's OWN aggregate / speaker_count / reconstruct / to_diarization / Binarize
    (oracle/ref_host.py imports them by path) on seeded decisions -> count, discrete diarization, RTTM text; plus
    Inference.aggregate with hamming / warm-up / NaN.  oracle/host_stage.py and the product's run_host_stage must both
    reproduce every array and every RTTM byte (tests/test_host.py)."""
    from oracle import ref_host
    out = {}
    g = np.load(GOLD / "e2e_EN2002a_30s.npz")
    for tag, hard in (("", g["hard_clusters"]), ("_vbx", g["hard_clusters_vbx"])):
        rttm = ref_host.host_stage(g["seg"], hard, 8.0, 0.1, 20, "EN2002a")
        assert rttm == (GOLD / f"e2e_EN2002a_30s{tag}.rttm").read_text(), "reference host stage != committed e2e golden"
    names = []
    for name, seed, C, L, S, dur, ratio, K, max_spk in HOST_REF_CASES:
        seg, hard = synth_decisions(seed, C, L, S, K)
        rttm, parts = ref_host.host_stage(seg, hard, dur, ratio, max_spk, name, return_parts=True)
        out[f"{name}_seg"], out[f"{name}_hard"] = seg, hard
        out[f"{name}_args"] = np.array([dur, ratio, max_spk], dtype=np.float64)
        out[f"{name}_count"], out[f"{name}_binary"] = parts["count"], parts["binary"].astype(np.uint8)
        out[f"{name}_activations"] = parts["activations"]
        out[f"{name}_frames"] = parts["frames"]
        out[f"{name}_rttm"] = np.frombuffer(rttm.encode(), dtype=np.uint8)
        names.append(name)
        print(f"host_ref {name}: frames {parts['binary'].shape}, max count {int(parts['count'].max())}, "
              f"{len(rttm.splitlines())} RTTM lines")
    # Inference.aggregate on soft scores: hamming window, warm-up, missing (NaN) chunks
    r = np.random.default_rng(31)
    for i, (C, L, K, dur, step, ham, wu, skip) in enumerate([(12, 99, 3, 2.0, 0.2, True, (0.0, 0.0), False),
                                                               (9, 249, 2, 5.0, 1.0, False, (0.5, 0.25), False),
                                                               (15, 99, 4, 2.0, 0.4, True, (0.2, 0.2), True)]):
        sc = r.random((C, L, K)).astype(np.float32)
        sc[r.random((C, 1, K)).repeat(L, 1) < 0.2] = np.nan
        data, fr = ref_host.aggregate(sc, 0.0, dur, step, hamming=ham, warm_up=wu, skip_average=skip, missing=np.nan)
        out[f"agg{i}_scores"], out[f"agg{i}_out"] = sc, data
        out[f"agg{i}_args"] = np.array([dur, step, float(ham), wu[0], wu[1], float(skip)], dtype=np.float64)
        out[f"agg{i}_frames"] = np.array(fr)
    out["cases"] = np.array(names)
    np.savez_compressed(GOLD / "host_ref.npz", **out)



# ------------------------------------------------------------------ clustering: the branches host_clustering.npz does not reach
FORCED_CASES = [
    # name, seed, C, n_spk, threshold, min_cluster_size, call kwargs, max_num_embeddings, random.seed, constrained
    ("min_walk", 41, 90, 3, 0.6, 5, dict(min_clusters=5, max_clusters=20), np.inf, None, True),      # n_large < min_clusters
    ("min_walk_found", 50, 120, 6, 1.3, 5, dict(min_clusters=6, max_clusters=20), np.inf, None, True),  # walk finds an exact fit
    ("max_walk", 42, 120, 5, 0.5, 5, dict(min_clusters=1, max_clusters=2), np.inf, None, True),      # n_large > max_clusters
    ("num_exact", 43, 100, 4, 0.6, 4, dict(num_clusters=6), np.inf, None, True),                     # num_clusters given
    ("num_fewer", 44, 100, 4, 0.6, 4, dict(num_clusters=2), np.inf, None, True),
    ("num_unreachable", 45, 40, 3, 0.6, 13, dict(num_clusters=9), np.inf, None, True),               # best-candidate re-apply
    ("subsample", 46, 150, 4, 0.6, 6, dict(min_clusters=1, max_clusters=20), 120, 1234, True),       # max_num_embeddings
    ("subsample_forced", 47, 150, 4, 0.6, 6, dict(num_clusters=3), 90, 99, True),
    ("argmax_assign", 48, 80, 3, 0.6, 5, dict(min_clusters=1, max_clusters=20), np.inf, None, False),  # unconstrained
    ("one_cluster", 49, 30, 2, 0.6, 5, dict(num_clusters=1), np.inf, None, True),                    # max_clusters < 2
]


def gen_host_forced():
    """tests/golden/host_clustering_forced.npz: the reference's AgglomerativeClustering (PA/pipelines/clustering.py) on
    the branches the plain fixtures never take: the min / max / num_clusters dendrogram walk (:429-481), its
    best-candidate re-application, `max_num_embeddings` sub-sampling (:160-166, seeded `random`), unconstrained argmax
    assignment and the single-cluster shortcut (:296-302).  (`num_large_clusters == 0`, :483-485, cannot be reached:
    min_clusters >= 1 always starts the walk, whose last merge holds every embedding.)"""
    import random
    cl = load_reference_clustering()
    out = {"cases": np.array([c[0] for c in FORCED_CASES])}
    for name, seed, C, nspk, thr, mcs, kw, mne, rseed, constrained in FORCED_CASES:
        seg, emb = synth_host_case(seed, C=C, n_spk=nspk)
        ahc = cl.AgglomerativeClustering(metric="cosine", max_num_embeddings=mne, constrained_assignment=constrained)
        ahc.method, ahc.threshold, ahc.min_cluster_size = "centroid", thr, mcs
        if rseed is not None:
            random.seed(rseed)
        hard, soft, cent = ahc(embeddings=emb.copy(), segmentations=types.SimpleNamespace(data=seg), **kw)
        out[f"{name}_hard"], out[f"{name}_centroids"] = hard, cent
        print(f"host_forced {name}: clusters {int(hard.max()) + 1}, centroids {cent.shape}")
    np.savez_compressed(GOLD / "host_clustering_forced.npz", **out)



# ------------------------------------------------------------------ the reference's architecture tables
def gen_configs():
    """tests/golden/wavlm_configs.json: `get_config(name)` of diarizen/models/module/wavlm_config.py, verbatim, for its four
    names — the source of oracle/configs.py (the product's diarizen_amd/configs.py is checked against it in tests)."""
    import json
    _ref_path()
    from diarizen.models.module.wavlm_config import get_config
    out = {n: get_config(n) for n in ("wavlm_base", "wavlm_large", "wavlm_base_s80_md", "wavlm_large_s80_md")}
    (GOLD / "wavlm_configs.json").write_text(json.dumps(out, indent=1, sort_keys=True) + "\n")
    print("wavlm_configs:", {n: len(v["encoder_ff_interm_features"]) for n, v in out.items()})


# ------------------------------------------------------------------ row f4: dense model, checkpoint-embedded config
def reference_model_from_wavlm_checkpoint(ckpt_path: str, cfg, full_sd):
    """The reference's load_wavlm FILE branch (diarizen/models/eend/model_wavlm_conformer.py:209-221) followed by what
    Model.from_pretrained does with the hub's pytorch_model.bin (PA/core/model.py:360-369): the WavLM is built from the
    checkpoint's own "config" with `wavlm_model(**ckpt["config"])`, initialised from its "state_dict" with strict=False,
    then the FULL model state_dict is loaded over it.  Returns a forward(wave) -> log-probs."""
    _ref_path()
    from diarizen.models.module.wav2vec2.model import wav2vec2_model
    from diarizen.models.module.conformer import ConformerEncoder
    import torch.nn as nn
    ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=False)
    if "config" not in ckpt or "state_dict" not in ckpt:
        raise ValueError("Checkpoint must contain 'config' and 'state_dict'.")
    for k, v in ckpt["config"].items():
        if "prune" in k and v is not False:
            raise ValueError(f"Pruning must be disabled. Found: {k}={v}")
    wavlm = wav2vec2_model(**ckpt["config"])
    wavlm.load_state_dict(ckpt["state_dict"], strict=False)
    conformer = ConformerEncoder(attention_in=cfg.attention_in, ffn_hidden=cfg.ffn_hidden, num_head=cfg.conf_heads,
                                 num_layer=cfg.conf_layers, kernel_size=cfg.conf_kernel, dropout=0.1, use_posi=False,
                                 output_activate_function=False)
    weight_sum = nn.Linear(cfg.wavlm_layer_num, 1, bias=False)
    proj = nn.Linear(cfg.embed_dim, cfg.attention_in)
    lnorm = nn.LayerNorm(cfg.attention_in)
    classifier = nn.Linear(cfg.attention_in, cfg.n_classes)

    def sub(prefix):
        return {k[len(prefix):]: v for k, v in full_sd.items() if k.startswith(prefix)}
    for m, pre in ((wavlm, "wavlm_model."), (conformer, "conformer."), (weight_sum, "weight_sum."), (proj, "proj."),
                   (lnorm, "lnorm."), (classifier, "classifier.")):
        m.load_state_dict(sub(pre), strict=True)
        m.eval()

    @torch.inference_mode()
    def forward(wave):
        reps, _ = wavlm.extract_features(wave)
        feat = torch.squeeze(weight_sum(torch.stack(reps, dim=-1)), -1)
        return torch.log_softmax(classifier(conformer(lnorm(proj(feat)))), dim=-1)
    return forward


def tiny_wavlm_kwargs(cfg):
    """`wav2vec2_model(**kwargs)` of a tiny custom architecture, in the format a WavLM checkpoint stores under "config" """
    _ref_path()
    from diarizen.models.module.wavlm_config import get_config
    rc = dict(get_config("wavlm_large" if cfg.extractor_layer_norm else "wavlm_base"))
    rc["extractor_conv_layer_config"] = [[c, k, s] for c, k, s in zip(cfg.conv_channels, cfg.conv_kernels, cfg.conv_strides)]
    rc["encoder_embed_dim"] = cfg.embed_dim
    rc["encoder_pos_conv_kernel"] = cfg.pos_conv_kernel
    rc["encoder_pos_conv_groups"] = cfg.pos_conv_groups
    rc["encoder_num_layers"] = cfg.n_layers
    rc["encoder_use_attention"] = [bool(u) for u in cfg.use_attention]
    rc["encoder_use_feed_forward"] = [True] * cfg.n_layers
    rc["encoder_total_num_heads"] = [cfg.total_heads] * cfg.n_layers
    rc["encoder_remaining_heads"] = [list(h) for h in cfg.remaining_heads]
    rc["encoder_ff_interm_features"] = list(cfg.ffn_dims)
    return rc


def gen_f4():
    """Row f4.  (1) seg_ckpt_tiny_ln.npz / seg_ckpt_tiny_gn.npz: a {"config", "state_dict"} WavLM checkpoint file (the
    architecture ONLY in the file, no name the tables know) through the reference's load_wavlm file branch -> log-probs;
    the fixture carries the config as JSON so the GPU test can rebuild the same file without /root/reference.
    (2) seg_wavlm_large.npz: the DENSE wavlm_large (24 layers x 16 heads, FFN 4096: 315 M parameters,
    wavlm_config.py:76-112) through the reference modules on 1 s of audio."""
    import json
    import tempfile
    for name, wseed, xseed in (("tiny_ln", 3, 41), ("tiny_gn", 4, 42)):
        cfg = get_seg_config(name)
        sd = seg_model.seg_state_dict(cfg, wseed)
        rc = tiny_wavlm_kwargs(cfg)
        wav_sd = {k[len("wavlm_model."):]: v for k, v in sd.items() if k.startswith("wavlm_model.")}
        with tempfile.TemporaryDirectory() as td:
            path = str(Path(td) / "wavlm_custom.pt")
            # the checkpoint's own state_dict is a DIFFERENT initialisation (seed + 100): the full model state_dict that
            # from_pretrained loads afterwards must win
            other = seg_model.seg_state_dict(cfg, wseed + 100)
            torch.save({"config": rc, "state_dict": {k[len("wavlm_model."):]: v for k, v in other.items()
                                                      if k.startswith("wavlm_model.")}}, path)
            fwd = reference_model_from_wavlm_checkpoint(path, cfg, sd)
            wave = synth_wave(2, 8000, xseed)
            logp = fwd(wave)
        chk, _ = build_reference_seg(cfg, sd)(wave)      # the by-name construction of the same architecture agrees
        assert (chk - logp).abs().max().item() == 0.0
        np.savez_compressed(GOLD / f"seg_ckpt_{name}.npz", B=2, N=8000, weight_seed=wseed, wave_seed=xseed,
                            config_json=np.array(json.dumps(rc)), logp=logp.numpy(),
                            n_wavlm_keys=len(wav_sd))
        print(f"seg_ckpt_{name}: logp {tuple(logp.shape)}")
    cfg = get_seg_config("wavlm_large")
    sd = seg_model.seg_state_dict(cfg, 0)
    fwd = build_reference_seg(cfg, sd)
    wave = synth_wave(1, 16000, 15)
    logp, reps = fwd(wave)
    np.savez_compressed(GOLD / "seg_wavlm_large.npz", B=1, N=16000, weight_seed=0, wave_seed=15, logp=logp.numpy(),
                        rep0=reps[0].numpy(), rep_last=reps[-1].numpy())
    print(f"seg_wavlm_large (dense, {sum(v.numel() for v in sd.values()) / 1e6:.0f} M parameters): logp {tuple(logp.shape)}")


# ------------------------------------------------------------------ centroid linkage at scale (row f1)
def linkage_scale_case(n: int = 30011, dim: int = 48, K: int = 9, seed: int = 23) -> np.ndarray:
    """speaker-structured float32 embeddings built from ELEMENTWISE float32 operations on seeded draws only (no sums, no
    BLAS): the same bits on any host running this image, so a dendrogram computed here can be compared elsewhere."""
    r = np.random.default_rng(seed)
    cent = r.standard_normal((K, dim), dtype=np.float32)
    lab = r.integers(0, K, n)
    noise = r.standard_normal((n, dim), dtype=np.float32)
    return (cent[lab] * np.float32(0.25) + noise * np.float32(0.11)).astype(np.float32)


def gen_linkage_scale():
    """tests/golden/linkage_30k.npz: scipy.cluster.hierarchy.linkage(method="centroid", metric="euclidean") — the call the
    reference makes (PA/pipelines/clustering.py:407-416) — on linkage_scale_case(): n = 30 011 (the
    device linkage at n >= 30 000; scipy needs ~4 min and 3.6 GB for it, so the dendrogram is committed, 0.4 MB)."""
    import hashlib
    from scipy.cluster.hierarchy import linkage
    e = linkage_scale_case()
    Z = linkage(e, method="centroid", metric="euclidean")
    np.savez_compressed(GOLD / "linkage_30k.npz", ids=Z[:, :2].astype(np.int32), size=Z[:, 3].astype(np.int32), dist=Z[:, 2],
                        emb_md5=np.array(hashlib.md5(e.tobytes()).hexdigest()))
    print("linkage_30k:", Z.shape, "last merges", Z[-3:, 2])


# ------------------------------------------------------------------ host stage AT SCALE by the reference's own code (r5)
HOST30_CLUSTERING = {"method": "AgglomerativeClustering", "min_speakers": 1, "max_speakers": 20, "ahc_criterion": "distance",
                     "ahc_threshold": 0.1, "min_cluster_size": 13}      # bench.py:pipeline_conf — BASELINE configs[2]'s host stage


def gen_host30(src: str = "gpurun_out/host30.npz"):
    """tests/golden/host30.npz: what left the DEVICE for the bench's 30-min synthetic recording (seed 3407, seeded turn-taking
    weights: scripts/dump_device_outputs.py on the GPU box -> u8 decisions [2241, 399, 4], f32 embeddings [2241, 4, 256]) and what
    the REFERENCE's own host code makes of it: `AgglomerativeClustering.__call__` (PA/pipelines/clustering.py:175-245, 363-513 —
    filter_embeddings, centroid linkage, fcluster, small-cluster reassignment, centroids, constrained assignment) -> hard_clusters,
    then speaker_count / reconstruct / to_diarization / Binarize (oracle/ref_host.py) -> RTTM.  8964 (window, speaker) rows:
    past the product's device-linkage (2048 embeddings), device-cdist (8192 rows) and vectorised-assignment (> 64 windows)
    switches, so tests/test_host_ref.py (scipy backends, CPU) and tests/test_pipeline_gpu.py (device backends) hold ONE run of
    the product's run_host_stage to one run of the reference's code on the same arrays."""
    from oracle import ref_host
    g = np.load(ROOT / src)
    seg, emb = g["seg"], g["emb"]
    assert seg.dtype == np.uint8 and emb.dtype == np.float32 and seg.shape[0] == emb.shape[0] >= 2241
    cl = load_reference_clustering()
    clu = HOST30_CLUSTERING
    ahc = cl.AgglomerativeClustering(metric="cosine")
    ahc.method, ahc.threshold, ahc.min_cluster_size = "centroid", clu["ahc_threshold"], clu["min_cluster_size"]
    swf = types.SimpleNamespace(data=seg.astype(np.float32))
    import time
    t0 = time.perf_counter()
    hard, _, _ = ahc(embeddings=emb.copy(), segmentations=swf, min_clusters=clu["min_speakers"], max_clusters=clu["max_speakers"])
    t1 = time.perf_counter()
    train, _, _ = ahc.filter_embeddings(emb.copy(), segmentations=swf)
    rttm, parts = ref_host.host_stage(seg, hard, 8.0, 0.1, clu["max_speakers"], "host30", return_parts=True)
    t2 = time.perf_counter()
    # frames whose top-`count` selection cuts through EQUAL activations: there the reference's own result depends on the
    # tie order of np.argsort (PA/pipelines/utils/diarization.py:228-236), which differs between numpy builds / CPU feature
    # sets (AVX-512 sort kernels) — the tests compare everything else exactly and these frames as "any valid selection"
    act, cnt = parts["activations"], parts["count"].reshape(-1).astype(np.int64)
    n = min(len(act), len(cnt))
    srt = -np.sort(-act[:n], axis=1)
    kk = np.clip(cnt[:n], 0, act.shape[1])
    tie = np.zeros(n, dtype=bool)
    inner = (kk > 0) & (kk < act.shape[1])
    idx = np.nonzero(inner)[0]
    tie[idx] = srt[idx, kk[idx] - 1] == srt[idx, kk[idx]]
    # Unlike the 30 s fixture this one is NOT robust to perturbation, and says so: the seeded ResNet's embeddings carry little
    # speaker structure, and among 8964 rows the dendrogram has merges that a 1e-7 relative change of the embeddings reorders —
    # after which ~19 % of the (window, speaker) assignments land in another cluster.  Equality of the product's result with this
    # golden on the SAME bits is therefore a test of arithmetic order (linkage updates, float64 cosine scores, tie handling of
    # the assignment), not only of semantics; it says nothing about different embeddings (that is what the DER tests are for).
    r = np.random.default_rng(0)
    pert = (emb * (1 + 1e-7 * r.normal(size=emb.shape))).astype(np.float32)
    h2, _, _ = ahc(embeddings=pert, segmentations=swf, min_clusters=clu["min_speakers"], max_clusters=clu["max_speakers"])
    moved = int((h2 != hard).sum())
    speakers = sorted({ln.split()[7] for ln in rttm.splitlines()})
    print(f"host30: {seg.shape[0]} windows, {len(train)} training embeddings of {seg.shape[0] * seg.shape[2]} rows, "
          f"{int(hard.max()) + 1} clusters, {len(speakers)} RTTM speakers, {len(rttm.splitlines())} lines; reference clustering "
          f"{t1 - t0:.1f} s, reconstruction + Binarize {t2 - t1:.1f} s; a 1e-7 relative perturbation of the embeddings moves {moved} of "
          f"{hard.size} assignments")
    np.savez_compressed(GOLD / "host30.npz", seg=seg, emb=emb, hard_clusters=hard.astype(np.int8),
                        rttm=np.frombuffer(rttm.encode(), dtype=np.uint8), n_train=len(train), moved_by_1e7_perturbation=moved,
                        count=parts["count"].astype(np.int8), binary=parts["binary"].astype(np.uint8),
                        activations=parts["activations"].astype(np.float32), boundary_tie_frames=np.nonzero(tie)[0].astype(np.int32))
    print(f"host30: {int(tie.sum())} of {n} frames select through a tie at the count boundary")


GENERATORS = {"host30": gen_host30, "linkage_scale": gen_linkage_scale, "configs": gen_configs, "seg": gen_seg, "seg_tt": gen_seg_tt, "emb": gen_emb, "kat": gen_statspool_powerset, "host": gen_host,
              "e2e": gen_e2e, "host_ref": gen_host_ref, "host_forced": gen_host_forced, "f4": gen_f4,
              "seg_outlier": gen_seg_outlier, "seg_loud": gen_seg_loud, "emb_outlier": gen_emb_outlier}

if __name__ == "__main__":
    GOLD.mkdir(parents=True, exist_ok=True)
    todo = sys.argv[1:] or [g for g in GENERATORS if g not in ("linkage_scale", "host30")]   # (4 min of scipy / needs the GPU dump: on request)
    torch.set_num_threads(8)
    for k in todo:
        GENERATORS[k]()
