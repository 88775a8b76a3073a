"""Segmentation forward through the C ABI vs (a) the golden outputs of the reference's own
modules (tests/golden/seg_*.npz, made by oracle/gen_golden.py) and (b) the oracle restatement,
on the same seeded weights + inputs.

Tolerance, fp32 engine ("strict" mode of SURVEY.md §8d): max |d logp| <= 1e-3 and identical
argmax (hence identical hard multilabel) on these fixtures.  bf16 engine: max |d logp| <= 1e-1 (random seeded weights amplify operand rounding),
argmax agreement >= 99.5 %.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
TAPS = ["conv0", "features", "featproj", "rep0", "layer0", "layer1", "layer2", "layer3", "wsum",
        "head_in", "conf0", "conf1"]


def _run_case(name, gpu, precision, want_taps=False):
    from diarizen_amd.configs import get_seg_config
    from diarizen_amd.engine import Engine
    from oracle import seg_model
    from oracle.gen_golden import synth_wave
    cfg = get_seg_config(name)
    g = np.load(os.path.join(GOLD, f"seg_{name}.npz"))
    sd = seg_model.seg_state_dict(cfg, int(g["weight_seed"]))
    wave = synth_wave(int(g["B"]), int(g["N"]), int(g["wave_seed"]))
    if want_taps:
        os.environ["DZN_DEBUG_TAPS"] = "1"
    try:
        eng = Engine(cfg, sd, max_batch=int(g["B"]), max_samples=int(g["N"]), precision=precision,
                     device=gpu)
    finally:
        os.environ.pop("DZN_DEBUG_TAPS", None)
    logp, ml = eng.segment(wave.to(gpu))
    torch.cuda.synchronize()
    return cfg, sd, wave, g, eng, logp.cpu(), ml.cpu()


def _tap_report(cfg, sd, wave, eng):
    from oracle import seg_model
    taps = {}
    seg_model.seg_forward(sd, cfg, wave, taps)
    taps["rep0"] = None
    lines = []
    for k in TAPS:
        if k not in taps or taps[k] is None:
            continue
        try:
            got = eng.debug_fetch(k)
        except Exception as e:  # noqa
            lines.append(f"{k}: <no tap> {e}")
            continue
        ref = taps[k].numpy().reshape(-1)
        if got.size != ref.size:
            lines.append(f"{k}: size {got.size} vs {ref.size}")
            continue
        lines.append(f"{k}: max|d|={np.abs(got - ref).max():.3e} ref max={np.abs(ref).max():.3e}")
    return "\n".join(lines)


# "f32"  = fp32 MFMA (v_mfma_f32_16x16x4_f32); "f32s" = fp32 by exact 3-way bf16 operand split, six
# bf16 MFMA products, fp32 accumulate (csrc/gemm_split.hip).  Same strict tolerance for both.
@pytest.mark.parametrize("precision", ["f32", "f32s", "f32h"])
@pytest.mark.parametrize("name", ["tiny_ln", "tiny_gn", "wavlm_large_s80_md", "wavlm_base_s80_md"])
def test_seg_fp32_matches_reference_golden(built_lib, gpu, name, precision):
    cfg, sd, wave, g, eng, logp, ml = _run_case(name, gpu, precision, want_taps=True)
    ref = torch.from_numpy(g["logp"])
    err = (logp - ref).abs().max().item()
    report = _tap_report(cfg, sd, wave, eng) if err > 1e-3 else ""
    assert err <= 1e-3, f"max |dlogp| = {err}\n{report}"
    assert torch.equal(logp.argmax(-1), ref.argmax(-1))
    # hard multilabel == Powerset.to_multilabel(soft=False) of the reference log-probs
    from oracle import seg_model
    exp_ml = seg_model.to_multilabel(ref, cfg).to(torch.uint8)
    assert torch.equal(ml, exp_ml)
    # unused: BatchNorm.num_batches_tracked, and transformer.layer_norm (never applied in
    # get_intermediate_outputs for pre-norm encoders, W2V/components.py:1004-1024)
    assert eng.num_ignored_keys <= cfg.conf_layers + 2


@pytest.mark.parametrize("precision", ["f32", "f32s", "f32h"])
@pytest.mark.parametrize("name", ["tiny_ln", "tiny_gn", "wavlm_large_s80_md", "wavlm_base_s80_md"])
def test_seg_fp32_matches_reference_golden_turn_taking(built_lib, gpu, name, precision):
    """NON-degenerate goldens (reference modules + seeded turn-taking weights on real audio): many powerset
    classes and ~15 transitions per window, at BASELINE configs[1] size for base-s80 (5 s x 32) and the bench
    geometry for large-s80 (8 s).  Strict bar: max |dlogp| <= 1e-3 AND every argmax / u8 decision identical."""
    from diarizen_amd.configs import get_seg_config
    from diarizen_amd.engine import Engine
    from testkit.weights import turn_taking_state_dict
    from oracle import seg_model
    from oracle.gen_golden import tt_windows
    cfg = get_seg_config(name)
    g = np.load(os.path.join(GOLD, f"seg_tt_{name}.npz"))
    ref = torch.from_numpy(g["logp"])
    assert len(torch.unique(ref.argmax(-1))) >= 5
    wave = tt_windows(g["starts"].tolist(), int(g["N"]))
    B = wave.shape[0]
    eng = Engine(cfg, turn_taking_state_dict(cfg, int(g["weight_seed"])), max_batch=B, max_samples=int(g["N"]),
                 precision=precision, device=gpu)
    logp, ml = eng.segment(wave.to(gpu))
    torch.cuda.synchronize()
    logp, ml = logp.cpu(), ml.cpu()
    err = (logp - ref).abs().max().item()
    top2 = ref.topk(2, dim=-1).values
    print(f"[{name} {precision}] max|dlogp|={err:.2e}  min top-2 margin of the reference={float((top2[..., 0] - top2[..., 1]).min()):.2e}")
    assert err <= 1e-3
    assert torch.equal(logp.argmax(-1), ref.argmax(-1))
    assert torch.equal(ml, seg_model.to_multilabel(ref, cfg).to(torch.uint8))


def _outlier_case(name, gpu, precision):
    from diarizen_amd.configs import get_seg_config
    from diarizen_amd.engine import Engine
    from testkit.weights import outlier_state_dict
    from oracle.gen_golden import tt_windows
    cfg = get_seg_config(name)
    g = np.load(os.path.join(GOLD, f"seg_outlier_{name}.npz"))
    wave = tt_windows(g["starts"].tolist(), int(g["N"]))
    os.environ["DZN_DEBUG_TAPS"] = "1"
    try:
        eng = Engine(cfg, outlier_state_dict(cfg, int(g["weight_seed"])), max_batch=wave.shape[0], max_samples=int(g["N"]),
                     precision=precision, device=gpu)
    finally:
        os.environ.pop("DZN_DEBUG_TAPS", None)
    logp, ml = eng.segment(wave.to(gpu))
    torch.cuda.synchronize()
    return cfg, g, eng, logp.cpu(), ml.cpu()


@pytest.mark.parametrize("precision", ["f32", "f32s", "f32h"])
@pytest.mark.parametrize("name", ["tiny_ln", "tiny_gn", "wavlm_large_s80_md", "wavlm_base_s80_md"])
def test_seg_fp32_matches_reference_golden_planted_outliers(built_lib, gpu, name, precision):
    """Reference-made goldens with PLANTED MASSIVE ACTIVATIONS (testkit/weights.py:outlier_state_dict): four
    residual-stream channels 2^10 ... 2^13 x the typical magnitude from encoder layer 2 to the last layer and in the layer sum
    (pre-norm large: W2V/components.py:920-935 via `output_dense` :805-813; post-norm base: every LayerNorm re-emits them), as
    trained WavLM checkpoints have them.  What they meet here: the per-window power-of-two operand scales of the f32h / f32s
    splits (typical elements sit 2^10 ... 2^13 below the window |max|), the folded-LayerNorm epilogue (rstd * (x W' - mean
    colsum): the row mean is 2-3 sigma, the rstd 2^-9), the row statistics, the deferred layer sum and `proj`.  STRICT bar: max
    |dlogp| <= 1e-3, every argmax and u8 decision identical.  The reference's own fp32-vs-float64 distance on the fixture is
    stored in it (2e-4): the bar leaves 5 x that."""
    from oracle import seg_model
    cfg, g, eng, logp, ml = _outlier_case(name, gpu, precision)
    ref = torch.from_numpy(g["logp"])
    assert float(g["massive_over_typical"]) >= 256 and len(torch.unique(ref.argmax(-1))) >= 5
    # the engine's last encoder layer really carries the massive channels (the fixture is not silently defused on the way in)
    last = eng.debug_fetch(f"layer{cfg.n_layers - 1}").reshape(ref.shape[0], ref.shape[1], cfg.embed_dim)
    got_out = last[..., g["chans"]]
    assert np.abs(got_out - g["rep_last_outlier"]).max() <= 2e-4 * np.abs(g["rep_last_outlier"]).max()
    err = (logp - ref).abs().max().item()
    err64 = (logp.double() - torch.from_numpy(g["logp64"])).abs().max().item()
    print(f"[outlier {name} {precision}] max|dlogp| vs the reference fp32 {err:.2e}, vs its float64 run {err64:.2e} "
          f"(reference fp32 vs float64: {float(g['ref_fp32_vs_fp64']):.2e}); massive / typical = {float(g['massive_over_typical']):.0f}")
    assert err <= 1e-3
    assert torch.equal(logp.argmax(-1), ref.argmax(-1))
    assert torch.equal(ml, seg_model.to_multilabel(ref, cfg).to(torch.uint8))


@pytest.mark.parametrize("name", ["tiny_ln", "tiny_gn", "wavlm_large_s80_md", "wavlm_base_s80_md"])
def test_seg_f16_planted_outliers_meet_the_reduced_bar(built_lib, gpu, name):
    """The reduced mode (fp16 hi*hi + fp8 cross terms, csrc/gemm_mx.hip) on the planted-outlier goldens: its fp8 operands share ONE
    power-of-two scale per window / weight row, so typical elements sit 10-13 binades under the |max| that sets it - e4m3 spans
    17.  Reduced bar of SURVEY 8d: max |dlogp| <= 5e-2, argmax >= 99.5 %."""
    import json
    cfg, g, eng, logp, ml = _outlier_case(name, gpu, "f16")
    ref = torch.from_numpy(g["logp"])
    err = (logp - ref).abs().max().item()
    agree = (logp.argmax(-1) == ref.argmax(-1)).float().mean().item()
    rec = {"fixture": f"seg_outlier_{name}", "frames": int(ref.shape[0] * ref.shape[1]), "max_abs_dlogp": err,
           "argmax_agreement": agree, "massive_over_typical": float(g["massive_over_typical"])}
    os.makedirs("gpurun_out", exist_ok=True)
    path = "gpurun_out/f16_outlier_bar.json"
    allrec = json.load(open(path)) if os.path.exists(path) else {}
    allrec[name] = rec
    json.dump(allrec, open(path, "w"), indent=1)
    print(json.dumps(rec))
    top2 = ref.topk(2, dim=-1).values
    flipped = logp.argmax(-1) != ref.argmax(-1)
    near_tie = (not flipped.any()) or float((top2[..., 0] - top2[..., 1])[flipped].max()) <= 2 * err
    assert err <= 5e-2 and (agree >= 0.995 or (ref.shape[0] * ref.shape[1] < 200 and near_tie)), rec


@pytest.mark.parametrize("precision", ["f32", "f32s", "f32h", "f16"])
@pytest.mark.parametrize("name", ["tiny_ln", "tiny_gn", "wavlm_large_s80_md", "wavlm_base_s80_md"])
def test_seg_loudness_extremes_in_one_batch(built_lib, gpu, name, precision):
    """One batch of four windows: as recorded, near-silent (x 1e-4: under the eps of the waveform LayerNorm and of conv0's
    channel norm), clipped (x 8 clamped to +-1) and DIGITAL SILENCE (all zeros: every |max| tracker of the window reads 0).
    Reference-made golden (oracle/gen_golden.py:gen_seg_loud).  Every operand scale is per window, so (a) each window meets its
    bar next to the others and (b) a window's bits do not depend on its neighbours: the batch equals four single-window calls."""
    from diarizen_amd.configs import get_seg_config
    from diarizen_amd.engine import Engine
    from testkit.weights import turn_taking_state_dict
    from oracle import seg_model
    from oracle.gen_golden import loud_windows
    cfg = get_seg_config(name)
    g = np.load(os.path.join(GOLD, f"seg_loud_{name}.npz"))
    ref = torch.from_numpy(g["logp"])
    wave = loud_windows(int(g["N"]), int(g["start"]))
    eng = Engine(cfg, turn_taking_state_dict(cfg, int(g["weight_seed"])), max_batch=4, max_samples=int(g["N"]),
                 precision=precision, device=gpu)
    logp, ml = eng.segment(wave.to(gpu))
    torch.cuda.synchronize()
    logp, ml = logp.cpu().clone(), ml.cpu().clone()
    assert torch.isfinite(logp).all()
    errs = [(logp[b] - ref[b]).abs().max().item() for b in range(4)]
    print(f"[loud {name} {precision}] max|dlogp| as recorded / x1e-4 / clipped / zeros: " + " ".join(f"{e:.2e}" for e in errs))
    if precision == "f16":
        assert max(errs) <= 5e-2
        agree = (logp.argmax(-1) == ref.argmax(-1)).float().mean().item()
        assert agree >= 0.995 or ref.shape[1] < 50
    else:
        assert max(errs) <= 1e-3
        assert torch.equal(logp.argmax(-1), ref.argmax(-1))
        assert torch.equal(ml, seg_model.to_multilabel(ref, cfg).to(torch.uint8))
    for b in range(4):
        one, one_ml = eng.segment(wave[b:b + 1].to(gpu))
        torch.cuda.synchronize()
        assert torch.equal(one.cpu()[0], logp[b]) and torch.equal(one_ml.cpu()[0], ml[b]), f"window {b} depends on its neighbours"


@pytest.mark.parametrize("name", ["tiny_ln", "wavlm_large_s80_md"])
def test_seg_bf16_within_tolerance(built_lib, gpu, name):
    from conftest import needs_bf16_mode
    needs_bf16_mode(built_lib)
    cfg, sd, wave, g, eng, logp, ml = _run_case(name, gpu, "bf16")
    ref = torch.from_numpy(g["logp"])
    err = (logp - ref).abs().max().item()
    agree = (logp.argmax(-1) == ref.argmax(-1)).float().mean().item()
    assert err <= 1e-1, f"max |dlogp| = {err}"
    assert agree >= 0.995


@pytest.mark.parametrize("name", ["tiny_ln", "wavlm_large_s80_md", "wavlm_base_s80_md"])
def test_seg_f16_within_tolerance(built_lib, gpu, name):
    """DZN_PREC_F16 (BASELINE configs[4] "fp16"): single-term fp16 contractions inside the f32h engine.  Reduced
    precision bar of SURVEY §8(d): max |d logp| <= 5e-2, argmax agreement >= 99.5 % — on the reference-made PLAIN-seed
    goldens (one class wins every frame; the non-degenerate fixtures are the next test)."""
    cfg, sd, wave, g, eng, logp, ml = _run_case(name, gpu, "f16")
    ref = torch.from_numpy(g["logp"])
    err = (logp - ref).abs().max().item()
    agree = (logp.argmax(-1) == ref.argmax(-1)).float().mean().item()
    print(f"[{name} f16] max|dlogp|={err:.2e} argmax agreement={agree:.4f}")
    assert err <= 5e-2, f"max |dlogp| = {err}"
    # these plain goldens are 24-49 frames of ONE winning class with near ties behind it: a single flipped frame is 2-4 %,
    # so the 99.5 % rule is applied as "every flip is a near tie of the reference itself" (top-2 margin <= 2 err)
    top2 = ref.topk(2, dim=-1).values
    flipped = logp.argmax(-1) != ref.argmax(-1)
    assert agree >= 0.995 or float((top2[..., 0] - top2[..., 1])[flipped].max()) <= 2 * err


@pytest.mark.parametrize("name", ["tiny_ln", "wavlm_large_s80_md", "wavlm_base_s80_md"])
def test_seg_f16_meets_the_reduced_bar_on_the_turn_taking_fixtures(built_lib, gpu, name):
    """The reduced mode on the NON-degenerate reference goldens (seg_tt_*: many classes, ~15 transitions per window, top-2 margins
    down to 1e-4; base-s80 at BASELINE configs[1]'s full size, 32 windows).  SURVEY 8d's reduced bar: max |dlogp| <= 5e-2, argmax
    >= 99.5 %.  r2-r4's single-term fp16 mode measured 0.15 / 0.18 here and a test REPORTED that; since r5 every linear
    contraction of the segmentation model keeps its two cross terms in fp8 (csrc/gemm_mx.hip) and this test ASSERTS the bar.
    The measured figures go to gpurun_out/f16_turn_taking_bar.json (committed as part of profiles/r5_reduced_mode_parity.json,
    which bench.py quotes)."""
    import json
    from diarizen_amd.configs import get_seg_config
    from diarizen_amd.engine import Engine
    from testkit.weights import turn_taking_state_dict
    from oracle.gen_golden import tt_windows
    cfg = get_seg_config(name)
    g = np.load(os.path.join(GOLD, f"seg_tt_{name}.npz"))
    ref = torch.from_numpy(g["logp"])
    wave = tt_windows(g["starts"].tolist(), int(g["N"]))
    eng = Engine(cfg, turn_taking_state_dict(cfg, int(g["weight_seed"])), max_batch=wave.shape[0], max_samples=int(g["N"]),
                 precision="f16", device=gpu)
    logp, _ = eng.segment(wave.to(gpu))
    torch.cuda.synchronize()
    logp = logp.cpu()
    err = (logp - ref).abs().max().item()
    agree = (logp.argmax(-1) == ref.argmax(-1)).float().mean().item()
    rec = {"fixture": f"seg_tt_{name}", "frames": int(ref.shape[0] * ref.shape[1]), "max_abs_dlogp": err, "argmax_agreement": agree,
           "bar": {"max_abs_dlogp": 5e-2, "argmax_agreement": 0.995},
           "meets_survey_8d_bar": bool(err <= 5e-2 and agree >= 0.995)}
    os.makedirs("gpurun_out", exist_ok=True)
    path = "gpurun_out/f16_turn_taking_bar.json"
    allrec = json.load(open(path)) if os.path.exists(path) else {}
    allrec[name] = rec
    json.dump(allrec, open(path, "w"), indent=1)
    print(json.dumps(rec))
    assert err <= 5e-2 and agree >= 0.995, rec


def test_seg_f16_single_term_switch_reproduces_the_r4_arithmetic(built_lib, gpu, monkeypatch):
    """DZN_F16_MX=0 (read at dzn_create) switches the cross terms off: the engine then runs r2-r4's single-term contractions,
    which is what keeps the r4 measurements (0.15 on this fixture) reproducible — and shows the cross terms are what meets the bar."""
    from diarizen_amd.configs import get_seg_config
    from diarizen_amd.engine import Engine
    from testkit.weights import turn_taking_state_dict
    from oracle.gen_golden import tt_windows
    name = "wavlm_large_s80_md"
    cfg = get_seg_config(name)
    g = np.load(os.path.join(GOLD, f"seg_tt_{name}.npz"))
    ref = torch.from_numpy(g["logp"])
    wave = tt_windows(g["starts"].tolist(), int(g["N"]))
    monkeypatch.setenv("DZN_F16_MX", "0")
    eng = Engine(cfg, turn_taking_state_dict(cfg, int(g["weight_seed"])), max_batch=wave.shape[0], max_samples=int(g["N"]),
                 precision="f16", device=gpu)
    logp, _ = eng.segment(wave.to(gpu))
    torch.cuda.synchronize()
    err = (logp.cpu() - ref).abs().max().item()
    print(f"[single-term f16, {name}] max |dlogp| = {err:.3f}")
    assert 5e-2 < err <= 0.3


def test_seg_batch_and_ragged_lengths(built_lib, gpu):
    """windows are independent: a batch equals its items run alone; shorter N re-uses the engine."""
    from diarizen_amd.configs import get_seg_config
    from diarizen_amd.engine import Engine
    from oracle import seg_model
    from oracle.gen_golden import synth_wave
    cfg = get_seg_config("tiny_ln")
    sd = seg_model.seg_state_dict(cfg, 0)
    eng = Engine(cfg, sd, max_batch=5, max_samples=12000, precision="f32", device=gpu)
    wave = synth_wave(5, 12000, 3).to(gpu)
    full, _ = eng.segment(wave)
    one, _ = eng.segment(wave[2:3].contiguous())
    torch.cuda.synchronize()
    assert (full[2:3] - one).abs().max().item() < 1e-5
    short = wave[:, :7000].contiguous()
    lp, ml = eng.segment(short)
    torch.cuda.synchronize()
    ref = seg_model.seg_forward(sd, cfg, short.cpu())
    assert lp.shape == ref.shape
    assert (lp.cpu() - ref).abs().max().item() < 1e-3
    with pytest.raises(Exception):
        eng.segment(torch.zeros(6, 12000, device=gpu))  # B > max_batch must fail loudly


@pytest.mark.parametrize("precision", ["f32", "f32s", "f32h"])
def test_seg_16s_window_cli_default(built_lib, gpu, precision):
    """the reference CLI default is 16 s windows (diarizen/pipelines/inference.py:224-228): L = 799"""
    from diarizen_amd.configs import get_seg_config
    from diarizen_amd.engine import Engine
    from oracle import seg_model
    from oracle.gen_golden import synth_wave
    cfg = get_seg_config("wavlm_large_s80_md")
    sd = seg_model.seg_state_dict(cfg, 0)
    wave = synth_wave(1, 256000, 41)
    eng = Engine(cfg, sd, max_batch=1, max_samples=256000, precision=precision, device=gpu)
    logp, ml = eng.segment(wave.to(gpu))
    torch.cuda.synchronize()
    ref = seg_model.seg_forward(sd, cfg, wave)
    assert logp.shape == ref.shape == (1, 799, 11)
    assert (logp.cpu() - ref).abs().max().item() < 1e-3
    assert torch.equal(ml.cpu(), seg_model.to_multilabel(ref, cfg).to(torch.uint8))


@pytest.mark.parametrize("precision", ["f32", "f32s", "f32h"])
def test_seg_dense_wavlm_base(built_lib, gpu, precision):
    """un-pruned wavlm_base (12 x 12 heads, FFN 3072, 512-channel extractor, post-norm, group-norm)"""
    from diarizen_amd.configs import get_seg_config
    from diarizen_amd.engine import Engine
    from oracle import seg_model
    from oracle.gen_golden import synth_wave
    cfg = get_seg_config("wavlm_base")
    sd = seg_model.seg_state_dict(cfg, 3)
    wave = synth_wave(2, 16000, 42)
    eng = Engine(cfg, sd, max_batch=2, max_samples=16000, precision=precision, device=gpu)
    logp, ml = eng.segment(wave.to(gpu))
    torch.cuda.synchronize()
    ref = seg_model.seg_forward(sd, cfg, wave)
    assert (logp.cpu() - ref).abs().max().item() < 1e-3
    assert torch.equal(logp.cpu().argmax(-1), ref.argmax(-1))


@pytest.mark.parametrize("precision", ["f32h", "f32"])
def test_seg_dense_wavlm_large_matches_reference_golden(built_lib, gpu, precision):
    """Row f4: the DENSE wavlm_large (24 layers x 16 heads, FFN 4096, 512-channel extractor: 322 M parameters,
    diarizen/models/module/wavlm_config.py:76-112) against tests/golden/seg_wavlm_large.npz — made by the reference's own
    wav2vec2_model + ConformerEncoder with a strict state_dict load (oracle/gen_golden.py f4).  Strict fp32 bar."""
    cfg, sd, wave, g, eng, logp, ml = _run_case("wavlm_large", gpu, precision)
    assert all(len(h) == 16 for h in cfg.remaining_heads) and set(cfg.ffn_dims) == {4096} and set(cfg.conv_channels) == {512}
    ref = torch.from_numpy(g["logp"])
    err = (logp - ref).abs().max().item()
    print(f"[dense wavlm_large {precision}] max|dlogp|={err:.2e} over {ref.shape[1]} frames")
    assert err <= 1e-3
    assert torch.equal(logp.argmax(-1), ref.argmax(-1))
    eng.close()


@pytest.mark.parametrize("name", ["tiny_ln", "tiny_gn"])
def test_seg_from_a_checkpoint_embedded_config_end_to_end(built_lib, gpu, name, tmp_path):
    """Row f4: `wavlm_src=<file>` — the load_wavlm FILE branch (diarizen/models/eend/model_wavlm_conformer.py:209-221): the
    architecture comes from the "config" entry of a {"config", "state_dict"} WavLM checkpoint (no name the tables know), the
    checkpoint's own weights are loaded strict=False, then the full model state_dict replaces them
    (PA/core/model.py:360-369).  End to end through the plugin class on the GPU, against tests/golden/seg_ckpt_*.npz, which
    the REFERENCE made by exactly those steps (oracle/gen_golden.py:reference_model_from_wavlm_checkpoint).  The file is
    rebuilt here from the config JSON the fixture carries, with a DIFFERENT initialisation in its own state_dict: the full
    state_dict must win."""
    import json
    from diarizen_amd.configs import get_seg_config
    from diarizen_amd.models import WavLMConformer
    from oracle import seg_model
    from oracle.gen_golden import synth_wave
    g = np.load(os.path.join(GOLD, f"seg_ckpt_{name}.npz"))
    head = get_seg_config(name)                    # the Conformer head's sizes only (the WavLM part comes from the file)
    rc = json.loads(str(g["config_json"]))
    seed = int(g["weight_seed"])
    sd = seg_model.seg_state_dict(head, seed)
    other = seg_model.seg_state_dict(head, seed + 100)
    path = str(tmp_path / "wavlm_custom.pt")
    torch.save({"config": rc, "state_dict": {k[len("wavlm_model."):]: v for k, v in other.items()
                                              if k.startswith("wavlm_model.")}}, path)
    model = WavLMConformer(wavlm_src=path, wavlm_layer_num=head.wavlm_layer_num, wavlm_feat_dim=head.embed_dim,
                           attention_in=head.attention_in, ffn_hidden=head.ffn_hidden, num_head=head.conf_heads,
                           num_layer=head.conf_layers, kernel_size=head.conf_kernel, chunk_size=1, max_batch=int(g["B"]))
    assert model.cfg.conv_channels == head.conv_channels and model.cfg.ffn_dims == head.ffn_dims
    assert model.cfg.remaining_heads == head.remaining_heads and model.cfg.name == "wavlm_custom.pt"
    model.load_state_dict(sd).eval().to(gpu)
    wave = synth_wave(int(g["B"]), int(g["N"]), int(g["wave_seed"]))
    logp = model(wave[:, None, :]).cpu()
    ref = torch.from_numpy(g["logp"])
    assert (logp - ref).abs().max().item() <= 1e-3
    assert torch.equal(logp.argmax(-1), ref.argmax(-1))
    # the reference's two refusals (model_wavlm_conformer.py:212-218)
    bad = dict(rc)
    bad["encoder_prune_attention_heads"] = True
    torch.save({"config": bad, "state_dict": {}}, str(tmp_path / "pruned.pt"))
    with pytest.raises(ValueError, match="Pruning must be disabled"):
        WavLMConformer(wavlm_src=str(tmp_path / "pruned.pt"), wavlm_layer_num=head.wavlm_layer_num, wavlm_feat_dim=head.embed_dim)
    torch.save({"state_dict": {}}, str(tmp_path / "noconfig.pt"))
    with pytest.raises(ValueError, match="must contain"):
        WavLMConformer(wavlm_src=str(tmp_path / "noconfig.pt"), wavlm_layer_num=head.wavlm_layer_num, wavlm_feat_dim=head.embed_dim)


@pytest.mark.parametrize("name", ["tiny_gn", "wavlm_base_s80_md"])
def test_seg_bf16_group_norm_models(built_lib, gpu, name):
    """bf16 engine on the base-style models (group-norm extractor, post-norm encoder, 48-channel
    positional-conv groups for base: that contraction stays on fp32 activations)"""
    from conftest import needs_bf16_mode
    needs_bf16_mode(built_lib)
    cfg, sd, wave, g, eng, logp, ml = _run_case(name, gpu, "bf16")
    ref = torch.from_numpy(g["logp"])
    assert (logp - ref).abs().max().item() <= 1e-1
    # decisions may only differ on frames where the reference's own top-2 margin is inside the tolerance
    differ = logp.argmax(-1) != ref.argmax(-1)
    top2 = ref.topk(2, dim=-1).values
    margin = top2[..., 0] - top2[..., 1]
    assert (margin[differ] <= 2e-1).all()
    assert differ.float().mean().item() <= 0.03


def test_conv01_fusion_matches_unfused(built_lib, gpu, monkeypatch):
    """frontend_fused.hip (f32h): conv0 + LN + GELU + conv1 in one kernel (conv0's activations never reach HBM) against
    the same engine with the fusion switched off — both run through the oracle-checked f32h path; windows of different
    lengths exercise the ragged last tile (T1 % 128 != 0) and conv0 frames past the end of the strip."""
    from diarizen_amd.configs import get_seg_config
    from diarizen_amd.engine import Engine
    from testkit.weights import turn_taking_state_dict
    from oracle import seg_model
    from oracle.gen_golden import tt_windows
    cfg = get_seg_config("wavlm_large_s80_md")
    sd = turn_taking_state_dict(cfg, 0)
    for N in (128000, 33333):
        wave = tt_windows([0, 100000, 250000], N)
        fused = Engine(cfg, sd, max_batch=3, max_samples=N, precision="f32h", device=gpu)
        monkeypatch.setenv("DZN_NO_CONV01_FUSION", "1")
        plain = Engine(cfg, sd, max_batch=3, max_samples=N, precision="f32h", device=gpu)
        monkeypatch.delenv("DZN_NO_CONV01_FUSION")
        lf, mf = fused.segment(wave.to(gpu))
        lp_, mp = plain.segment(wave.to(gpu))
        # (r3) the fused kernel also finishes conv1's LayerNorm + GELU in its epilogue; DZN_CONV01_NO_LN=1 (read per call)
        # sends the raw conv1 output through the stand-alone row pass instead
        monkeypatch.setenv("DZN_CONV01_NO_LN", "1")
        l2, m2 = fused.segment(wave.to(gpu))
        monkeypatch.delenv("DZN_CONV01_NO_LN")
        # (r6) the default is the producer / consumer kernel (conv01_ws_kernel: 127-frame tiles, one lane per conv0 frame, packed
        # fp32, pairwise LayerNorm statistics in the epilogue); DZN_CONV01_WS=0 (read per call) runs the phase-alternating kernel it
        # replaced, with and without the LayerNorm epilogue
        monkeypatch.setenv("DZN_CONV01_WS", "0")
        l3, m3 = fused.segment(wave.to(gpu))
        monkeypatch.setenv("DZN_CONV01_NO_LN", "1")
        l4, m4 = fused.segment(wave.to(gpu))
        monkeypatch.delenv("DZN_CONV01_NO_LN")
        monkeypatch.delenv("DZN_CONV01_WS")
        torch.cuda.synchronize()
        d = (lf - lp_).abs().max().item()
        d2 = (lf - l2).abs().max().item()
        d3 = max((lf - l3).abs().max().item(), (l2 - l4).abs().max().item())
        print(f"N={N}: max |logp fused - unfused| = {d:.2e}; LN in the epilogue vs stand-alone = {d2:.2e}; "
              f"producer / consumer kernel vs the phase-alternating one = {d3:.2e}")
        assert d <= 2e-4 and d2 <= 2e-4 and d3 <= 2e-4
        assert torch.equal(mf, mp) and torch.equal(mf, m2) and torch.equal(mf, m3) and torch.equal(mf, m4)
        if N == 33333:
            ref = seg_model.seg_forward(sd, cfg, wave)
            assert (lf.cpu() - ref).abs().max().item() <= 1e-3
        fused.close()
        plain.close()


def test_conv0_layernorm_statistics_survive_band_pass_taps_on_low_frequency_audio(built_lib, gpu, monkeypatch):
    """conv0's LayerNorm statistics come from the frame's 10 input samples (mean = wbar . x, var = x^T Q x with
    Q = cov_c(w_c)).  With TRAINED-looking taps — every channel a zero-DC band-pass filter — and low-frequency / quiet audio
    the frame lies near Q's small-eigenvalue directions; a plain sum of products cancels there.  The kernels evaluate
    |F x|^2 with Q = F^T F factored in double at load time (engine.cpp): compare with the two-pass statistics over the C0
    outputs themselves (DZN_CONV0_WAVE_STATS=1) and with the oracle."""
    from diarizen_amd.configs import get_seg_config
    from diarizen_amd.engine import Engine
    from oracle import seg_model
    cfg = get_seg_config("tiny_ln")
    sd = dict(seg_model.seg_state_dict(cfg, 3))
    key = "wavlm_model.feature_extractor.conv_layers.0.conv.weight"
    C0, k = sd[key].shape[0], sd[key].shape[-1]
    g = torch.Generator().manual_seed(5)
    t = torch.arange(k, dtype=torch.float32)
    taps = []
    for c in range(C0):      # windowed sinusoids between 1.5 and 7 kHz, DC removed exactly
        f = 1500.0 + 5500.0 * torch.rand(1, generator=g).item()
        w = torch.hann_window(k + 2, periodic=False)[1:-1] * torch.cos(2 * np.pi * f / 16000.0 * t + 6.28 * torch.rand(1, generator=g).item())
        taps.append(w - w.mean())
    sd[key] = (torch.stack(taps) * 0.5).reshape(sd[key].shape).contiguous()
    N = 16000
    n = torch.arange(N) / 16000.0
    wave = torch.stack([0.5 * torch.sin(2 * np.pi * 60.0 * n) + 1e-4 * torch.randn(N, generator=g),      # loud hum, faint noise
                        1e-3 * torch.sin(2 * np.pi * 35.0 * n + 1.0) + 1e-6 * torch.randn(N, generator=g),   # quiet, low frequency
                        0.2 * torch.sin(2 * np.pi * 110.0 * n) * (n > 0.5)])                                   # half silence
    ref = seg_model.seg_forward(sd, cfg, wave)
    outs = {}
    for mode in ("quadratic", "two_pass"):
        if mode == "two_pass":
            monkeypatch.setenv("DZN_CONV0_WAVE_STATS", "1")
        eng = Engine(cfg, sd, max_batch=3, max_samples=N, precision="f32", device=gpu)
        monkeypatch.delenv("DZN_CONV0_WAVE_STATS", raising=False)
        logp, ml = eng.segment(wave.to(gpu))
        torch.cuda.synchronize()
        outs[mode] = logp.cpu()
        eng.close()
    dq = (outs["quadratic"] - ref).abs().max().item()
    dt = (outs["two_pass"] - ref).abs().max().item()
    print(f"max |dlogp| vs oracle: factored quadratic form {dq:.2e}, two-pass {dt:.2e}")
    assert dq <= 1e-3 and dt <= 1e-3
    assert dq <= 3.0 * dt + 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["f32h", "f32"])
def test_deferred_layer_weighted_sum_gives_the_same_bits(built_lib, gpu, monkeypatch, precision):
    """(r4) pre-norm encoders keep every layer's output rows in the layer's own buffer and form the layer-weighted sum
    (model_wavlm_conformer.py:236,253-254) in ONE pass after the last layer (frontend.hip:ws_sum_kernel), instead of a
    read-modify-write of the sum in every FFN-output epilogue.  Same additions in the same order: the log-probabilities must be
    the bits of the engine with DZN_NO_WS_DEFER=1 (read at dzn_create) — on the pruned large model (layers without attention /
    without FFN) and on a dense tiny one, ragged batch included."""
    from diarizen_amd.configs import get_seg_config
    from diarizen_amd.engine import Engine
    from testkit.weights import turn_taking_state_dict
    from oracle.gen_golden import tt_windows
    for name, N in (("wavlm_large_s80_md", 40000), ("tiny_ln", 16000)):
        cfg = get_seg_config(name)
        sd = turn_taking_state_dict(cfg, 0)
        wave = tt_windows([0, 100000, 250000], N).to(gpu)
        deferred = Engine(cfg, sd, max_batch=3, max_samples=N, precision=precision, device=gpu)
        monkeypatch.setenv("DZN_NO_WS_DEFER", "1")
        fused = Engine(cfg, sd, max_batch=3, max_samples=N, precision=precision, device=gpu)
        monkeypatch.delenv("DZN_NO_WS_DEFER")
        for w in (wave, wave[:2, : N - 4321].contiguous()):
            a, ma = deferred.segment(w)
            b, mb = fused.segment(w)
            torch.cuda.synchronize()
            assert torch.equal(a, b) and torch.equal(ma, mb), (name, (a - b).abs().max().item())
        deferred.close()
        fused.close()
