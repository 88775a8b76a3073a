"""Embedding forward (kaldi fbank -> ResNet34 -> multi-mask TSTP -> seg_1) through the C ABI vs
the reference's own ResNet34/StatsPool outputs (tests/golden/emb_resnet.npz) and the oracle.

Tolerance (fp32 engine): fbank max |d| <= 2e-3 log-mel units (fp32 direct-DFT contraction vs fp32
FFT; the absolute level is ~20), embedding cosine >= 0.9999 and max |d| <= 1e-4 * max|emb|; an all-zero
mask reproduces seg_1.bias exactly (pyannote-audio/tests/test_stats_pool.py:111-131).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _engine(gpu, B, N, precision="f32", taps=False):
    from diarizen_amd.configs import RESNET34, get_seg_config
    from diarizen_amd.engine import Engine
    from oracle import emb_model, seg_model
    cfg = get_seg_config("tiny_ln")
    if taps:
        os.environ["DZN_DEBUG_TAPS"] = "1"
    try:
        return Engine(cfg, seg_model.seg_state_dict(cfg, 0), RESNET34, emb_model.emb_state_dict(0),
                      max_batch=B, max_samples=N, precision=precision, device=gpu)
    finally:
        os.environ.pop("DZN_DEBUG_TAPS", None)


@pytest.mark.parametrize("precision", ["f32", "f32s", "f32h"])
def test_embedding_matches_reference_golden(built_lib, gpu, precision):
    from oracle import emb_model
    from oracle.gen_golden import synth_wave
    g = np.load(os.path.join(GOLD, "emb_resnet.npz"))
    B, N = int(g["B"]), int(g["N"])
    wave = synth_wave(B, N, int(g["wave_seed"]))
    masks = torch.from_numpy(g["masks"])
    eng = _engine(gpu, B, N, precision=precision, taps=True)
    emb = eng.embed(wave.to(gpu), masks.to(gpu))
    torch.cuda.synchronize()
    emb = emb.cpu()
    fb = eng.debug_fetch("fbank").reshape(g["fbank"].shape)
    fb_err = np.abs(fb - g["fbank"]).max()
    assert fb_err < 2e-3, f"fbank max err {fb_err}"
    ref = torch.from_numpy(g["emb"])
    err = (emb - ref).abs().max().item()
    cos = torch.nn.functional.cosine_similarity(emb.reshape(-1, 256), ref.reshape(-1, 256), dim=-1)
    pool_ref = {}
    emb_model.emb_forward(emb_model.emb_state_dict(0), wave, masks, pool_ref)
    pool = eng.debug_fetch("pool").reshape(pool_ref["pool"].shape)
    perr = np.abs(pool - pool_ref["pool"].numpy()).max()
    rel = err / ref.abs().max().item()   # seeded He-init weights give |emb| ~ 1e3
    assert rel < 1e-4 and cos.min().item() > 0.9999, \
        f"emb rel err {rel}, min cos {cos.min().item()}, pool err {perr}"
    # inactive speaker: exactly the bias
    assert torch.equal(emb[0, 2], emb_model.emb_state_dict(0)["resnet.seg_1.bias"])


def test_embedding_single_mask_equals_multi_mask(built_lib, gpu):
    """drop-in form (one mask per item, PA/pipelines/speaker_verification.py:693-705) == shared trunk"""
    from oracle.gen_golden import synth_wave
    B, N, L = 3, 32000, 99
    wave = synth_wave(B, N, 77).to(gpu)
    g = torch.Generator().manual_seed(1)
    masks = (torch.rand(B, 4, L, generator=g) > 0.4).float().to(gpu)
    eng = _engine(gpu, B, N)
    multi = eng.embed(wave, masks)
    singles = torch.stack([eng.embed(wave, masks[:, s:s + 1].contiguous())[:, 0] for s in range(4)], dim=1)
    torch.cuda.synchronize()
    assert torch.equal(multi, singles)


def test_embedding_rejects_too_short_waveform(built_lib, gpu):
    """kaldi fbank asserts >= one 400-sample frame; min_num_samples' bisection relies on the raise
    (PA/pipelines/speaker_verification.py:677-691)"""
    eng = _engine(gpu, 1, 16000)
    with pytest.raises(Exception):
        eng.embed(torch.zeros(1, 399, device=gpu), torch.ones(1, 1, 10, device=gpu))
    out = eng.embed(torch.randn(1, 1600, device=gpu) * 0.1, torch.ones(1, 1, 4, device=gpu))
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()


def test_embedding_16s_window_and_masks_kernel(built_lib, gpu):
    """16 s windows (T = 1598 fbank frames, L = 799) + on-device median filter / mask logic vs the
    numpy restatement of diarizen/pipelines/inference.py:131-132 and speaker_diarization.py:268-322"""
    import math
    from scipy.ndimage import median_filter
    from oracle import emb_model
    from oracle.gen_golden import synth_wave
    B, N, L = 2, 256000, 799
    wave = synth_wave(B, N, 55)
    g = torch.Generator().manual_seed(9)
    # random two-state decisions with overlaps, short blips and an almost-fully-overlapped speaker
    raw = (torch.rand(B, L, 4, generator=g) > 0.55).to(torch.uint8)
    raw[0, :, 3] = 0
    raw[1, :, 2] = raw[1, :, 1]
    eng = _engine(gpu, B, N)
    min_frames = math.ceil(L * 400 / N)
    filt, masks = eng.prepare_masks(raw.to(gpu), 11, True, min_frames)
    torch.cuda.synchronize()
    ref_f = median_filter(raw.numpy().astype(np.float32), size=(1, 11, 1), mode="reflect")
    assert np.array_equal(filt.cpu().numpy(), ref_f.astype(np.uint8))
    clean = ref_f * (ref_f.sum(axis=2, keepdims=True) < 2)
    ref_m = np.stack([[clean[b, :, s] if clean[b, :, s].sum() > min_frames else ref_f[b, :, s]
                       for s in range(4)] for b in range(B)])
    assert np.array_equal(masks.cpu().numpy(), ref_m.astype(np.float32))
    emb = eng.embed(wave.to(gpu), masks)
    torch.cuda.synchronize()
    ref = emb_model.emb_forward(emb_model.emb_state_dict(0), wave, torch.from_numpy(ref_m.astype(np.float32)))
    rel = (emb.cpu() - ref).abs().max().item() / ref.abs().max().item()
    assert rel < 1e-4


@pytest.mark.parametrize("precision", ["f32", "f32s", "f32h", "f16"])
def test_embedding_planted_batchnorm_outliers(built_lib, gpu, precision):
    """The reference ResNet34 with PLANTED BatchNorm outliers (testkit/weights.py:emb_outlier_state_dict; golden
    emb_resnet_outlier.npz from oracle/gen_golden.py:gen_emb_outlier): two channels of the 32-plane stream (the fused BasicBlock
    kernels, whose intermediate is split under an a-priori |max| bound) and two of the 128-plane stream (generic contractions) sit
    2^10 x above the typical activation to the end of their stage.  fp32 modes: rel 1e-4 + cosine 0.9999 (the bar of the plain
    golden); reduced mode: cosine 0.999."""
    from diarizen_amd.configs import RESNET34, get_seg_config
    from diarizen_amd.engine import Engine
    from oracle import seg_model
    from oracle.gen_golden import synth_wave
    from testkit.weights import emb_outlier_state_dict
    g = np.load(os.path.join(GOLD, "emb_resnet_outlier.npz"))
    B, N = int(g["B"]), int(g["N"])
    cfg = get_seg_config("tiny_ln")
    esd = emb_outlier_state_dict(int(g["weight_seed"]))
    eng = Engine(cfg, seg_model.seg_state_dict(cfg, 0), RESNET34, esd, max_batch=B, max_samples=N, precision=precision, device=gpu)
    emb = eng.embed(synth_wave(B, N, int(g["wave_seed"])).to(gpu), torch.from_numpy(g["masks"]).to(gpu))
    torch.cuda.synchronize()
    emb = emb.cpu()
    ref = torch.from_numpy(g["emb"])
    rel = (emb - ref).abs().max().item() / ref.abs().max().item()
    cos = torch.nn.functional.cosine_similarity(emb.reshape(-1, 256), ref.reshape(-1, 256), dim=-1).min().item()
    print(f"[emb outlier {precision}] rel err {rel:.2e} min cosine {cos:.7f} (reference fp32 vs float64: {float(g['ref_fp32_vs_fp64_rel']):.1e})")
    if precision == "f16":
        assert cos > 0.999
    else:
        assert rel < 1e-4 and cos > 0.9999
    assert torch.equal(emb[0, 2], esd["resnet.seg_1.bias"])


@pytest.mark.parametrize("precision", ["bf16", "f16"])
def test_embedding_reduced_precision_engines(built_lib, gpu, precision):
    """bf16 engine mode (bf16 ResNet images / operands, fp32 accumulate, fbank + pooling + seg_1 fp32) and f16 mode
    (fp32 images, single-term fp16 contractions in stages 2-4, two-term 3x3 convs in stage 1):
    cosine >= 0.999 vs the reference golden, inactive speaker still exactly the bias"""
    from oracle import emb_model
    from oracle.gen_golden import synth_wave
    if precision == "bf16":
        from conftest import needs_bf16_mode
        needs_bf16_mode(built_lib)
    g = np.load(os.path.join(GOLD, "emb_resnet.npz"))
    B, N = int(g["B"]), int(g["N"])
    eng = _engine(gpu, B, N, precision=precision)
    emb = eng.embed(synth_wave(B, N, int(g["wave_seed"])).to(gpu), torch.from_numpy(g["masks"]).to(gpu))
    torch.cuda.synchronize()
    ref = torch.from_numpy(g["emb"])
    cos = torch.nn.functional.cosine_similarity(emb.cpu().reshape(-1, 256), ref.reshape(-1, 256), dim=-1)
    print(f"[{precision}] embeddings vs the reference golden: min cosine {cos.min().item():.7f}")
    if precision == "f16":
        import json
        os.makedirs("gpurun_out", exist_ok=True)
        json.dump({"min_cosine_vs_reference_golden": cos.min().item(), "bar": 0.999, "items": int(cos.numel())},
                  open("gpurun_out/f16_embedding_cosine.json", "w"))
    assert cos.min().item() > 0.999
    assert torch.equal(emb.cpu()[0, 2], emb_model.emb_state_dict(0)["resnet.seg_1.bias"])


def test_fbank_linear_mel_error_on_real_audio_vs_float64(built_lib, gpu):
    """How close is the device fbank (fp32 MFMA direct DFT + mel contraction) to the kaldi restatement evaluated in FLOAT64,
    on real speech (tests/golden/EN2002a_30s.wav, 4 windows of 8 s)?  Reported in log-mel units (what the ResNet sees,
    after the mean subtraction) and as linear mel energy relative to the frame's largest bin (VERDICT r2 #1d).  The
    float32 restatement itself sits at 4e-5 / 1.1e-5 (tests/test_oracle.py::test_fbank_cross_check_transformers)."""
    import json
    from oracle import emb_model
    from oracle.gen_golden import tt_windows
    B, N = 4, 128000
    wave = tt_windows([0, 90000, 200000, 352000], N)
    eng = _engine(gpu, B, N, precision="f32h", taps=True)
    masks = torch.ones(B, 4, 399)
    eng.embed(wave.to(gpu), masks.to(gpu))
    torch.cuda.synchronize()
    torch.set_default_dtype(torch.float64)
    try:
        raw64 = torch.stack([emb_model.kaldi_fbank(w.double() * (1 << 15)) for w in wave])       # log-mel before CMN
    finally:
        torch.set_default_dtype(torch.float32)
    ref64 = (raw64 - raw64.mean(dim=1, keepdim=True)).numpy()
    fb = eng.debug_fetch("fbank").reshape(ref64.shape).astype(np.float64)
    d = fb - ref64
    e = np.exp(raw64.numpy())
    weight = e / e.max(axis=2, keepdims=True)
    lin = np.abs(np.expm1(d)) * weight
    out = {"log_mel_max_abs": float(np.abs(d).max()), "log_mel_rms": float(np.sqrt((d ** 2).mean())),
           "linear_mel_rel_to_frame_max": float(lin.max()), "frames": int(ref64.shape[1]) * B}
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/fbank_vs_float64.json", "w"), indent=1)
    print(out)
    assert out["log_mel_max_abs"] < 2e-3 and out["linear_mel_rel_to_frame_max"] < 1e-4, out


def test_device_fbank_reproduces_the_hand_derived_known_answers(built_lib, gpu):
    """The device filter bank (frame_prep -> fp32 MFMA DFT -> power -> mel contraction -> log, embed.hip) against the
    CLOSED-FORM frames of tests/golden/fbank_known_answers.json (oracle/fbank_known_answers.py; neither restatement is
    involved).  The tap sits behind the per-window mean subtraction, so each window carries the known frame first and a DC
    stretch further on, whose frames are ln(FLT_EPSILON) exactly: tap[frame 0] - tap[DC frame] + ln(eps) is the known
    frame's log-mel vector whatever the mean was."""
    import json
    from oracle import fbank_known_answers as ka
    g = json.load(open(os.path.join(GOLD, "fbank_known_answers.json")))
    frames = ka.frames()
    names = ["impulse", "sine", "dc"]
    N = 400 + 9 * 160
    wave = torch.zeros(len(names), N, dtype=torch.float64)
    for b, nm in enumerate(names):
        wave[b, :400] = torch.tensor(frames[nm], dtype=torch.float64)
        wave[b, 800:] = 1234.0                      # frames 5.. lie inside the constant stretch
    wave = (wave / (1 << 15)).float()               # the model multiplies by 2^15 itself (wespeaker/__init__.py:96)
    eng = _engine(gpu, len(names), N, precision="f32h", taps=True)
    L = eng.num_frames(N)
    eng.embed(wave.to(gpu), torch.ones(len(names), 4, L, device=gpu))
    torch.cuda.synchronize()
    T = 1 + (N - 400) // 160
    fb = eng.debug_fetch("fbank").reshape(len(names), T, 80).astype(np.float64)
    ln_eps = -23.0 * np.log(2.0)
    for b, nm in enumerate(names):
        assert np.abs(fb[b, 5:] - fb[b, 5:6]).max() < 1e-6                      # the DC frames are all the floor
        got = fb[b, 0] - fb[b, 6] + ln_eps
        ref = np.array(g["log_mel"][nm])
        # 4 sinusoid / impulse samples of the int16 range are exactly representable after the 2^-15 scaling
        assert np.abs(got - ref).max() < 2e-3, (nm, float(np.abs(got - ref).max()))
        print(nm, "device fbank vs closed form: max |d log-mel| =", float(np.abs(got - ref).max()))
    eng.close()


@pytest.mark.parametrize("precision", ["f32h", "f32"])
def test_trunk_skipped_for_windows_without_active_speaker(built_lib, gpu, monkeypatch, precision):
    """(r3, VERDICT r2 missing #8; r4: decided on the device) A window whose S masks are all zero needs no ResNet trunk:
    zero weights pool to zero, so all its embeddings are seg_1's bias (pyannote-audio/tests/test_stats_pool.py:111-131).
    dzn_embed_forward builds the list of active windows ON THE DEVICE and launches the trunk over that subset; nothing is
    read back.  With silent windows in the batch (first, middle, last, two in a row) the result must be BIT-identical to
    the dense pass (an engine created under DZN_EMB_NO_SKIP=1) — the active windows because a window's result does not
    depend on its position in the batch, the silent ones because the dense pass gives the bias too — even though the
    skipped windows' image buffers hold the previous batch's data; an all-silent batch runs no trunk at all; the
    (device-side) counters say what was skipped."""
    from oracle import emb_model
    from oracle.gen_golden import synth_wave
    B, N, L = 9, 32000, 99
    eng = _engine(gpu, B, N, precision=precision)
    monkeypatch.setenv("DZN_EMB_NO_SKIP", "1")           # read once, at dzn_create
    eng_dense = _engine(gpu, B, N, precision=precision)
    monkeypatch.delenv("DZN_EMB_NO_SKIP")
    wave = synth_wave(B, N, 17).to(gpu)
    r = np.random.default_rng(3)
    masks = torch.from_numpy((r.random((B, 4, L)) < 0.4).astype(np.float32))
    for b in (0, 3, 4, 8):
        masks[b] = 0.0
    masks[5, 1:] = 0.0                                   # one active speaker keeps the window in
    masks = masks.to(gpu)
    eng.embed(synth_wave(B, N, 18).to(gpu), torch.ones_like(masks))     # every image buffer holds another batch's data
    w0, k0 = eng.embed_skip_stats()
    assert (w0, k0) == (B, 0)
    emb = eng.embed(wave, masks).clone()
    torch.cuda.synchronize()
    w1, k1 = eng.embed_skip_stats()
    assert (w1 - w0, k1 - k0) == (B, 4)
    dense = eng_dense.embed(wave, masks).clone()
    torch.cuda.synchronize()
    assert eng_dense.embed_skip_stats() == (B, 0)        # the switch really is the dense pass: every window counted, none skipped
    assert torch.equal(emb, dense)
    bias = emb_model.emb_state_dict(0)["resnet.seg_1.bias"]
    for b in (0, 3, 4, 8):
        assert all(torch.equal(emb[b, s].cpu(), bias) for s in range(4))
    assert torch.equal(emb[5, 1].cpu(), bias) and not torch.equal(emb[5, 0].cpu(), bias)
    ref = emb_model.emb_forward(emb_model.emb_state_dict(0), wave.cpu(), masks.cpu())
    assert (emb.cpu() - ref).abs().max().item() / ref.abs().max().item() < 1e-4
    # all silent: no trunk pass; no silent window: every window in the list
    allz = eng.embed(wave, torch.zeros_like(masks))
    torch.cuda.synchronize()
    assert all(torch.equal(allz[b, s].cpu(), bias) for b in range(B) for s in range(4))
    assert eng.embed_skip_stats()[1] == k1 + B
    eng.embed(wave, torch.ones_like(masks))
    torch.cuda.synchronize()
    assert eng.embed_skip_stats()[1] == k1 + B
    eng.close()
    eng_dense.close()


def test_trunk_subset_with_more_windows_than_one_scan_block(built_lib, gpu):
    """compact_active_kernel builds the list of active windows with one 256-thread workgroup that scans the flags in
    blocks of 256: B = 600 (the bench's launches hold 561) crosses two block boundaries.  Every third window is silent, plus
    runs of silent windows around the boundaries (254 .. 258, 510 .. 513): the embeddings must be bit-identical to the dense
    engine's, the silent windows must be the bias, and the counter must say how many were skipped."""
    from oracle import emb_model
    from oracle.gen_golden import synth_wave
    B, N, L = 600, 8000, 24
    eng = _engine(gpu, B, N, precision="f32h")
    import os
    os.environ["DZN_EMB_NO_SKIP"] = "1"
    try:
        dense_eng = _engine(gpu, B, N, precision="f32h")
    finally:
        os.environ.pop("DZN_EMB_NO_SKIP", None)
    wave = synth_wave(8, N, 23).repeat(75, 1)[:B].contiguous()
    wave = (wave * torch.linspace(0.5, 1.5, B)[:, None]).to(gpu)
    r = np.random.default_rng(11)
    masks = torch.from_numpy((r.random((B, 4, L)) < 0.5).astype(np.float32))
    silent = sorted(set(range(0, B, 3)) | set(range(254, 259)) | set(range(510, 514)))
    masks[silent] = 0.0
    masks = masks.to(gpu)
    emb = eng.embed(wave, masks)
    dense = dense_eng.embed(wave, masks)
    torch.cuda.synchronize()
    assert torch.equal(emb, dense)
    bias = emb_model.emb_state_dict(0)["resnet.seg_1.bias"].to(gpu)
    assert torch.equal(emb[silent], bias.expand(len(silent), 4, -1))
    assert eng.embed_skip_stats() == (B, len(silent))
    eng.close()
    dense_eng.close()


def test_forwards_only_enqueue_and_replay_from_a_hip_graph(built_lib, gpu):
    """include/dzn.h: "calls only ENQUEUE work on the given HIP stream".  Once the per-geometry tables exist (first call),
    dzn_segment_forward -> dzn_prepare_masks -> dzn_embed_forward must be capturable in a HIP graph — a stream
    synchronisation or a blocking copy inside any of them fails the capture (r3's embed call read the window flags back)
    — and the replay must reproduce the eager results bit for bit, silent windows included."""
    from oracle.gen_golden import synth_wave, tt_windows
    from testkit.weights import emb_state_dict, turn_taking_state_dict
    from diarizen_amd.configs import RESNET34, get_seg_config
    from diarizen_amd.engine import Engine
    cfg = get_seg_config("tiny_ln")
    B, N = 4, 16000
    eng = Engine(cfg, turn_taking_state_dict(cfg, 0), RESNET34, emb_state_dict(0), max_batch=B, max_samples=N, device=gpu)
    wave = tt_windows([32000, 96000, 160000, 200000], N).to(gpu)
    wave[2] = 0.0

    def forward():
        _, ml = eng.segment(wave, want_logp=False)
        filt, masks = eng.prepare_masks(ml, 11, True, 1)
        masks[1] = 0.0                                # a window without any active speaker
        return filt, masks, eng.embed(wave, masks)
    eager = [t.clone() for t in forward()]           # builds the per-geometry tables (their uploads do synchronise, once)
    torch.cuda.synchronize()
    skipped0 = eng.embed_skip_stats()[1]
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(device=gpu)
    side.wait_stream(torch.cuda.current_stream(gpu))
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            outs = forward()
    torch.cuda.synchronize()
    again = [t.clone() for t in forward()]           # eager twice: the forward itself is deterministic
    torch.cuda.synchronize()
    for name, a, b in zip(("decisions", "masks", "embeddings"), eager, again):
        assert torch.equal(a, b), f"eager {name} not reproducible: max |d| = {(a.float() - b.float()).abs().max().item():.3e}"
    problems = []
    for rep in range(3):
        for t in outs:
            t.zero_()
        g.replay()
        torch.cuda.synchronize()
        for name, a, b in zip(("decisions", "masks", "embeddings"), eager, outs):
            if not torch.equal(a, b):
                bad = (a != b).reshape(a.shape[0], -1).any(1).nonzero().flatten().tolist()
                d = (a.float() - b.float()).abs()
                problems.append({"replay": rep, "tensor": name, "windows": bad, "max_abs_diff": d.max().item(),
                                 "n_diff": int((a != b).sum()), "scale": a.float().abs().max().item()})
    if problems:
        import json
        os.makedirs("gpurun_out", exist_ok=True)
        json.dump(problems, open("gpurun_out/graph_replay_diff.json", "w"), indent=1)
    assert not problems, problems
    assert eng.embed_skip_stats()[1] >= skipped0 + 2     # the replayed kernels counted their skipped window
    eng.close()
