"""Size-independent properties at BASELINE.json's full window size (wavlm-large-s80, 8 s windows),
where the CPU oracle is too slow to be the checker, plus ragged / degenerate inputs.

  * windows are independent: any batch composition / any contiguous shard of the window range gives
    BIT-IDENTICAL per-window results (this is the invariant the multi-GPU sharding relies on);
  * log-probabilities are normalised, hard decisions are a valid powerset row (<= 2 speakers/frame);
  * masks: median filter is idempotent on its own output for runs >= 6 frames; an inactive speaker's
    embedding is exactly seg_1.bias; embeddings do not depend on the other speakers' masks;
  * ragged inputs: recording shorter than one window, exact multiple, and a ragged tail follow the
    reference window plan (PA/core/inference.py:285-299) and the zero-padded last window equals the
    explicitly padded waveform.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["f32h", "f32s", "f32"])
def engine(built_lib, gpu, request):
    from diarizen_amd.configs import RESNET34, get_seg_config
    from diarizen_amd.engine import Engine
    from testkit.weights import emb_state_dict, seg_state_dict
    cfg = get_seg_config("wavlm_large_s80_md")
    return Engine(cfg, seg_state_dict(cfg, 0), RESNET34, emb_state_dict(0), max_batch=48,
                  max_samples=128000, precision=request.param, device=gpu)


def _recording(seconds, seed=5):
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from bench import synth_recording
    return synth_recording(int(seconds * 16000), seed=seed)


def test_batch_composition_and_sharding_are_bit_invariant(engine, gpu):
    from diarizen_amd.inference import WindowRunner
    wave = _recording(8.0 + 0.8 * 40).to(gpu)              # 41 windows
    r48 = WindowRunner(engine, 8.0, 0.1, batch_size=48)
    r7 = WindowRunner(engine, 8.0, 0.1, batch_size=7)
    full = r48.run(wave)
    small = r7.run(wave)
    torch.cuda.synchronize()
    assert full.segmentations.shape == (41, 399, 4) and full.embeddings.shape == (41, 4, 256)
    assert torch.equal(full.segmentations, small.segmentations)
    assert torch.equal(full.embeddings, small.embeddings)
    # contiguous shards (what each rank computes) concatenate to the full result
    from diarizen_amd.dist import shard_range
    parts = [r48.run(wave, window_range=shard_range(41, r, 3)) for r in range(3)]
    torch.cuda.synchronize()
    assert torch.equal(torch.cat([p.segmentations for p in parts]), full.segmentations)
    assert torch.equal(torch.cat([p.embeddings for p in parts]), full.embeddings)


def test_two_handles_on_two_streams_give_the_same_bits(built_lib, gpu):
    """(r4) WindowRunner(extra_engines=...): consecutive batches alternate over two handles with the same weights, each on
    its own HIP stream, so that independent batches overlap on the device.  Decisions and embeddings must be the bits of
    the one-handle, one-stream run (a window's result depends neither on its batch nor on the handle), also when the
    caller's stream still has the upload queued and when the number of batches is odd."""
    from diarizen_amd.configs import RESNET34, get_seg_config
    from diarizen_amd.engine import Engine
    from diarizen_amd.inference import WindowRunner
    from testkit.weights import emb_state_dict, turn_taking_state_dict
    cfg = get_seg_config("tiny_ln")
    engine, second = (Engine(cfg, turn_taking_state_dict(cfg, 0), RESNET34, emb_state_dict(0), max_batch=16, max_samples=16000,
                             precision="f32h", device=gpu) for _ in range(2))
    one = WindowRunner(engine, 1.0, 0.1, 16)
    two = WindowRunner(engine, 1.0, 0.1, 7, extra_engines=(second,))       # 7-window batches: odd batch count below
    host = _recording(6.3)
    for _ in range(2):
        wave = host.to(gpu, non_blocking=True)                              # the upload is still in flight on this stream
        b = two.run(wave)
        a = one.run(wave)
        torch.cuda.synchronize()
        assert a.segmentations.shape[0] == 54
        assert torch.equal(a.segmentations, b.segmentations) and torch.equal(a.embeddings, b.embeddings)
    engine.close()
    second.close()


def test_outputs_are_well_formed_at_full_size(engine, gpu):
    wave = _recording(8.0 + 0.8 * 31, seed=9).to(gpu)
    from diarizen_amd.inference import WindowRunner
    views = WindowRunner(engine, 8.0, 0.1, 32).windows_view(wave)
    logp, ml = engine.segment(views[:32].contiguous())
    torch.cuda.synchronize()
    assert torch.isfinite(logp).all()
    assert torch.allclose(logp.exp().sum(-1), torch.ones_like(logp[..., 0]), atol=1e-4)
    assert ml.sum(-1).max().item() <= 2                     # powerset: at most 2 speakers per frame
    from testkit.weights import emb_state_dict
    filt, masks = engine.prepare_masks(ml, 11, True, 2)
    filt2, _ = engine.prepare_masks(filt, 11, True, 2)
    emb = engine.embed(views[:32].contiguous(), masks)
    torch.cuda.synchronize()
    assert set(np.unique(filt.cpu().numpy()).tolist()) <= {0, 1}
    assert (filt2.cpu() != filt.cpu()).float().mean().item() < 0.02   # near-idempotent (blips already removed)
    inactive = masks.sum(-1) == 0
    bias = emb_state_dict(0)["resnet.seg_1.bias"].to(gpu)
    if inactive.any():
        assert torch.equal(emb[inactive], bias.expand_as(emb[inactive]))
    # an embedding depends only on its own mask
    m2 = masks.clone()
    m2[:, 1:] = 0
    emb2 = engine.embed(views[:32].contiguous(), m2)
    torch.cuda.synchronize()
    assert torch.equal(emb2[:, 0], emb[:, 0])


@pytest.mark.parametrize("seconds,expect", [(3.0, 1), (8.0, 1), (8.8, 2), (9.0, 3), (20.33, 17)])
def test_ragged_recordings_follow_reference_window_plan(engine, gpu, seconds, expect):
    from diarizen_amd.inference import WindowRunner
    wave = _recording(seconds, seed=3)
    r = WindowRunner(engine, 8.0, 0.1, batch_size=48)
    assert r.num_windows(wave.numel()) == expect
    res = r.run(wave.to(gpu), with_embeddings=False)
    torch.cuda.synchronize()
    assert res.segmentations.shape[0] == expect
    # the last window of a ragged recording == the explicitly zero-padded waveform run alone
    start = (expect - 1) * r.step
    last = torch.zeros(1, r.window)
    tail = wave[start:start + r.window]
    last[0, : tail.numel()] = tail
    _, ml = engine.segment(last.to(gpu), want_logp=False)
    filt, _ = engine.prepare_masks(ml, 11, True, 2, want_masks=False)
    torch.cuda.synchronize()
    assert torch.equal(filt[0], res.segmentations[-1])


def test_degenerate_waveforms_match_oracle(engine, gpu):
    """digital silence, a DC offset and a full-scale clipped square wave: finite outputs, strict parity with the
    oracle (the waveform LayerNorm divides by sqrt(var + eps) with var == 0 on the first two)."""
    from diarizen_amd.configs import get_seg_config
    from testkit.weights import seg_state_dict
    from oracle import seg_model
    cfg = get_seg_config("wavlm_large_s80_md")
    sd = seg_state_dict(cfg, 0)
    N = 16000
    t = torch.arange(N) / 16000.0
    wave = torch.stack([torch.zeros(N), torch.full((N,), 0.25), torch.sign(torch.sin(2 * np.pi * 220.0 * t))])
    logp, ml = engine.segment(wave.to(gpu))
    torch.cuda.synchronize()
    assert torch.isfinite(logp).all()
    ref = seg_model.seg_forward(sd, cfg, wave)
    assert (logp.cpu() - ref).abs().max().item() < 1e-3
    assert torch.equal(ml.cpu(), seg_model.to_multilabel(ref, cfg).to(torch.uint8))
    masks = torch.ones(3, 4, logp.shape[1], device=gpu)
    emb = engine.embed(wave.to(gpu), masks)
    torch.cuda.synchronize()
    assert torch.isfinite(emb).all()


def test_c_abi_without_python(built_lib, gpu, tmp_path):
    """The drop-in boundary is a C ABI: tests/c_abi_smoke.c (gcc -std=c11, built by diarizen_amd/build.py) creates a
    handle, loads every tensor, finalizes, runs dzn_segment_forward on ITS OWN hipMalloc'ed buffers and stream and
    checks the log-probs against the oracle's — no Python, torch or C++ on its side.  Python only writes the blob."""
    import struct
    import subprocess
    import ctypes as C
    from diarizen_amd import build as b
    from diarizen_amd.configs import get_seg_config
    from diarizen_amd.engine import make_dzn_config
    from testkit.weights import turn_taking_state_dict
    from oracle import seg_model
    from oracle.gen_golden import tt_windows
    exe = b.build_c_harness()
    if exe is None:
        pytest.skip("gcc not available")
    cfg = get_seg_config("tiny_ln")
    sd = turn_taking_state_dict(cfg, 0)
    B, N = 3, 12000
    wave = tt_windows([16000, 80000, 200000], N)
    ref = seg_model.seg_forward(sd, cfg, wave).numpy()
    L = ref.shape[1]
    zc = make_dzn_config(cfg, None, B, N, "f32h")
    blob = tmp_path / "blob.bin"
    with open(blob, "wb") as f:
        f.write(b"DZNBLOB1")
        f.write(bytes(zc))
        items = [(k, v) for k, v in sd.items()]
        f.write(struct.pack("<i", len(items)))
        for k, v in items:
            t = v.detach().cpu().contiguous()
            if t.dtype == torch.int64:
                dt = 2
            else:
                t, dt = t.float(), 0
            kb = k.encode()
            f.write(struct.pack("<i", len(kb)) + kb + struct.pack("<ii", dt, t.dim()))
            f.write(struct.pack(f"<{t.dim()}q", *t.shape))
            f.write(t.numpy().tobytes())
        f.write(struct.pack("<ii", B, N) + wave.numpy().astype(np.float32).tobytes())
        f.write(struct.pack("<ii", L, cfg.n_classes) + ref.astype(np.float32).tobytes() + struct.pack("<f", 1e-3))
    r = subprocess.run([str(exe), str(blob)], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "argmax flips=0" in r.stdout


def test_device_postprocess_equals_numpy(built_lib, gpu):
    """row f2: speaker counting and cluster activations aggregated on the device (integer atomics over the u8
    decisions) are bit-identical to the numpy restatement of Inference.aggregate / reconstruct — on the e2e golden
    and on a seeded 20-minute case with unassigned speakers, empty clusters and a cluster count above the speakers."""
    import os
    from diarizen_amd.core import SlidingWindow
    from diarizen_amd.postprocess import DevicePost, receptive_field, reconstruct, speaker_count
    frames = receptive_field(16000)
    cases = []
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "e2e_EN2002a_30s.npz"))
    cases.append((g["seg"], g["hard_clusters"].copy(), 8.0))
    r = np.random.default_rng(5)
    C, L = 1500, 399
    seg = (r.random((C, L, 4)) < 0.3).astype(np.uint8)
    seg = seg * (r.random((C, 1, 4)) >= 0.3)                         # inactive speakers
    seg = np.ascontiguousarray(seg.astype(np.uint8))
    hard = r.integers(-2, 9, size=(C, 4)).astype(np.int8)            # -2 / -1 never contribute; cluster 7 may be empty
    hard[hard == 7] = 3
    cases.append((seg, hard, 8.0))
    for seg, hard, dur in cases:
        chunks = SlidingWindow(start=0.0, duration=dur, step=0.1 * dur)
        hard = np.array(hard, copy=True)
        hard[seg.sum(1) == 0] = -2
        segf = seg.astype(np.float32)
        c_np = speaker_count(segf, chunks, frames)
        post = DevicePost(seg, chunks, frames, gpu)
        c_dev = post.speaker_count()
        assert c_dev.data.dtype == np.uint8 and np.array_equal(c_dev.data, c_np.data)
        c_np.data = np.minimum(c_np.data, 20).astype(np.int8)
        b_np, a_np = reconstruct(segf, chunks, hard, c_np)
        b_dev, a_dev = post.reconstruct(hard, c_np)
        assert np.array_equal(a_dev.data, a_np.data.astype(np.float32))
        assert np.array_equal(b_dev.data, b_np.data)
