"""Host stage against goldens made by the REFERENCE's OWN functions (row f2 of SURVEY §8f).

tests/golden/host_ref.npz is written by `oracle/gen_golden.py host_ref`: oracle/ref_host.py imports, by path from
/root/reference, `Inference.aggregate / trim` (PA/core/inference.py:544-714), `speaker_count` / `to_diarization`
(PA/pipelines/utils/diarization.py:121-239), `SpeakerDiarization.reconstruct` (PA/pipelines/speaker_diarization.py:377-425)
and `Binarize` (PA/utils/signal.py:254-317) and runs them in the order of diarizen/pipelines/inference.py:137-185 on seeded
hard decisions.  Here BOTH re-implementations must reproduce every count, every discrete-diarization frame and every RTTM
byte: the oracle's loop-for-loop restatement (oracle/host_stage.py) and the product's vectorised host stage
(diarizen_amd/postprocess.py).  When /root/reference is present the recipe itself is re-run and compared with the
committed file.
"""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")
G = np.load(os.path.join(GOLD, "host_ref.npz"))
CASES = [str(c) for c in G["cases"]]


def _case(name):
    dur, ratio, max_spk = G[f"{name}_args"]
    return (G[f"{name}_seg"], G[f"{name}_hard"], float(dur), float(ratio), int(max_spk),
            G[f"{name}_count"], G[f"{name}_binary"], G[f"{name}_rttm"].tobytes().decode())


@pytest.mark.parametrize("name", CASES)
def test_oracle_host_stage_equals_reference_run(name):
    from oracle import host_stage as hs
    seg, hard, dur, ratio, max_spk, count_ref, binary_ref, rttm_ref = _case(name)
    segf = seg.astype(np.float32)
    chunks = hs._SW(start=0.0, duration=dur, step=ratio * dur)
    frames = hs._SW(*hs.RECEPTIVE_FIELD)
    count, count_frames = hs.speaker_count(segf, chunks, frames, warm_up=(0.0, 0.0))
    count = np.minimum(count, max_spk).astype(np.int8)
    assert np.array_equal(count, count_ref)
    h = np.array(hard, copy=True)
    h[np.sum(segf, axis=1) == 0] = -2
    binary = hs.reconstruct(segf, chunks, h, count, count_frames)
    assert np.array_equal(binary.astype(np.uint8), binary_ref)
    assert hs.host_stage(seg, hard, dur, ratio, max_spk, name) == rttm_ref


@pytest.mark.parametrize("name", CASES)
def test_product_host_stage_equals_reference_run(name):
    from diarizen_amd.core import SlidingWindow
    from diarizen_amd.pipeline import run_host_stage
    seg, hard, dur, ratio, max_spk, count_ref, binary_ref, rttm_ref = _case(name)
    seen = {}

    def clustering(embeddings, segmentations, min_clusters, max_clusters):
        return np.array(hard, copy=True), None, None          # the golden's clusters: this test is about everything else

    def hook(step, artifact):
        seen[step] = artifact
    emb = np.zeros(seg.shape[:1] + (seg.shape[2], 4), dtype=np.float32)
    ann = run_host_stage(seg, emb, chunks=SlidingWindow(start=0.0, duration=dur, step=ratio * dur), clustering=clustering,
                         min_speakers=1, max_speakers=max_spk, sess_name=name, hook=hook)
    # the hook sees the count before the max_speakers cap (inference.py:137-142 vs :163): cap it here
    assert np.array_equal(np.minimum(seen["speaker_counting"].data, max_spk).astype(np.int8), count_ref)
    assert np.array_equal(seen["discrete_diarization"].data.astype(np.uint8), binary_ref)
    fr = seen["discrete_diarization"].sliding_window
    assert np.allclose([fr.start, fr.duration, fr.step], G[f"{name}_frames"], rtol=0, atol=0)
    assert ann.to_rttm() == rttm_ref


@pytest.mark.parametrize("i", [0, 1, 2])
def test_aggregate_soft_scores_equals_reference_run(i):
    """Inference.aggregate with a Hamming window, warm-up and missing (NaN) chunks: oracle restatement always; the
    product's aggregate has no warm-up argument (this path uses (0, 0)), so it is compared on the warm-up-free case."""
    from oracle import host_stage as hs
    dur, step, ham, wl, wr, skip = G[f"agg{i}_args"]
    sc, ref = G[f"agg{i}_scores"], G[f"agg{i}_out"]
    # the reference takes warm_up in SECONDS here (inference.py:600-607); the restatement uses the same convention
    out, fr = hs.aggregate(sc, hs._SW(0.0, dur, step), hs._SW(*hs.RECEPTIVE_FIELD), hamming=bool(ham), missing=np.nan,
                           skip_average=bool(skip), warm_up=(wl, wr))
    assert out.shape == ref.shape
    assert np.array_equal(np.isnan(out), np.isnan(ref))
    assert np.array_equal(np.nan_to_num(out), np.nan_to_num(ref))
    assert np.allclose([fr.start, fr.duration, fr.step], G[f"agg{i}_frames"], rtol=0, atol=0)
    if wl == 0 and wr == 0:
        from diarizen_amd.core import SlidingWindow
        from diarizen_amd.postprocess import aggregate, receptive_field
        p = aggregate(sc, SlidingWindow(start=0.0, duration=dur, step=step), receptive_field(), hamming=bool(ham),
                      missing=np.nan, skip_average=bool(skip))
        assert np.array_equal(np.isnan(p.data), np.isnan(ref))
        assert np.array_equal(np.nan_to_num(p.data), np.nan_to_num(ref))


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs /root/reference (build container only)")
def test_reference_functions_reproduce_committed_goldens():
    """the recipe: the reference's own functions, run now, give the committed fixtures (incl. both e2e RTTMs)"""
    from oracle import ref_host
    for name in CASES:
        seg, hard, dur, ratio, max_spk, count_ref, binary_ref, rttm_ref = _case(name)
        rttm, parts = ref_host.host_stage(seg, hard, dur, ratio, max_spk, name, return_parts=True)
        assert rttm == rttm_ref
        assert np.array_equal(parts["count"], count_ref) and np.array_equal(parts["binary"].astype(np.uint8), binary_ref)
    g = np.load(os.path.join(GOLD, "e2e_EN2002a_30s.npz"))
    for tag, key in (("", "hard_clusters"), ("_vbx", "hard_clusters_vbx")):
        assert ref_host.host_stage(g["seg"], g[key], 8.0, 0.1, 20, "EN2002a") == \
            open(os.path.join(GOLD, f"e2e_EN2002a_30s{tag}.rttm")).read()


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs /root/reference (build container only)")
def test_reference_get_embeddings_drives_the_embedding_facade(monkeypatch):
    """(b) boundary, embedding side (VERDICT r2 missing #7): the reference's OWN `SpeakerDiarization.get_embeddings`
    (PA/pipelines/speaker_diarization.py:228-375, imported by path) — its overlap-excluded masks with the `min_num_frames`
    fallback, its `Audio.crop(mode="pad")` calls, its batching of (chunk, speaker) pairs — drives
    `diarizen_amd.models.SpeakerEmbedding` through exactly the surface it reads: `min_num_samples` (the bisection of
    speaker_verification.py:677-691 runs against the facade), `sample_rate`, `__call__(waveforms [B, 1, N], masks=[B, F])
    -> np.ndarray [B, 256]`.  The engine behind the facade is a CPU stand-in that answers `embed()` with the oracle's
    ResNet34 (this test is about the INTERFACE; the HIP engine has its own parity tests), so the result must be the
    embeddings of the committed e2e fixture, which the oracle's restatement of this loop produced."""
    import torch
    from diarizen_amd.audio import first_channel_16k
    from diarizen_amd.configs import RESNET34
    from diarizen_amd.models import SpeakerEmbedding
    from oracle import emb_model, ref_host
    from testkit.weights import emb_state_dict
    ns = ref_host.load()
    esd = emb_state_dict(0)
    calls = []

    class CpuEngine:                          # interface stand-in for diarizen_amd.engine.Engine
        device = torch.device("cpu")
        seg = None
        emb = RESNET34

        def embed(self, w, m):                # [B, N], [B, S, L] -> [B, S, 256]
            calls.append((tuple(w.shape), tuple(m.shape)))
            if w.shape[1] < 400:              # shorter than one fbank frame: the raise min_num_samples bisects on
                raise RuntimeError("waveform shorter than one fbank frame")
            return emb_model.emb_forward(esd, w, m)

    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    facade = SpeakerEmbedding(engine=CpuEngine())
    assert facade.min_num_samples == 400 and facade.sample_rate == 16000 and facade.dimension == 256
    wave = torch.from_numpy(first_channel_16k(os.path.join(GOLD, "EN2002a_30s.wav")))[None]      # [1, N]
    g = np.load(os.path.join(GOLD, "e2e_EN2002a_30s.npz"))
    C = 5                                     # 20 ResNet34 passes of the oracle on the CPU

    class Audio:                              # Audio.crop(file, chunk, duration=, mode="pad") (PA/core/io.py:268-436)
        def crop(self, file, chunk, duration=None, mode="raise"):
            assert mode == "pad"
            s, n = round(chunk.start * 16000), round(duration * 16000)
            w = torch.zeros(1, n)
            have = file["waveform"][:, s:s + n]
            w[:, :have.shape[1]] = have
            return w, 16000

    pipe = object.__new__(ns.SpeakerDiarization)      # no __init__: that would load checkpoints through pyannote.audio
    pipe._embedding, pipe._audio, pipe.embedding_batch_size, pipe.training = facade, Audio(), 6, False
    seg = ns.core.SlidingWindowFeature(g["seg"][:C].astype(np.float32), ns.core.SlidingWindow(start=0.0, duration=8.0, step=0.8))
    n0 = len(calls)
    emb = pipe.get_embeddings({"waveform": wave}, seg, exclude_overlap=True)
    assert emb.shape == (C, 4, 256)
    assert [c[0][0] for c in calls[n0:]] == [6, 6, 6, 2]                  # ceil(5 * 4 / 6) batches of (chunk, speaker) pairs
    assert all(c[1][1] == 1 for c in calls[n0:])                          # one mask per row, as the reference calls it
    assert np.abs(emb - g["emb"][:C]).max() <= 1e-5 * np.abs(g["emb"][:C]).max()


def test_product_host_stage_equals_reference_run_at_30min_scale():
    """(r5) ONE run of the product's run_host_stage (scipy / numpy backends: no device here) against ONE run
    of the reference's own `AgglomerativeClustering` + reconstruct + Binarize on the same arrays — the device outputs of the
    bench's 30-min recording, 2241 windows = 8964 (window, speaker) rows, 2415 training embeddings: past the sizes at which the
    product switches to its vectorised assignment shortcut, and (with a device: tests/test_pipeline_gpu.py) to its device
    linkage and device cdist.  tests/_host30.py: hard clusters, counts and every frame whose selection is defined by the data are
    exact; frames cut through equal activations may differ only as another valid selection (np.argsort tie order)."""
    from tests._host30 import run_and_check
    res = run_and_check(device=None, linkage_backend="scipy", cdist_backend="scipy")
    print(res)


def test_device_score_deviation_cannot_move_the_reference_result_at_30min_scale(monkeypatch):
    """The device cosine scores (csrc/linkage.hip:dzn_cdist_cosine) agree with scipy.cdist to 2e-15 — and the 30-min fixture
    has hundreds of windows whose constrained assignment is an EXACT tie (local speakers with bit-identical embeddings: seg_1's
    bias for every inactive or zero-weight speaker), which linear_sum_assignment breaks by the last bit of the scores: the first
    GPU run of the device backends moved 242 active assignments and one RTTM frame.  clustering._exact_scores_for_tied_rows gives
    tied rows (and whole windows whose tie involves an active speaker) scipy's own bits.  Here, without a GPU: a stand-in for the
    device kernel = scipy's scores plus a 2e-15-level deviation that (like the kernel) is a function of the row alone; every
    active assignment, every frame and the RTTM must equal the reference's run."""
    from scipy.spatial.distance import cdist
    from diarizen_amd import clustering as cl, ops
    from tests._host30 import run_and_check

    def fake_cdist(flat, cent, device=-1):
        d = cdist(flat, cent, metric="cosine")
        h = (np.abs(flat[:, :8].astype(np.float64)).sum(1) * 1e6 % 1.0) - 0.5
        return d + 2e-15 * h[:, None] * np.arange(1, d.shape[1] + 1)
    monkeypatch.setattr(ops, "cdist_cosine", fake_cdist)
    monkeypatch.setattr(cl, "_hip_ready", lambda: True)
    res = run_and_check(device=None, linkage_backend="scipy", cdist_backend="hip")
    assert res["of_them_active"] == 0 and res["rttm_equal"] and res["tie_frames_resolved_differently"] == 0, res
    # and without the repair the same deviation does move the result (the test would be vacuous otherwise)
    monkeypatch.setattr(cl, "_exact_scores_for_tied_rows", lambda *a, **k: None)
    with pytest.raises(AssertionError):
        run_and_check(device=None, linkage_backend="scipy", cdist_backend="hip")
