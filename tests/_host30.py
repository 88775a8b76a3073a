"""shared checker for the 30-min host-stage golden (tests/golden/host30.npz, oracle/gen_golden.py:gen_host30): one run of the
product's run_host_stage against one run of the REFERENCE's own clustering / reconstruction code on the same device outputs."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def run_and_check(device=None, linkage_backend="auto", cdist_backend="auto"):
    from diarizen_amd.clustering import AgglomerativeClustering
    from diarizen_amd.core import SlidingWindow
    from diarizen_amd.pipeline import run_host_stage
    g = np.load(os.path.join(GOLD, "host30.npz"))
    seg, emb = g["seg"], g["emb"]
    cl = AgglomerativeClustering(metric="cosine", method="centroid", min_cluster_size=13, threshold=0.1,
                                 linkage_backend=linkage_backend)
    cl.cdist_backend = cdist_backend
    if device is not None:
        cl.device = device.index if device.index is not None else 0
    seen = {}

    def clustering(**kw):
        out = cl(**kw)
        seen["hard"] = np.array(out[0], copy=True)
        return out

    def hook(step, artifact):
        seen[step] = artifact
    ann = run_host_stage(seg, emb, chunks=SlidingWindow(start=0.0, duration=8.0, step=0.8), clustering=clustering, min_speakers=1,
                         max_speakers=20, sess_name="host30", device=device, hook=hook)
    # 1. the clustering step: every (window, speaker) in the reference's cluster
    ref_hard = g["hard_clusters"].astype(np.int64)
    assert np.array_equal(seen["hard"].astype(np.int64), ref_hard), \
        f"{int((seen['hard'] != ref_hard).sum())} of {ref_hard.size} hard-cluster entries differ from the reference's"
    # 2. speaker counting
    assert np.array_equal(np.minimum(seen["speaker_counting"].data, 20).astype(np.int8).reshape(-1), g["count"].reshape(-1))
    # 3. discrete diarization: exact wherever the reference's own result is defined by the data; at frames whose top-`count`
    #    selection cuts through equal activations (np.argsort's tie order: numpy-build / CPU dependent) any valid selection
    binary = seen["discrete_diarization"].data.astype(np.uint8)
    ref_bin = g["binary"]
    assert binary.shape == ref_bin.shape
    tie = np.zeros(len(ref_bin), dtype=bool)
    tie[g["boundary_tie_frames"]] = True
    diff = np.nonzero((binary != ref_bin).any(axis=1))[0]
    assert not len(np.setdiff1d(diff, np.nonzero(tie)[0])), "discrete diarization differs at frames without a boundary tie"
    act, cnt = g["activations"], g["count"].reshape(-1).astype(np.int64)
    for t in diff:                                   # a differing tie frame must still be a valid top-count selection
        k = int(min(cnt[t], act.shape[1]))
        sel = binary[t].astype(bool)
        assert sel.sum() == k
        kth = np.sort(act[t])[::-1][k - 1]
        assert (act[t][sel] >= kth).all() and sel[act[t] > kth].all()
    rttm_ref = bytes(g["rttm"]).decode()
    if not len(diff):
        assert ann.to_rttm() == rttm_ref
    return {"tie_frames_resolved_differently": int(len(diff)), "rttm_equal": ann.to_rttm() == rttm_ref,
            "n_train": int(g["n_train"]), "rows": int(ref_hard.size)}
