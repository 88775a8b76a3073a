"""Host stage (counting, AHC, assignment, reconstruction, Binarize) cost vs recording length, CPU only.
python scripts/host_scaling.py 30 60 120   (minutes)"""
import sys, time, cProfile, pstats
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from oracle.gen_golden import synth_host_case
from diarizen_amd import clustering as cl
from diarizen_amd.core import SlidingWindow
from diarizen_amd.postprocess import binarize, receptive_field, reconstruct, speaker_count

for minutes in map(float, sys.argv[1:]):
    C = int((minutes * 60 - 8.0) / 0.8) + 1
    seg, emb = synth_host_case(1, C=C, L=399, n_spk=4)
    chunks = SlidingWindow(start=0.0, duration=8.0, step=0.8)
    frames = receptive_field(16000)
    t = {}
    t0 = time.perf_counter(); count = speaker_count(seg, chunks, frames); t["count"] = time.perf_counter() - t0
    ahc = cl.AgglomerativeClustering(threshold=0.7, min_cluster_size=13)
    t0 = time.perf_counter(); hard, _, _ = ahc(embeddings=emb, segmentations=seg, min_clusters=1, max_clusters=20)
    t["cluster"] = time.perf_counter() - t0
    count.data = np.minimum(count.data, 20).astype(np.int8)
    hard = np.array(hard, copy=True); hard[np.sum(seg, axis=1) == 0] = -2
    t0 = time.perf_counter(); disc, _ = reconstruct(seg, chunks, hard, count); t["reconstruct"] = time.perf_counter() - t0
    t0 = time.perf_counter(); ann = binarize(disc, onset=0.5, offset=0.5, uri="x"); t["binarize"] = time.perf_counter() - t0
    print(f"{minutes:g} min: C={C} E<={4*C}", {k: round(v, 3) for k, v in t.items()}, "total", round(sum(t.values()), 3), flush=True)
