import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from diarizen_amd import ops
from oracle.gen_golden import synth_host_case
C = int((120 * 60 - 8.0) / 0.8) + 1
seg, emb = synth_host_case(3, C=C, L=99, n_spk=4)
e = emb[seg.sum(1) > 0].astype(np.float32); e /= np.linalg.norm(e, axis=-1, keepdims=True)
ops.linkage_centroid(e[:500])
t0 = time.perf_counter(); Z = ops.linkage_centroid(e); print(len(e), time.perf_counter() - t0)
