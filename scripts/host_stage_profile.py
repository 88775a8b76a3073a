"""Where the HOST stage of one long recording goes (the serial tail of BASELINE configs[3]).
    python scripts/host_stage_profile.py [minutes=240] [batch=576] [repeats=3]
Runs the device stage of a synthetic recording once, then `run_host_stage` `repeats` times on its (decisions, embeddings) and
prints the wall time of each pass and a cProfile of the last one."""
import copy
import cProfile
import io
import pstats
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
from bench import synth_recording
from diarizen_amd.configs import get_seg_config
from diarizen_amd.pipeline import DiariZenPipeline
from testkit.weights import emb_state_dict, turn_taking_state_dict

minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 576
repeats = int(sys.argv[3]) if len(sys.argv) > 3 else 3
CONFIG = {
    "model": {"path": "diarizen.models.eend.model_wavlm_conformer.Model",
              "args": {"wavlm_src": "wavlm_large_s80_md", "wavlm_layer_num": 25, "wavlm_feat_dim": 1024, "chunk_size": 8}},
    "inference": {"args": {"seg_duration": 8, "segmentation_step": 0.1, "batch_size": batch, "apply_median_filtering": True}},
    "clustering": {"args": {"method": "AgglomerativeClustering", "min_speakers": 1, "max_speakers": 20,
                            "ahc_criterion": "distance", "ahc_threshold": 0.1, "min_cluster_size": 13}},
}
cfg = get_seg_config("wavlm_large_s80_md")
pipe = DiariZenPipeline(None, None, config=copy.deepcopy(CONFIG), device=torch.device("cuda:0"),
                        seg_state=turn_taking_state_dict(cfg, 0), emb_state=emb_state_dict(0))
x = synth_recording(int(minutes * 60 * 16000)).numpy()
t0 = time.perf_counter()
seg, emb = pipe.device_stage(x)
print(f"device stage {time.perf_counter() - t0:.3f} s: decisions {seg.shape} {seg.dtype}, embeddings {emb.shape}", flush=True)
for r in range(repeats):
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    ann = pipe.host_stage(seg, emb, "synth")
    pr.disable()
    print(f"host stage pass {r}: {time.perf_counter() - t0:.3f} s, {len(ann.labels())} speakers, {len(list(ann.itertracks()))} turns", flush=True)
for key in ("cumulative", "tottime"):
    buf = io.StringIO()
    pstats.Stats(pr, stream=buf).sort_stats(key).print_stats(32)
    print(buf.getvalue())
