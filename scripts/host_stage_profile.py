"""cProfile of the HOST stage (counting, clustering, reconstruction, Binarize) of a long synthetic recording:
python scripts/host_stage_profile.py [minutes=240] [batch=384]  — the device stage runs once, the host stage three times."""
import copy, cProfile, pstats, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from diarizen_amd.configs import get_seg_config
from diarizen_amd.pipeline import DiariZenPipeline
from testkit.synth import synth_recording_range
from testkit.weights import emb_state_dict, turn_taking_state_dict

minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 384
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
n = int(minutes * 60 * 16000)
x = synth_recording_range(0, n, total=n).numpy()
seg, emb = pipe.device_stage(x)
print("device stage done:", seg.shape, emb.shape, flush=True)
pipe.host_stage(seg, emb, "warm")
for it in range(2):
    t = time.perf_counter(); pipe.host_stage(seg, emb, "x"); print("host_s", round(time.perf_counter() - t, 3))
pr = cProfile.Profile(); pr.enable()
pipe.host_stage(seg, emb, "x")
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(40)
st.sort_stats("cumulative").print_stats(45)
st.print_callers("reduce", "astype", "copy")
