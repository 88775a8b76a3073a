"""End-to-end DiariZenPipeline timing on a synthetic recording (device stage vs host stage)."""
import copy, sys, time, wave as wavmod
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from bench import synth_recording
from diarizen_amd.configs import get_seg_config
from diarizen_amd.pipeline import DiariZenPipeline
from testkit.weights import emb_state_dict, turn_taking_state_dict

minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 128
CONFIG = {
    "model": {"path": "diarizen.models.eend.model_wavlm_conformer.Model",
              "args": {"wavlm_src": "wavlm_large_s80_md", "wavlm_layer_num": 25, "wavlm_feat_dim": 1024,
                       "chunk_size": 8}},
    "inference": {"args": {"seg_duration": 8, "segmentation_step": 0.1, "batch_size": batch,
                           "apply_median_filtering": True}},
    "clustering": {"args": {"method": "AgglomerativeClustering", "min_speakers": 1, "max_speakers": 20,
                            "ahc_criterion": "distance", "ahc_threshold": 0.1, "min_cluster_size": 13}},
}
cfg = get_seg_config("wavlm_large_s80_md")
pipe = DiariZenPipeline(None, None, config=copy.deepcopy(CONFIG), device=torch.device("cuda:0"),
                        seg_state=turn_taking_state_dict(cfg, 0), emb_state=emb_state_dict(0))
x = synth_recording(int(minutes * 60 * 16000))
path = "/tmp/synth.wav"
with wavmod.open(path, "wb") as w:
    w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
    w.writeframes((x.numpy() * 32767).astype("<i2").tobytes())
pipe(path, "warm")                      # warm-up (tables, allocator)
import cProfile, pstats
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
ann = pipe(path, "synth")
pr.disable()
dt = time.perf_counter() - t0
print("timings", {k: round(v, 3) for k, v in pipe.timings.items()}, "total", round(dt, 3),
      "audio-s/s e2e", round(pipe.timings["audio_s"] / dt, 1), "turns", len(list(ann.itertracks())),
      "speakers", len(ann.labels()))
import json
print("E2E_JSON", json.dumps({"minutes": minutes, "batch": batch, "timings_s": {k: round(v, 3) for k, v in pipe.timings.items()},
                              "total_s": round(dt, 3), "audio_seconds_per_s": round(pipe.timings["audio_s"] / dt, 1),
                              "speakers": len(ann.labels()), "turns": len(list(ann.itertracks())),
                              "weights": "seeded turn-taking weights", "precision": pipe.engine.precision}))
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
