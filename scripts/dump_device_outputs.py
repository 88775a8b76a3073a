"""Run the DEVICE stage of the pipeline on the bench's synthetic 30-min recording (seed 3407, seeded turn-taking weights) and
save what leaves the device — u8 decisions [C, L, 4] and f32 embeddings [C, 4, 256] — plus the product's own host-stage result
(device backends ON) for tests/golden/host_30min.* (oracle/gen_golden_host30.py pins them to the reference's own clustering /
reconstruction code on the CPU).  GPU box only:

    python scripts/dump_device_outputs.py [minutes=30] [out=gpurun_out/host30]
"""
import copy
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from bench import pipeline_conf  # noqa: E402
from diarizen_amd.configs import get_seg_config  # noqa: E402
from diarizen_amd.pipeline import DiariZenPipeline  # noqa: E402
from testkit.synth import synth_recording  # noqa: E402
from testkit.weights import emb_state_dict, turn_taking_state_dict  # noqa: E402


class A:
    model = "wavlm_large_s80_md"
    window = 8.0
    batch = 576


def main():
    minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
    out = Path(sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/host30")
    out.parent.mkdir(parents=True, exist_ok=True)
    cfg = get_seg_config(A.model)
    dev = torch.device("cuda:0")
    pipe = DiariZenPipeline(None, None, config=copy.deepcopy(pipeline_conf(A, cfg)), device=dev, precision="f32h",
                            seg_state=turn_taking_state_dict(cfg, 0), emb_state=emb_state_dict(0))
    x = np.ascontiguousarray(synth_recording(int(minutes * 60 * 16000), seed=3407).numpy())
    seg, emb = pipe.device_stage(x)
    ann = pipe.host_stage(seg, emb, "host30")
    np.savez_compressed(str(out) + ".npz", seg=seg, emb=emb)
    Path(str(out) + ".rttm").write_text(ann.to_rttm())
    print(json.dumps({"windows": int(seg.shape[0]), "frames": int(seg.shape[1]), "speakers": len(ann.labels()),
                      "rttm_lines": len(ann.to_rttm().splitlines()), "emb_rows": int(seg.shape[0] * seg.shape[2]),
                      "nan_rows": int(np.isnan(emb).any(-1).sum())}))


if __name__ == "__main__":
    main()
