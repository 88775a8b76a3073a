"""worker for tests/test_dist_gpu.py: one rank of a torch.distributed job on ONE visible GPU (DZN_BENCH_ONE_DEVICE
style): shards the windows of a recording, runs them through the HIP engine and all-gathers through the backend
given in DZN_TEST_BACKEND ("nccl" = RCCL).  Rank 0 compares with the unsharded run and prints DIST_OK."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import torch.distributed as dist

from diarizen_amd import dist as dz
from diarizen_amd.configs import RESNET34, get_seg_config
from diarizen_amd.engine import Engine
from diarizen_amd.inference import WindowRunner
from testkit.synth import synth_recording
from testkit.weights import emb_state_dict, turn_taking_state_dict

backend = os.environ.get("DZN_TEST_BACKEND", "nccl")
# one device per rank wherever the box has them (then `nccl` IS a real multi-rank RCCL job); ranks share device 0 only
# on a box with fewer devices than ranks (RCCL refuses duplicate devices there: the test stages that run through gloo)
_world = int(os.environ.get("WORLD_SIZE", "1"))
_local = int(os.environ.get("LOCAL_RANK", "0"))
dev = torch.device("cuda", _local if torch.cuda.device_count() >= _world else 0)
torch.cuda.set_device(dev)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
if backend == "nccl":
    dist.init_process_group(backend="nccl", device_id=dev)
else:
    dist.init_process_group(backend=backend)
rank, world = dist.get_rank(), dist.get_world_size()
cfg = get_seg_config("tiny_ln")
eng = Engine(cfg, turn_taking_state_dict(cfg, 0), RESNET34, emb_state_dict(0), max_batch=8, max_samples=16000,
             precision="f32h", device=dev)
runner = WindowRunner(eng, 1.0, 0.1, 8)
wave = synth_recording(16000 * 7 + 321, seed=5)
C = runner.num_windows(wave.numel())
rng = dz.my_window_range(C)
if rng is None:
    rng = (0, C)
c0, c1 = rng
n = (c1 - c0 - 1) * runner.step + runner.window if c1 > c0 else 0
sl = torch.zeros(n)
have = wave[c0 * runner.step:c0 * runner.step + n]
sl[:have.numel()] = have                                  # slice + halo only, zero-extended like the last window
res = runner.run(sl.to(dev), with_embeddings=True) if n else None
S = cfg.max_speakers_per_chunk
seg_l = res.segmentations if res is not None else torch.empty((0, runner.num_frames, S), device=dev, dtype=torch.uint8)
emb_l = res.embeddings if res is not None else torch.empty((0, S, 256), device=dev)
if backend != "nccl":
    seg_l, emb_l = seg_l.cpu(), emb_l.cpu()
seg, emb = dz.gather_windows(seg_l, emb_l, expected_total=C)      # one packed collective, blocks verified by their headers
if rank == 0:
    full = runner.run(wave.to(dev), with_embeddings=True)
    torch.cuda.synchronize()
    assert seg.shape[0] == C, (seg.shape, C)
    assert torch.equal(seg.cpu(), full.segmentations.cpu()), "sharded decisions differ"
    assert torch.equal(emb.cpu(), full.embeddings.cpu()), "sharded embeddings differ"
    print(f"DIST_OK backend={backend} world={world} windows={C} devices={min(world, torch.cuda.device_count())}")
dist.barrier()
eng.close()

# ---- the whole pipeline under the process group: every rank decodes only the byte range of its windows (audio.WavSource),
# runs them, the results are all-gathered with the block-partition assertion, rank 0 runs the host stage.  Its RTTM must be
# the committed 1-GPU golden (tests/golden/e2e_EN2002a_30s.rttm — the RTTM test_pipeline_gpu.py holds the unsharded
# pipeline to).
if os.environ.get("DZN_TEST_RTTM", "1") == "1":
    import copy
    from diarizen_amd.pipeline import DiariZenPipeline
    from oracle.gen_golden import E2E_CONFIG
    gold = Path(__file__).resolve().parent / "golden"
    big = get_seg_config("wavlm_large_s80_md")
    conf = copy.deepcopy(E2E_CONFIG)
    conf["inference"]["args"]["batch_size"] = 16
    pipe = DiariZenPipeline(None, None, config=conf, device=dev, precision="f32h",
                            seg_state=turn_taking_state_dict(big, 0), emb_state=emb_state_dict(0))
    ann = pipe(str(gold / "EN2002a_30s.wav"), sess_name="EN2002a")
    if rank == 0:
        assert ann.to_rttm() == (gold / "e2e_EN2002a_30s.rttm").read_text(), "sharded pipeline RTTM != 1-GPU golden"
        print(f"DIST_RTTM_OK backend={backend} world={world} load_s={pipe.timings['load_s']:.3f}")
    else:
        assert ann is None
    dist.barrier()
dist.destroy_process_group()
