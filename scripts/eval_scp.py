"""Data-set inference + scoring, the recipe's stage 3 (recipes/diar_ssl/infer_avg.py:28-97 + run_stage.sh:84-91) on the
MI355X pipeline — BASELINE.json configs[4]'s acceptance harness ("DER vs reference on AMI-SDM"):

    python scripts/eval_scp.py -i data/test/AMI/wav.scp -o out/AMI --ref-rttm data/test/AMI/rttm --uem data/test/AMI/all.uem \\
        --diarizen-hub <hub dir> [--embedding-model <ckpt>] [--precision f32h|f16] [--collar 0]
    python -m torch.distributed.run --nproc-per-node 8 ... scripts/eval_scp.py ...      # recordings sharded over the ranks

Every recording of the Kaldi wav.scp goes through DiariZenPipeline (one process per GPU; with N ranks, rank r takes
recordings r, r+N, ...: files are independent, no collective on the data path), the RTTMs are written to the output
directory and rank 0 scores them against the reference RTTM (+ UEM) with diarizen_amd/der.py (collar 0, overlaps scored,
optimal mapping — what the recipe's dscore call computes).  No trained weights exist offline: `--synthetic-weights` runs the
harness on the seeded turn-taking weights (a plumbing check; the DER means nothing).
"""
from __future__ import annotations

import argparse
import copy
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def load_scp(path: str):
    """{rec: wav path} in file order (infer_avg.py:23-26)"""
    out = {}
    for line in open(path):
        f = line.strip().split(None, 1)
        if len(f) == 2:
            out[f[0]] = f[1]
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("-i", "--in_wav_scp", required=True)
    ap.add_argument("-o", "--out_dir", required=True)
    ap.add_argument("--ref-rttm", default=None, help="reference RTTM of the whole set (scoring is skipped without it)")
    ap.add_argument("--uem", default=None)
    ap.add_argument("--collar", type=float, default=0.0)
    ap.add_argument("--diarizen-hub", default=None, help="hub directory (config.toml, pytorch_model.bin, plda/)")
    ap.add_argument("--embedding-model", default=None)
    ap.add_argument("--precision", default="f32h", choices=["f32h", "f32s", "f32", "f16", "bf16"])
    ap.add_argument("--serial", action="store_true", help="one recording after the other (no host-stage overlap across recordings)")
    ap.add_argument("--synthetic-weights", action="store_true",
                    help="seeded turn-taking weights + the e2e fixture's configuration (no checkpoints offline)")
    ap.add_argument("--seg_duration", type=float, default=None)
    ap.add_argument("--batch_size", type=int, default=None)
    args = ap.parse_args(argv)

    import torch
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    from diarizen_amd.pipeline import DiariZenPipeline
    if args.synthetic_weights:
        from diarizen_amd.configs import get_seg_config
        from oracle.gen_golden import E2E_CONFIG          # the fixture's [model] / [inference] / [clustering] tables
        from testkit.weights import emb_state_dict, turn_taking_state_dict
        conf = copy.deepcopy(E2E_CONFIG)
        if args.seg_duration:
            conf["inference"]["args"]["seg_duration"] = args.seg_duration
        if args.batch_size:
            conf["inference"]["args"]["batch_size"] = args.batch_size
        cfg = get_seg_config(conf["model"]["args"]["wavlm_src"])
        pipe = DiariZenPipeline(None, None, config=conf, device=dev, precision=args.precision,
                                seg_state=turn_taking_state_dict(cfg, 0), emb_state=emb_state_dict(0), rttm_out_dir=args.out_dir)
    else:
        if not args.diarizen_hub:
            ap.error("--diarizen-hub (or --synthetic-weights) is required")
        emb = args.embedding_model or str(Path(args.diarizen_hub) / "wespeaker" / "pytorch_model.bin")
        pipe = DiariZenPipeline(args.diarizen_hub, emb, device=dev, precision=args.precision, rttm_out_dir=args.out_dir)
    os.makedirs(args.out_dir, exist_ok=True)
    # ranks share only the output directory: a done-file per rank, named after THIS job (launcher run id + rendezvous port) so
    # that the leftovers of an aborted earlier run are not mistaken for this run's, removed before any work starts, and
    # written atomically (temp file + os.replace) so that rank 0 never reads half a file
    # (r5) without a launcher the key carries the pid — two solo jobs sharing an output directory no longer take each
    # other's files — and only THIS job's leftovers are removed
    if "TORCHELASTIC_RUN_ID" in os.environ:
        job = f"{os.environ['TORCHELASTIC_RUN_ID']}_{os.environ.get('MASTER_PORT', '0')}"
    else:
        job = f"solo{os.getpid()}_{os.environ.get('MASTER_PORT', '0')}"

    def done_file(r):
        return Path(args.out_dir) / f".done_{job}_rank{r}.json"
    done_file(rank).unlink(missing_ok=True)
    scp = load_scp(args.in_wav_scp)
    mine = list(scp.items())[rank::world]
    audio_s = 0.0
    # the corpus loop of the reference's entry point (diarizen/pipelines/inference.py:365-368), host stage of recording i beside
    # the decode + device stage of recording i+1 (DiariZenPipeline.diarize_many); --serial = one `pipe(wav)` after the other
    t_all = time.perf_counter()
    for k, (rec, _) in enumerate(pipe.diarize_many([w for _, w in mine], sess_names=[r for r, _ in mine], overlap=not args.serial)):
        t = pipe.corpus_timings[k] if not args.serial else pipe.timings
        audio_s += t["audio_s"]
        print(f"[rank {rank}] {rec}: {t['audio_s']:.1f} s of audio: load {t['load_s']:.2f} s, device {t['device_s']:.2f} s, "
              f"host {t['host_s']:.2f} s", flush=True)
    wall = time.perf_counter() - t_all
    tmp = done_file(rank).with_suffix(".tmp")
    tmp.write_text(json.dumps({"files": [r for r, _ in mine], "audio_s": audio_s, "wall_s": wall}))
    os.replace(tmp, done_file(rank))
    if rank != 0:
        return None
    while not all(done_file(r).exists() for r in range(world)):
        time.sleep(0.2)
    done = [json.loads(done_file(r).read_text()) for r in range(world)]
    for r in range(world):
        done_file(r).unlink()
    summary = {"recordings": len(scp), "ranks": world, "audio_s": sum(d["audio_s"] for d in done),
               "wall_s_max_rank": max(d["wall_s"] for d in done), "precision": args.precision}
    summary["audio_seconds_per_s"] = summary["audio_s"] / summary["wall_s_max_rank"] if summary["wall_s_max_rank"] > 0 else None
    if args.ref_rttm:
        from diarizen_amd.der import score_set
        hyp = {rec: (Path(args.out_dir) / f"{rec}.rttm").read_text() for rec in scp}
        res = score_set(open(args.ref_rttm).read(), hyp, open(args.uem).read() if args.uem else None, collar=args.collar)
        summary["der_overall"] = res["overall"]
        summary["der_files"] = {u: {k: v for k, v in f.items() if k != "mapping"} for u, f in res["files"].items()}
        summary["missing_in_reference"] = res["missing_in_reference"]
        (Path(args.out_dir) / f"result_collar{args.collar:g}").write_text(json.dumps(summary, indent=1))
    print(json.dumps(summary))
    return summary


if __name__ == "__main__":
    main()
