"""bench.py — headline benchmark: audio-seconds/s of the wavlm-large-s80 sliding-window hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--precision f32h|f32s|f32|f16] [--minutes 30]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the device hot path over ONE synthetic recording per rank
(BASELINE.json configs[2]: wavlm-large-s80, 30 min of 16 kHz mono, window 8 s, step 0.8 s ->
2241 windows in 6 balanced launches of 374 (--batch 384 is the maximum): windows are independent, results do not depend
on the batch): for every batch of windows  segmentation (WavLM + Conformer + powerset)
-> median filter + overlap-excluded masks -> ResNet34 embeddings (trunk shared by the 4 local
speakers), all through the C ABI of libdzn_hip.so, the recording already resident in HBM.  The
step ends with the hand-off the host clustering needs: u8 decisions + f32 embeddings copied to
the host (N=1) or all-gathered over RCCL (N>1; weak scaling: every rank owns its own 30 min).
Host clustering is a separate ("next") row and is not inside the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (in-situ HIP-event
timing of the dominant kernel class over the timed steps), `cpu_baseline` (the oracle — a CPU
port of the reference arithmetic — on a bounded sample of the same workload), `parity` (what holds the path to the
reference), the same workload in the other arithmetic modes, `e2e` (the whole pipeline incl. host clustering) and, with
N > 1 ranks (or --strong-minutes), `strong_scaling_e2e`: ONE 4 h recording sharded over the ranks, end to end.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

# MI355X dense MFMA peaks (MI355X_MICROARCH.md).  "f32s" contractions run fp32 arithmetic as 6 bf16 MFMA
# products per block (exact 3-way operand split, csrc/gemm_split.hip): their ALGORITHMIC peak is bf16 / 6.
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "f32s": 2500.0 / 6.0, "f32h": 2500.0 / 3.0, "f16": 2500.0}
DTYPE_NOTE = {"f32": "f32 (fp32 MFMA v_mfma_f32_16x16x4_f32)",
              "f32s": "f32 (operands split exactly into 3 bf16 terms, 6 bf16 MFMA products, fp32 accumulate)",
              "f32h": "f32 (operands split into 2 fp16 terms with exact power-of-two scaling = 22 significant bits, 3 fp16 MFMA "
                      "products, fp32 accumulate: the error-corrected '3xFP16/3xTF32' scheme; kernels without an fp16 variant "
                      "use the 3-term bf16 split)",
              "bf16": "bf16 (bf16 MFMA operands, fp32 accumulate / residual stream / norms)",
              "f16": "f16 — REDUCED precision (BASELINE configs[4]): the f32h engine with the linear / positional-conv / "
                     "ResNet contractions keeping only the leading fp16 term (1 fp16 MFMA product, fp32 accumulate, per-window "
                     "power-of-two scaling); attention, the fused conv frontend and the 32-channel 3x3 convs keep 2 terms; "
                     "data, norms, softmax, residual stream fp32"}
PEAK_HBM_GBS = 8000.0
TRAFFIC_SOURCE = ("committed rocprofv3 --pmc passes of this same command (separate FETCH_SIZE / WRITE_SIZE runs, gfx950 FETCH_SIZE x 2 "
                  "correction; scripts/final_measure_r3.sh -> profiles/) — not measured in this run")
# what holds the results of this path to the reference's (tests/ -m gpu, all through the C ABI; fixtures made by
# oracle/gen_golden.py from the reference's own code)
PARITY_NOTE = {
    "bar": "fp32 modes: max |dlogp| <= 1e-3 vs the reference-made goldens, identical u8 decisions, embeddings cos >= 0.9999, RTTM "
           "text identical; integer / index work bit-exact",
    "tests": ["test_seg_gpu.py (4 configs x f32h/f32s/f32 vs reference goldens)", "test_emb_gpu.py (ResNet34 + fbank vs float64)",
              "test_decisions_gpu.py (0 argmax flips on 102 144 frames: profiles/r3_decision_parity.json)",
              "test_f32h_grade_gpu.py (all 69 (N, K) of this step vs float64: f32h <= 0.84 x the fp32-MFMA error, "
              "profiles/r3_f32h_grade_per_shape.json)",
              "test_host_ref.py (host stage == the reference's own aggregate / speaker_count / to_diarization / reconstruct / "
              "Binarize)", "test_host.py (reference clustering incl. forced min/max speakers, max_num_embeddings)",
              "test_ops_gpu.py::test_linkage_centroid_30k_equals_scipy_golden", "test_pipeline_gpu.py (RTTM == golden, streaming)",
              "test_properties_gpu.py (batch / shard bit-invariance)", "test_dist_gpu.py (2-rank pipeline RTTM == 1-GPU golden)"],
    "unpinned": ["kaldi fbank vs torchaudio (absent offline; == transformers.audio_utils to 1e-6 in float64)",
                 "pyannote.core 5.0.0 frame arithmetic / RTTM writer (absent offline)"]}
ALT_STEPS = 5          # timed steps of every comparison leg (other fp32 modes, reduced precision)


from testkit.synth import synth_recording  # noqa: E402


# in-situ profiler class -> kernel symbol in the rocprofv3 tables (scripts/pmc_summary.py writes profiles/*.json)
PMC_SYMBOLS = {"conv0_ln_gelu": "conv0_kernel", "conv01_fused": "conv01_fused_kernel",
               "attention_relpos_f32s": "attn_split_kernel<true, 3>", "attention_relpos_f32h": "attn_split_kernel<true, 2>",
               "attention_f32s": "attn_split_kernel<false, 3>", "attention_f32h": "attn_split_kernel<false, 2>",
               "conv3x3_c32_f32s": "conv3x3_c32_split_kernel<3>", "conv3x3_c32_f32h": "conv3x3_c32_split_kernel<2>",
               "layernorm": "layernorm_kernel<4", "row_stats": "row_stats_kernel<16>", "gate_ln_stats": "gate_stats_kernel"}


def pmc_table(args):
    """HBM bytes per launch / MfmaUtil per kernel from the committed rocprofv3 --pmc passes of this same command
    (separate passes per counter, gfx950 FETCH_SIZE correction: scripts/pmc_traffic.py, scripts/pmc_mfma.py).
    bench.py cannot run rocprofv3 on itself, so the table is read from profiles/ and only when the workload matches
    the one the passes were taken on; otherwise `traffic` is null."""
    found = sorted((ROOT / "profiles").glob(f"r*_pmc_{args.precision}_30min_b{args.batch}.json"))   # newest round last
    if not found or args.minutes != 30.0 or args.model != "wavlm_large_s80_md" or args.window != 8.0:
        return None
    table = json.loads(found[-1].read_text())
    table["_file"] = f"profiles/{found[-1].name}"
    return table


def pmc_lookup(table, kernel_class: str, field: str):
    if not table:
        return None
    key = None
    for tag, planes in (("gemm_f32h_pre_", 2), ("gemm_f32s_pre_", 3), ("gemm_f32h_", 2), ("gemm_f32s_", 3)):
        if kernel_class.startswith(tag):
            bm, bn = kernel_class[len(tag):].split("x")
            sym = "gemm_split_pre_kernel" if "_pre_" in tag else "gemm_split_kernel"
            for k in table:     # template arguments: <BM, BN, WGM, WGN, S, NP[, OCC]>
                if k.startswith(sym + "<"):
                    targs = [t.strip() for t in k[len(sym) + 1:].rstrip(">").split(",")]
                    if len(targs) >= 6 and targs[0] == bm and targs[1] == bn and targs[5] == str(planes):
                        key = k
                        break
            break
    else:
        sym = PMC_SYMBOLS.get(kernel_class)
        key = next((k for k in table if sym and k.startswith(sym)), None)
    return table[key].get(field) if key else None


def cpu_baseline(seg_cfg, sd, esd, window: int, step_s: float, budget_windows: int = 8):
    """The oracle (CPU port of the reference arithmetic, oracle/) on a bounded sample: B windows
    through segmentation + the embedding stage AS THE REFERENCE EXECUTES IT (one ResNet pass per
    (window, local speaker), PA/pipelines/speaker_diarization.py:295-353)."""
    from oracle import emb_model, seg_model
    from oracle.gen_golden import synth_wave
    threads = torch.get_num_threads()
    wave = synth_wave(budget_windows, window, 99)
    t0 = time.perf_counter()
    logp = seg_model.seg_forward(sd, seg_cfg, wave)
    ml = seg_model.to_multilabel(logp, seg_cfg)
    masks = ml.permute(0, 2, 1).contiguous()
    for s in range(masks.shape[1]):
        emb_model.emb_forward(esd, wave, masks[:, s])
    dt = time.perf_counter() - t0
    return {"value": round(budget_windows * step_s / dt, 4), "unit": "audio-seconds/s", "cores": threads,
            "kind": "port",
            "sample": f"{budget_windows} windows of {window} samples: oracle seg forward + 4 ResNet34 "
                      f"passes per window (as the reference executes), fp32 torch CPU, {dt:.1f} s"}


def shard_slice(num_samples: int, window: int, step: int, rank: int, world: int):
    """strong scaling: rank -> (first window, one-past-last window, first sample, samples incl. the window-length halo)"""
    from diarizen_amd.dist import shard_range
    from diarizen_amd.inference import window_plan
    n, last = window_plan(num_samples, window, step)
    c0, c1 = shard_range(n + int(last), rank, world)
    if c1 <= c0:
        return c0, c1, 0, 0
    return c0, c1, c0 * step, (c1 - c0 - 1) * step + window


def batches_note(n_windows: int, batch: int) -> str:
    nb = max(1, -(-n_windows // batch))
    return f"{nb} balanced launches of <= {-(-n_windows // nb)} windows (max batch {batch})"


def run_mode(cfg, sd, esd, wave, args, window, precision, full, dev):
    """one untimed + ALT_STEPS timed steps of the same workload in another arithmetic mode (reported beside the headline)"""
    from diarizen_amd.configs import RESNET34
    from diarizen_amd.engine import Engine
    from diarizen_amd.inference import WindowRunner
    eng = Engine(cfg, sd, RESNET34, esd, max_batch=args.batch, max_samples=window, precision=precision, device=dev)
    r = WindowRunner(eng, args.window, 0.1, args.batch)

    def one():
        res = r.run(wave, with_embeddings=full)
        _ = (res.segmentations.cpu(), res.embeddings.cpu()) if full else res.segmentations.cpu()
    one()                                   # warm-up (allocations, tables)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(ALT_STEPS):
        one()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t1) / ALT_STEPS
    eng.close()
    return dt


def e2e_leg(args, dev, wave_host):
    """BASELINE configs[2] names segmentation + embedding + AHC: ONE untimed-for-the-headline pass of the whole
    DiariZenPipeline (device stage, then speaker counting / AHC / reconstruction / RTTM on the host) over the same
    recording, with the seeded turn-taking weights so that the host stage sees non-degenerate decisions."""
    import copy
    import numpy as np
    from diarizen_amd.configs import get_seg_config
    from diarizen_amd.pipeline import DiariZenPipeline
    from testkit.weights import emb_state_dict, turn_taking_state_dict
    cfg = get_seg_config(args.model)
    conf = {"model": {"path": "diarizen.models.eend.model_wavlm_conformer.Model",
                      "args": {"wavlm_src": args.model, "wavlm_layer_num": cfg.wavlm_layer_num,
                               "wavlm_feat_dim": cfg.embed_dim, "chunk_size": int(args.window)}},
            "inference": {"args": {"seg_duration": args.window, "segmentation_step": 0.1, "batch_size": args.batch,
                                   "apply_median_filtering": True}},
            "clustering": {"args": {"method": "AgglomerativeClustering", "min_speakers": 1, "max_speakers": 20,
                                    "ahc_criterion": "distance", "ahc_threshold": 0.1, "min_cluster_size": 13}}}
    pipe = DiariZenPipeline(None, None, config=copy.deepcopy(conf), device=dev, precision=args.precision,
                            seg_state=turn_taking_state_dict(cfg, 0), emb_state=emb_state_dict(0))
    x = np.ascontiguousarray(wave_host.numpy())
    best = None
    for _ in range(2):                      # first pass warms allocations / tables
        t0 = time.perf_counter()
        seg, emb = pipe.device_stage(x)
        t1 = time.perf_counter()
        ann = pipe.host_stage(seg, emb, "bench")
        t2 = time.perf_counter()
        best = (t1 - t0, t2 - t1, ann)
    dev_s, host_s, ann = best
    audio_s = len(x) / 16000.0
    active = int((seg.sum(1) > 0).sum())
    return {"device_s": round(dev_s, 3), "host_s": round(host_s, 3), "upload_included": True,
            "audio_seconds_per_s": round(audio_s / (dev_s + host_s), 1),
            "speakers": len(ann.labels()), "rttm_lines": len(ann.to_rttm().splitlines()),
            "active_window_speakers": active,
            "note": "DiariZenPipeline device stage (host->HBM upload + segmentation + masks + embeddings + D2H) then host "
                    "counting + AHC (centroid linkage) + constrained assignment + reconstruction + RTTM; seeded "
                    "turn-taking weights"}


def strong_leg(args, dev, rank, world, minutes):
    """BASELINE configs[3]: ONE recording of `minutes` whose windows are sharded over the ranks, END TO END — each rank
    reads + uploads only its slice, runs segmentation + embeddings on its windows, one RCCL all-gather per tensor, then
    rank 0 runs the host stage (counting, AHC, assignment, reconstruction, RTTM).  The serial part (gather + host) is
    reported next to the sharded part so that Amdahl's bound on the end-to-end speed-up is visible in the line."""
    import copy
    import torch.distributed as dist
    from diarizen_amd.configs import get_seg_config
    from diarizen_amd.pipeline import DiariZenPipeline
    from testkit.weights import emb_state_dict, turn_taking_state_dict
    cfg = get_seg_config(args.model)
    conf = {"model": {"path": "diarizen.models.eend.model_wavlm_conformer.Model",
                      "args": {"wavlm_src": args.model, "wavlm_layer_num": cfg.wavlm_layer_num,
                               "wavlm_feat_dim": cfg.embed_dim, "chunk_size": int(args.window)}},
            "inference": {"args": {"seg_duration": args.window, "segmentation_step": 0.1, "batch_size": args.batch,
                                   "apply_median_filtering": True}},
            "clustering": {"args": {"method": "AgglomerativeClustering", "min_speakers": 1, "max_speakers": 20,
                                    "ahc_criterion": "distance", "ahc_threshold": 0.1, "min_cluster_size": 13}}}
    pipe = DiariZenPipeline(None, None, config=copy.deepcopy(conf), device=dev, precision=args.precision,
                            seg_state=turn_taking_state_dict(cfg, 0), emb_state=emb_state_dict(0))
    # the recording is a real RIFF file on local disk, so the timed region pays the byte-range decode a deployment pays
    # (audio.WavSource), not the synthesis: every rank synthesises the samples of ITS window block (1 / world of the
    # work) and writes them at their byte offset of one shared 16-bit PCM file; blocks overlap by one window of halo
    # with identical data
    import struct
    from diarizen_amd import dist as dz_dist
    from diarizen_amd.audio import WavSource
    from testkit.synth import synth_recording_range
    total = int(minutes * 60 * 16000)
    path = os.path.join(os.environ.get("TMPDIR", "/tmp"),      # one name for all ranks of THIS job
                        f"dzn_strong_{int(minutes)}min_{os.getuid()}_{os.environ.get('MASTER_PORT', '0')}.wav")
    if rank == 0:
        with open(path, "wb") as f:
            f.write(b"RIFF" + struct.pack("<I", 36 + 2 * total) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, 16000, 32000, 2, 16)
                    + b"data" + struct.pack("<I", 2 * total))
            f.truncate(44 + 2 * total)
    if world > 1:
        dist.barrier()
    r = pipe._runner
    c0, c1 = dz_dist.shard_range(r.num_windows(total), rank, world)
    lo, n = c0 * r.step, ((c1 - c0 - 1) * r.step + r.window if c1 > c0 else 0)
    n = max(0, min(n, total - lo))
    if n:
        x = synth_recording_range(lo, n, total=total)
        with open(path, "r+b") as f:
            f.seek(44 + 2 * lo)
            f.write((x.numpy() * 32767.0).astype("<i2").tobytes())
        del x
    if world > 1:
        dist.barrier()
    src = WavSource(path)
    best = None
    for _ in range(2):                       # first pass warms allocations / tables / RCCL channels
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        seg, emb = pipe.device_stage(src)    # slice read + upload + device stage + all-gather (+ D2H)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        ann = pipe.host_stage(seg, emb, "bench") if rank == 0 else None
        t2 = time.perf_counter()
        best = (t1 - t0, t2 - t1, ann)
    dev_s, host_s, ann = best
    if world > 1:
        tt = torch.tensor([dev_s], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dev_s = tt.item()
    if rank != 0:
        return None
    audio_s = src.num_samples / 16000.0
    try:
        os.remove(path)
    except OSError:
        pass
    return {"workload": f"{args.model} full pipeline, ONE {minutes:g} min synthetic recording, {seg.shape[0]} windows sharded over "
                        f"{world} rank(s) (contiguous blocks), all-gather, host stage on rank 0",
            "scaling": "strong", "n_gpus": world, "audio_s": audio_s,
            "sharded_s": round(dev_s, 3), "serial_host_s": round(host_s, 3),
            "e2e_audio_seconds_per_s": round(audio_s / (dev_s + host_s), 1),
            "device_only_audio_seconds_per_s": round(audio_s / dev_s, 1),
            "amdahl_serial_frac": round(host_s / (dev_s + host_s), 4),
            "speakers": len(ann.labels()), "rttm_lines": len(ann.to_rttm().splitlines()),
            "note": "sharded_s = max over ranks of [byte-range decode of the rank's block of a 16-bit PCM file + upload + segmentation + embeddings + RCCL all-gather + D2H]; "
                    "serial_host_s = rank 0's counting + AHC + assignment + reconstruction + RTTM; seeded turn-taking weights"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--precision", default=os.environ.get("DZN_BENCH_PRECISION", "f32h"),
                    choices=["f32h", "f32s", "f32", "bf16", "f16"],
                    help="f32h (default), f32s and f32 are fp32 arithmetic held to the same strict parity tolerance: "
                         "f32h = 2-term fp16 split / 3 MFMA products, f32s = 3-term bf16 split / 6 products, "
                         "f32 = the fp32 MFMA instruction")
    ap.add_argument("--minutes", type=float, default=30.0)
    ap.add_argument("--window", type=float, default=8.0)
    ap.add_argument("--batch", type=int, default=384,
                    help="maximum windows per launch (the runner balances: 2241 windows -> 6 launches of 374)")
    ap.add_argument("--model", default="wavlm_large_s80_md")
    ap.add_argument("--stage", default="full", choices=["full", "seg"],
                    help="seg = segmentation-only (BASELINE configs[1]: --model wavlm_base_s80_md --window 5 "
                         "--batch 32 --stage seg)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank owns its own recording of --minutes; strong (BASELINE configs[3]): ONE "
                         "recording of --minutes whose windows are sharded over the ranks (each rank uploads only its "
                         "slice + one window of halo)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra steps in the other fp32 modes")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end (device + host AHC) leg")
    ap.add_argument("--strong-minutes", type=float, default=None,
                    help="length of the ONE recording of the strong-scaling end-to-end leg (BASELINE configs[3]: 240); "
                         "default: 240 when --gpus > 1, off at 1 GPU")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product has no CPU path)")
    if os.environ.get("DZN_BENCH_ONE_DEVICE"):       # debug only: exercise the N>1 code path on a 1-GPU box
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("DZN_BENCH_BACKEND", "nccl")   # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    from diarizen_amd import _lib
    from diarizen_amd.configs import RESNET34, get_seg_config
    from diarizen_amd.dist import gather_windows
    from diarizen_amd.engine import Engine
    from diarizen_amd.inference import WindowRunner
    from testkit.synth import synth_recording_range
    from testkit.weights import emb_state_dict, seg_state_dict   # seeded random init

    cfg = get_seg_config(args.model)
    sd = seg_state_dict(cfg, 0)
    esd = emb_state_dict(0)
    sr = 16000
    window = int(args.window * sr)
    eng = Engine(cfg, sd, RESNET34, esd, max_batch=args.batch, max_samples=window,
                 precision=args.precision, device=dev)
    runner = WindowRunner(eng, args.window, 0.1, args.batch)
    num_samples = int(args.minutes * 60 * sr)
    audio_s = num_samples / sr
    strong = args.scaling == "strong" and world > 1
    if strong:
        # ONE recording (seed 3407) for all ranks; this rank generates and uploads only the samples its windows touch
        c0, c1, s0, ns = shard_slice(num_samples, runner.window, runner.step, rank, world)
        wave_host = synth_recording_range(s0, ns, total=num_samples, seed=3407)
        n_windows = runner.num_windows(num_samples)
    else:
        wave_host = synth_recording(num_samples, seed=3407 + rank)
        n_windows = runner.num_windows(num_samples)
    wave = wave_host.to(dev)

    full = args.stage == "full"

    def step():
        res = runner.run(wave, with_embeddings=full) if wave.numel() else None
        if not full:
            return res.segmentations.cpu()
        if world > 1:
            if strong:      # ranks hold different window counts: padded all-gather in window order (dist.py)
                S, L = eng.seg.max_speakers_per_chunk, runner.num_frames
                seg_l = res.segmentations if res is not None else torch.empty((0, L, S), device=dev, dtype=torch.uint8)
                emb_l = res.embeddings if res is not None else torch.empty((0, S, eng.emb.embed_dim), device=dev)
                seg_g, emb_g = gather_windows(seg_l, emb_l)
                return (seg_g.cpu(), emb_g.cpu()) if rank == 0 else None
            if dist.get_backend() != "nccl":       # single-GPU rehearsal of the N > 1 path (gloo: host staging)
                seg_g, emb_g = gather_windows(res.segmentations, res.embeddings)
                return (seg_g.cpu(), emb_g.cpu()) if rank == 0 else None
            segs = [torch.empty_like(res.segmentations) for _ in range(world)]
            embs = [torch.empty_like(res.embeddings) for _ in range(world)]
            dist.all_gather(segs, res.segmentations)
            dist.all_gather(embs, res.embeddings)
            if rank == 0:
                return torch.cat(segs).cpu(), torch.cat(embs).cpu()
            return None
        return res.segmentations.cpu(), res.embeddings.cpu()

    for _ in range(args.warmup):
        step()
    # the in-situ HIP-event profiler runs INSIDE the timed steps (two event records per launch on the launch stream;
    # measured cost 1 % of the step, reported as `unprofiled_ms_per_step` from one extra step below)
    if not args.no_profile:
        # one untimed profiled step sizes the event pool: the timed region must not create HIP events
        _lib.profile_enable(True)
        step()
        launches = sum(p["launches"] for p in _lib.profile_collect())
        _lib.profile_reserve(2 * launches * args.steps + 1024)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    prof = [] if args.no_profile else _lib.profile_collect()
    _lib.profile_enable(False)
    if dist is not None:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    unprofiled_ms = (time.perf_counter() - t1) * 1e3

    strong_min = args.strong_minutes if args.strong_minutes is not None else (240.0 if world > 1 else 0.0)
    strong_res = None
    if strong_min > 0 and full and not args.no_e2e and world > 1:     # collective: every rank takes part
        del eng, runner, wave
        torch.cuda.empty_cache()
        strong_res = strong_leg(args, dev, rank, world, strong_min)
        eng = None

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        total_audio = audio_s if strong else world * audio_s
        value = total_audio * args.steps / dt
        roofline = None
        kernels = []
        extra = {}
        if prof:
            tot_ms = sum(p["ms"] for p in prof)
            for p in sorted(prof, key=lambda p: -p["ms"]):
                e = {"kernel": p["name"], "launches": p["launches"], "ms_total": round(p["ms"], 3),
                     "share_of_profiled": round(p["ms"] / tot_ms, 4)}
                if p["flops"] > 0 and p["ms"] > 0:
                    e["tflops"] = round(p["flops"] / (p["ms"] * 1e-3) / 1e12, 2)
                if p["bytes"] > 0 and p["ms"] > 0:
                    e["gbs"] = round(p["bytes"] / (p["ms"] * 1e-3) / 1e9, 1)
                    e["alg_bytes_per_launch"] = int(p["bytes"] / p["launches"])
                kernels.append(e)
            top = max(prof, key=lambda p: p["ms"])
            traffic = pmc_table(args)
            if top["flops"] > 0:
                prec = ("bf16" if "bf16" in top["name"] else "f32s" if "f32s" in top["name"] else
                        "f32h" if "f32h" in top["name"] else "f16" if "f16" in top["name"] else "f32")
                ach = top["flops"] / (top["ms"] * 1e-3) / 1e12
                roofline = {"kernel": top["name"], "bound": "mfma", "achieved": round(ach, 2),
                            "peak": round(PEAK_TFLOPS[prec], 1), "unit": "TFLOP/s",
                            "frac": round(ach / PEAK_TFLOPS[prec], 4),
                            "traffic": pmc_lookup(traffic, top["name"], "hbm_bytes_per_launch"),
                            "traffic_source": f"{TRAFFIC_SOURCE} [{traffic.get('_file')}]" if traffic else None,
                            "launches": top["launches"],
                            "avg_launch_ms": round(top["ms"] / top["launches"], 4),
                            "alg_gflop_per_launch": round(top["flops"] / top["launches"] / 1e9, 3),
                            "alg_bytes_per_launch": int(top["bytes"] / top["launches"]) if top["bytes"] > 0 else None}
                if prec == "f32h":
                    roofline["note"] = ("achieved = algorithmic fp32 flops / s; every 16x16x32 block costs 3 fp16 MFMAs "
                                        "(hi*hi + hi*lo + lo*hi), so peak = fp16 dense peak 2500 / 3; executed MFMA rate = "
                                        "3 x achieved")
                    roofline["executed_tflops"] = round(3 * ach, 1)
                if prec == "f32s":
                    roofline["note"] = ("achieved = algorithmic fp32 flops / s; every 16x16x32 block costs 6 bf16 "
                                        "MFMAs, so peak = bf16 dense peak 2500 / 6; executed MFMA rate = 6 x achieved")
                    roofline["executed_tflops"] = round(6 * ach, 1)
            else:
                ach = top["bytes"] / (top["ms"] * 1e-3) / 1e9
                roofline = {"kernel": top["name"], "bound": "hbm", "achieved": round(ach, 1),
                            "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4),
                            "traffic": pmc_lookup(traffic, top["name"], "hbm_bytes_per_launch"),
                            "launches": top["launches"],
                            "avg_launch_ms": round(top["ms"] / top["launches"], 4)}
            # north_star asks for two more figures: MFMA utilisation on the attention contractions and HBM GB/s on
            # the conv frontend.  Rates are live (HIP events of this run); MfmaUtil / PMC bytes come from the
            # committed rocprofv3 --pmc passes of the same command (profiles/, scripts/final_measure_r3.sh).
            for key, names in (("attention", ("attention_relpos_f32s", "attention_relpos_f32h", "attention_relpos_f32")),
                               ("conv_frontend", ("conv0_ln_gelu", "conv01_fused"))):
                p = next((p for p in prof if p["name"] in names), None)
                if p is None:
                    continue
                e = {"kernel": p["name"], "avg_launch_ms": round(p["ms"] / p["launches"], 4)}
                if p["flops"] > 0:
                    e["tflops"] = round(p["flops"] / (p["ms"] * 1e-3) / 1e12, 2)
                if p["bytes"] > 0:
                    e["alg_bytes_per_launch"] = int(p["bytes"] / p["launches"])
                    e["gbs"] = round(p["bytes"] / (p["ms"] * 1e-3) / 1e9, 1)
                    e["frac_of_hbm_peak"] = round(e["gbs"] / PEAK_HBM_GBS, 4)
                e["pmc_hbm_bytes_per_launch"] = pmc_lookup(traffic, p["name"], "hbm_bytes_per_launch")
                e["pmc_mfma_util_pct"] = pmc_lookup(traffic, p["name"], "mfma_util_pct")
                if p["name"] == "conv01_fused":
                    e["note"] = ("conv0 + LayerNorm + GELU + conv1 fused (frontend_fused.hip): conv0's 13.2 GB / launch of "
                                 "activations never reach HBM (was conv0_ln_gelu: 13.2 GB written at 2.7-2.8 TB/s = 0.34 of "
                                 "peak, then re-read by conv1); the kernel is MFMA/VALU-bound, its HBM traffic is the "
                                 "waveform in + conv1's raw output out (alg_bytes_per_launch; PMC beside it)")
                    e.pop("frac_of_hbm_peak", None)
                extra[key] = e
            extra["kernel_ms_per_step"] = round(tot_ms / args.steps, 2)
            extra["non_kernel_frac"] = round(1.0 - tot_ms / args.steps / ms_per_step, 4)
        out = {
            "metric": "audio-seconds/s (RTF) for wavlm-large-s80 pipeline, 16 kHz mono",
            "value": round(value, 2), "unit": "audio-seconds/s", "rtf": round(1.0 / value, 6),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 2), "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": DTYPE_NOTE[args.precision], "data": "synthetic",
            "config": {"workload": f"{args.model} hot path ({'segmentation + masks + ResNet34 embeddings' if full else 'segmentation only'}), "
                                   f"{args.minutes:g} min synthetic 16 kHz mono {'sharded over the ranks' if strong else 'per GPU'}, window "
                                   f"{args.window:g} s, step {0.1 * args.window:g} s, {n_windows} windows, "
                                   f"batch {args.batch}; host clustering excluded from `value` (see `e2e`)",
                       "windows_per_step": n_windows, "batch": args.batch, "launches": batches_note(n_windows, args.batch),
                       "weights": "seeded random init (no checkpoints offline)"},
            "windows_per_s": round((1 if strong else world) * n_windows * args.steps / dt, 1),
            "unprofiled_ms_per_step": round(unprofiled_ms, 2),
            "roofline": roofline,
            "roofline_extra": extra,
            "parity": PARITY_NOTE,
            "kernels": kernels,
        }
        if world == 1 and not args.no_alt and args.precision in ("f32h", "f32s"):
            # the same workload in the other fp32 modes (strict parity tests run in all three)
            alt = {}
            for prec in ("f32s", "f32h", "f32"):
                if prec == args.precision:
                    continue
                dt2 = run_mode(cfg, sd, esd, wave, args, window, prec, full, dev)
                alt[prec] = {"value": round(audio_s / dt2, 2), "unit": "audio-seconds/s",
                             "ms_per_step": round(dt2 * 1e3, 2), "steps": ALT_STEPS, "dtype": DTYPE_NOTE[prec]}
            out["other_fp32_modes"] = alt
            out["fp32_mfma_mode"] = alt["f32"]
            # REDUCED precision (BASELINE configs[4] "fp16"): reported beside the headline, never as `value`
            dt2 = run_mode(cfg, sd, esd, wave, args, window, "f16", full, dev)
            out["reduced_precision_mode"] = {"f16": {"value": round(audio_s / dt2, 2), "unit": "audio-seconds/s",
                                                     "ms_per_step": round(dt2 * 1e3, 2), "steps": ALT_STEPS,
                                                     "dtype": DTYPE_NOTE["f16"],
                                                     "parity": "reduced-precision bar (tests/test_seg_gpu.py::"
                                                               "test_seg_f16_within_tolerance: max |dlogp| <= 5e-2, "
                                                               "argmax >= 99.5 %; embeddings cos >= 0.999)"}}
        if world == 1 and full and not args.no_e2e and args.minutes <= 60:
            eng = None
            torch.cuda.empty_cache()
            out["e2e"] = e2e_leg(args, dev, wave_host)
        if strong_min > 0 and full and not args.no_e2e and world == 1:     # one GPU: the same leg as the N > 1 runs, for the curve
            eng = None
            torch.cuda.empty_cache()
            strong_res = strong_leg(args, dev, rank, world, strong_min)
        if strong_res is not None:
            out["strong_scaling_e2e"] = strong_res
        if not args.no_cpu_baseline and world == 1 and full:
            out["cpu_baseline"] = cpu_baseline(cfg, sd, esd, window, 0.1 * args.window)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
