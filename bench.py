"""bench.py — headline benchmark: audio-seconds/s of the wavlm-large-s80 sliding-window hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--precision f32|bf16] [--minutes 30]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the device hot path over ONE synthetic recording per rank
(BASELINE.json configs[2]: wavlm-large-s80, 30 min of 16 kHz mono, window 8 s, step 0.8 s ->
2241 windows, batch 256 by default: windows are independent, results do not depend on the batch): for every batch of windows  segmentation (WavLM + Conformer + powerset)
-> median filter + overlap-excluded masks -> ResNet34 embeddings (trunk shared by the 4 local
speakers), all through the C ABI of libdzn_hip.so, the recording already resident in HBM.  The
step ends with the hand-off the host clustering needs: u8 decisions + f32 embeddings copied to
the host (N=1) or all-gathered over RCCL (N>1; weak scaling: every rank owns its own 30 min).
Host clustering is a separate ("next") row and is not inside the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (in-situ HIP-event
timing of the dominant kernel class over the timed steps) and `cpu_baseline` (the oracle — a CPU
port of the reference arithmetic — on a bounded sample of the same workload).
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
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "f32s": 2500.0 / 6.0, "f32h": 2500.0 / 3.0}
DTYPE_NOTE = {"f32": "f32 (fp32 MFMA v_mfma_f32_16x16x4_f32)",
              "f32s": "f32 (operands split exactly into 3 bf16 terms, 6 bf16 MFMA products, fp32 accumulate)",
              "f32h": "f32 (operands split into 2 fp16 terms with exact power-of-two scaling = 22 significant bits, 3 fp16 MFMA "
                      "products, fp32 accumulate: the error-corrected '3xFP16/3xTF32' scheme; kernels without an fp16 variant "
                      "use the 3-term bf16 split)",
              "bf16": "bf16 (bf16 MFMA operands, fp32 accumulate / residual stream / norms)"}
PEAK_HBM_GBS = 8000.0


from diarizen_amd.synth import synth_recording  # noqa: E402


def pmc_traffic(kernel_class: str, args):
    """HBM bytes per launch of `kernel_class` from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE on this same command, gfx950 correction applied; scripts/pmc_traffic.py).  bench.py
    cannot run rocprofv3 on itself, so the figure is read from profiles/ and only when the workload
    matches the one the passes were taken on."""
    path = ROOT / "profiles" / f"r1_pmc_traffic_{args.precision}_30min_b{args.batch}.json"
    if not path.exists() or args.minutes != 30.0 or args.model != "wavlm_large_s80_md" or args.window != 8.0:
        return None
    table = json.loads(path.read_text())
    key = None
    for prefix, kern in (("gemm_f32s_", "gemm_split_kernel"), ("gemm_f32_", "gemm_glds_kernel")):
        if kernel_class.startswith(prefix):
            bm, bn = kernel_class[len(prefix):].split("x")
            key = next((k for k in table if k.startswith(f"{kern}<{bm}, {bn},")), None)
            break
    return table[key]["hbm_bytes_per_launch"] if key else None


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--precision", default=os.environ.get("DZN_BENCH_PRECISION", "f32s"),
                    choices=["f32s", "f32h", "f32", "bf16"],
                    help="f32s (default) and f32 are both fp32 arithmetic held to the strict parity tolerance; "
                         "f32 runs the contractions on the fp32 MFMA instead of the split bf16 products")
    ap.add_argument("--minutes", type=float, default=30.0)
    ap.add_argument("--window", type=float, default=8.0)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--model", default="wavlm_large_s80_md")
    ap.add_argument("--stage", default="full", choices=["full", "seg"],
                    help="seg = segmentation-only (BASELINE configs[1]: --model wavlm_base_s80_md --window 5 "
                         "--batch 32 --stage seg)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra fp32-MFMA-mode step reported beside f32s")
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
    from diarizen_amd.engine import Engine
    from diarizen_amd.inference import WindowRunner
    from diarizen_amd.weights import emb_state_dict, seg_state_dict   # seeded random init

    cfg = get_seg_config(args.model)
    sd = seg_state_dict(cfg, 0)
    esd = emb_state_dict(0)
    sr = 16000
    window = int(args.window * sr)
    eng = Engine(cfg, sd, RESNET34, esd, max_batch=args.batch, max_samples=window,
                 precision=args.precision, device=dev)
    runner = WindowRunner(eng, args.window, 0.1, args.batch)
    num_samples = int(args.minutes * 60 * sr)
    wave = synth_recording(num_samples, seed=3407 + rank).to(dev)
    n_windows = runner.num_windows(num_samples)
    audio_s = num_samples / sr

    full = args.stage == "full"

    def step():
        res = runner.run(wave, with_embeddings=full)
        if not full:
            return res.segmentations.cpu()
        if world > 1:
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
    if not args.no_profile:
        _lib.profile_enable(True)
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

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = world * audio_s * args.steps / dt
        roofline = None
        kernels = []
        if prof:
            tot_ms = sum(p["ms"] for p in prof)
            for p in sorted(prof, key=lambda p: -p["ms"]):
                e = {"kernel": p["name"], "launches": p["launches"], "ms_total": round(p["ms"], 3),
                     "share_of_profiled": round(p["ms"] / tot_ms, 4)}
                if p["flops"] > 0 and p["ms"] > 0:
                    e["tflops"] = round(p["flops"] / (p["ms"] * 1e-3) / 1e12, 2)
                if p["bytes"] > 0 and p["ms"] > 0:
                    e["gbs"] = round(p["bytes"] / (p["ms"] * 1e-3) / 1e9, 1)
                kernels.append(e)
            top = max(prof, key=lambda p: p["ms"])
            if top["flops"] > 0:
                prec = ("bf16" if "bf16" in top["name"] else "f32s" if "f32s" in top["name"] else
                        "f32h" if "f32h" in top["name"] else "f32")
                ach = top["flops"] / (top["ms"] * 1e-3) / 1e12
                roofline = {"kernel": top["name"], "bound": "mfma", "achieved": round(ach, 2),
                            "peak": round(PEAK_TFLOPS[prec], 1), "unit": "TFLOP/s",
                            "frac": round(ach / PEAK_TFLOPS[prec], 4),
                            "traffic": pmc_traffic(top["name"], args),
                            "launches": top["launches"],
                            "avg_launch_ms": round(top["ms"] / top["launches"], 4),
                            "alg_gflop_per_launch": round(top["flops"] / top["launches"] / 1e9, 3)}
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
                            "traffic": None, "launches": top["launches"],
                            "avg_launch_ms": round(top["ms"] / top["launches"], 4)}
        out = {
            "metric": "audio-seconds/s (RTF) for wavlm-large-s80 pipeline, 16 kHz mono",
            "value": round(value, 2), "unit": "audio-seconds/s", "rtf": round(1.0 / value, 6),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": DTYPE_NOTE[args.precision], "data": "synthetic",
            "config": {"workload": f"{args.model} hot path ({'segmentation + masks + ResNet34 embeddings' if full else 'segmentation only'}), "
                                   f"{args.minutes:g} min synthetic 16 kHz mono per GPU, window "
                                   f"{args.window:g} s, step {0.1 * args.window:g} s, {n_windows} windows, "
                                   f"batch {args.batch}; host clustering excluded",
                       "windows_per_step": n_windows, "batch": args.batch,
                       "weights": "seeded random init (no checkpoints offline)"},
            "windows_per_s": round(world * n_windows * args.steps / dt, 1),
            "roofline": roofline,
            "kernels": kernels,
        }
        if args.precision == "f32s" and world == 1 and not args.no_alt:
            # the same workload with the contractions on the fp32 MFMA (one extra untimed-warmup + timed step)
            eng2 = Engine(cfg, sd, RESNET34, esd, max_batch=args.batch, max_samples=window, precision="f32",
                          device=dev)
            r2 = WindowRunner(eng2, args.window, 0.1, args.batch)
            for timed in (False, True):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                res = r2.run(wave, with_embeddings=full)
                _ = (res.segmentations.cpu(), res.embeddings.cpu()) if full else res.segmentations.cpu()
                torch.cuda.synchronize()
                dt2 = time.perf_counter() - t1
            out["fp32_mfma_mode"] = {"value": round(audio_s / dt2, 2), "unit": "audio-seconds/s",
                                     "ms_per_step": round(dt2 * 1e3, 2), "steps": 1,
                                     "note": "--precision f32: same workload, contractions on v_mfma_f32_16x16x4_f32"}
        if not args.no_cpu_baseline and world == 1 and full:
            out["cpu_baseline"] = cpu_baseline(cfg, sd, esd, window, 0.1 * args.window)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
