"""Probe: do two engine handles on two HIP streams (alternate window batches) overlap the HBM-bound
kernels of one batch with the MFMA-bound kernels of the other?"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from bench import synth_recording
from diarizen_amd.configs import RESNET34, get_seg_config
from diarizen_amd.engine import Engine
from diarizen_amd.inference import WindowRunner
from testkit.weights import emb_state_dict, seg_state_dict
dev = torch.device("cuda:0")
cfg = get_seg_config("wavlm_large_s80_md"); sd = seg_state_dict(cfg, 0); esd = emb_state_dict(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
minutes = 10
wave = synth_recording(minutes * 60 * 16000).to(dev)
engs = [Engine(cfg, sd, RESNET34, esd, max_batch=B, max_samples=128000, precision="f32s", device=dev) for _ in range(2)]
runners = [WindowRunner(e, 8.0, 0.1, B) for e in engs]
views = runners[0].windows_view(wave)
C = views.shape[0]
def run(nstreams):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    cur = torch.cuda.current_stream()
    for s in streams: s.wait_stream(cur)
    outs = []
    for i, s0 in enumerate(range(0, C, B)):
        k = i % nstreams
        with torch.cuda.stream(streams[k]):
            r = runners[k]; eng = engs[k]
            chunk = views[s0:s0 + B].contiguous()
            _, ml = eng.segment(chunk, want_logp=False)
            filt, masks = eng.prepare_masks(ml, 11, True, r.min_num_frames, want_masks=True)
            outs.append((filt.clone(), eng.embed(chunk, masks).clone()))
    for s in streams: cur.wait_stream(s)
    torch.cuda.synchronize()
    return outs
for n in (1, 2, 1, 2):
    run(n)
    t0 = time.perf_counter(); o = run(n); dt = time.perf_counter() - t0
    print(f"batch {B} streams {n}: {dt*1e3:.1f} ms -> {minutes*60/dt:.1f} audio-s/s", flush=True)
