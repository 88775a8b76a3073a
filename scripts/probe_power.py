"""Board power and clocks while the dominant contraction runs back to back (rocm-smi sampled from a thread):
    python scripts/probe_power.py [seconds per case]
cases: idle, f32h 128x128 on random data, the same on an all-zero A, K = 4096, a plain HBM copy."""
import subprocess
import sys
import threading
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from diarizen_amd import _lib, ops

dev = torch.device("cuda:0")
lib = _lib.load()
SEC = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0


def smi():
    try:
        out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks", "--showmaxpower", "--csv"], capture_output=True,
                             text=True, timeout=20).stdout
    except Exception as e:  # noqa: BLE001
        return f"rocm-smi failed: {e}"
    return " | ".join(line for line in out.splitlines() if line.strip())


def sample(tag, stop):
    time.sleep(SEC * 0.4)
    k = 0
    while not stop.is_set() and k < 3:
        print(f"[{tag}] {smi()}", flush=True)
        k += 1
        time.sleep(SEC * 0.15)


def case(tag, fn):
    stop = threading.Event()
    th = threading.Thread(target=sample, args=(tag, stop))
    th.start()
    t0 = time.perf_counter()
    n = 0
    torch.cuda.synchronize()
    while time.perf_counter() - t0 < SEC:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        n += 20
    dt = (time.perf_counter() - t0) / max(n, 1)
    stop.set()
    th.join()
    print(f"[{tag}] {dt * 1e6:.1f} us per launch over {n} launches", flush=True)


def gemm_case(M, N, K, zero=False):
    A = torch.zeros(M, K, device=dev) if zero else torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) * 0.05
    R = torch.randn(M, N, device=dev)
    W3 = ops.split_weights(W)
    W2h, cs = ops.split_weights_h2(W)
    am = ops.amax(A) if not zero else torch.ones(1, device=dev)
    out = torch.empty(M, N, device=dev)
    kw = {"W3": W3, "W2h": W2h, "col_scale": cs, "a_amax": am}
    return lambda: ops.gemm(A, W, C_out=out, R=R, precision=3, **kw)


print("[idle]", smi(), flush=True)
case("f32h K1024 random", gemm_case(149226, 1024, 1024))
case("f32h K1024 zeroA", gemm_case(149226, 1024, 1024, zero=True))
case("f32h K4096 random", gemm_case(37306, 1024, 4096))
src = torch.randn(256 * 1024 * 1024, device=dev)
dst = torch.empty_like(src)
case("copy 1 GiB", lambda: dst.copy_(src))
