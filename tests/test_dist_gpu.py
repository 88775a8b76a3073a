"""(e) multi-GPU path on real hardware before an 8-GPU node sees it: the window sharding + padded all-gather of
diarizen_amd/dist.py through the `nccl` backend (RCCL on ROCm).  World size 1 always; world size 2 on two devices
through RCCL wherever the box has them (a failure there fails the test); on a 1-GPU box both ranks share the device, RCCL
refuses duplicate devices, and the 2-rank run stages through gloo (the RCCL path then stays covered at world size 1)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
WORKER = os.path.join(os.path.dirname(__file__), "_dist_worker.py")


def _run(nproc, backend, port):
    env = dict(os.environ, DZN_TEST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), WORKER]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    log = os.path.join(os.path.dirname(os.path.dirname(__file__)), "gpurun_out")
    os.makedirs(log, exist_ok=True)
    with open(os.path.join(log, f"dist_{backend}_{nproc}.log"), "w") as f:
        f.write(r.stdout[-6000:] + "\n---- stderr ----\n" + r.stderr[-12000:])
    return r


def test_window_shard_gather_rccl_world1(built_lib, gpu):
    r = _run(1, "nccl", 29611)
    assert r.returncode == 0 and "DIST_OK backend=nccl world=1" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
    assert "DIST_RTTM_OK backend=nccl world=1" in r.stdout


def test_window_shard_gather_two_ranks(built_lib, gpu):
    """two ranks.  On a box with >= 2 devices each rank takes its own (tests/_dist_worker.py: LOCAL_RANK) and the all-gather
    MUST run on RCCL — a failure there is a failure of the N > 1 path, never papered over by gloo.  On a
    1-GPU box RCCL refuses the duplicate device; only then both ranks share device 0 and stage through gloo."""
    import torch
    r = _run(2, "nccl", 29612)
    if torch.cuda.device_count() >= 2:
        assert r.returncode == 0 and "DIST_OK backend=nccl world=2" in r.stdout and "devices=2" in r.stdout, \
            (r.stdout[-2000:], r.stderr[-3000:])
        assert "DIST_RTTM_OK backend=nccl world=2" in r.stdout
        return
    if r.returncode == 0 and "DIST_OK backend=nccl world=2" in r.stdout:
        assert "DIST_RTTM_OK backend=nccl world=2" in r.stdout
        return
    r2 = _run(2, "gloo", 29613)
    assert r2.returncode == 0 and "DIST_OK backend=gloo world=2" in r2.stdout, (r.stderr[-1500:], r2.stdout[-1500:], r2.stderr[-3000:])
    # (r3) the 2-rank rehearsal of the WHOLE pipeline (range decode, gather with the partition assertion, host stage on
    # rank 0) reproduces the 1-GPU RTTM golden
    assert "DIST_RTTM_OK backend=gloo world=2" in r2.stdout, r2.stdout[-1500:]
