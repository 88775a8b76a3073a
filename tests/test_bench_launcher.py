"""`python bench.py --gpus N` must itself produce an N-rank job: the driver's SCALE run calls exactly
that command when no launcher wraps it.  CPU-only tests of the launcher path: the real spawn through
torch.distributed.run (gloo rendezvous on 127.0.0.1), the refusal when fewer than N devices are visible, and the
refusal when a launcher started a different number of ranks than --gpus says."""
from __future__ import annotations

import importlib.util
import json
import os
import subprocess
import sys
from pathlib import Path
from types import SimpleNamespace

ROOT = Path(__file__).resolve().parents[1]


def _clean_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "DZN_BENCH_ONE_DEVICE")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_gpus_flag_spawns_that_many_ranks():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--rendezvous-check"], env=_clean_env(),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["rendezvous"] == "ok" and out["n_gpus"] == 2
    assert sorted(x["rank"] for x in out["ranks"]) == [0, 1]
    assert sorted(x["local_rank"] for x in out["ranks"]) == [0, 1]          # one device index per rank
    assert len({x["pid"] for x in out["ranks"]}) == 2                         # two processes, not one rank counted twice
    assert out["sum_of_rank_ids"] == 3.0                                     # both contributed to the collective


def test_eight_rank_dry_run_binds_distinct_devices_under_a_visibility_permutation():
    """(r6) `--gpus 8` without a GPU: eight processes, rank -> cuda:LOCAL_RANK -> the physical device the
    HIP_VISIBLE_DEVICES permutation puts there (eight distinct ones), the 4 h recording's window blocks tile it with their
    sample ranges, and the packed all-gather returns every window in order on every rank."""
    env = dict(_clean_env(), HIP_VISIBLE_DEVICES="3,2,1,0,7,6,5,4", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--rendezvous-check"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["rendezvous"] == "ok" and out["n_gpus"] == 8 and out["problems"] == []
    by_rank = sorted(out["ranks"], key=lambda x: x["rank"])
    assert [x["torch_device"] for x in by_rank] == [f"cuda:{i}" for i in range(8)]
    assert [x["physical_device"] for x in by_rank] == ["3", "2", "1", "0", "7", "6", "5", "4"]
    assert out["visible_devices_var"] == "HIP_VISIBLE_DEVICES"
    C = out["strong_leg_windows"]
    assert C == 17991 and out["gathered_windows"] == C                       # 4 h, 8 s windows, 0.8 s step
    assert by_rank[0]["windows"][0] == 0 and by_rank[-1]["windows"][1] == C
    assert all(a["windows"][1] == b["windows"][0] for a, b in zip(by_rank, by_rank[1:]))
    assert out["sum_of_rank_ids"] == 36.0


def test_dry_run_refuses_two_ranks_on_one_physical_device():
    env = dict(_clean_env(), HIP_VISIBLE_DEVICES="0,0", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--rendezvous-check"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "share a physical device" in (r.stdout + r.stderr)


def test_world_size_must_equal_gpus():
    env = _clean_env()
    env.update(WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode != 0
    assert "WORLD_SIZE=3" in r.stderr and "--gpus 2" in r.stderr


def test_refuses_fewer_devices_than_ranks(monkeypatch, capsys):
    b = _bench()
    import torch
    calls = []
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: calls.append((cmd, env)) or 0)
    args = SimpleNamespace(gpus=2, rendezvous_check=False)
    assert b.self_launch(args, ["--gpus", "2"], one_device=False) != 0
    assert not calls, "must not start ranks it has no devices for"
    assert "needs 2 visible HIP devices" in capsys.readouterr().err
    # the one-device rehearsal is an explicit opt-in and does launch
    assert b.self_launch(args, ["--gpus", "2", "--steps", "1"], one_device=True) == 0
    (cmd, env), = calls
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=2" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "2", "--steps", "1"] and cmd[-5].endswith("bench.py")
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_no_gpu_is_a_loud_failure():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2"], env=_clean_env(), capture_output=True,
                       text=True, timeout=600)
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0 and "HIP device" in r.stderr


def test_power_sampler_reads_the_rocm_smi_csv(monkeypatch):
    """bench.py's `power` object: package power / cap / sclk parsed from `rocm-smi --showpower --showclocks --showmaxpower --csv`
    (the layout recorded on the MI355X box in profiles/r4_power_cap.txt); a missing or silent rocm-smi yields None, never an error."""
    import subprocess
    import bench
    line = next(ln for ln in (ROOT / "profiles" / "r4_power_cap.txt").read_text().splitlines() if ln.startswith("[f32h K1024 random] device"))
    csv_text = line.split("] ", 1)[1].replace(" | ", "\n")

    class Done:
        stdout = csv_text

    monkeypatch.setattr(subprocess, "run", lambda *a, **k: Done)
    ps = bench.PowerSampler()
    ps._poll()
    ps._poll()
    res = ps.result()
    assert res["cap_w"] == 1400.0 and res["package_w"] == {"mean": 1400.0, "max": 1400.0}
    assert res["sclk_mhz"]["min"] == 1920 and res["samples"] == 2

    def boom(*a, **k):
        raise FileNotFoundError("rocm-smi")

    monkeypatch.setattr(subprocess, "run", boom)
    silent = bench.PowerSampler()
    silent._poll()
    assert silent.result() is None


def test_power_sampler_picks_the_card_of_the_torch_device(monkeypatch):
    """(r5) the sampled rocm-smi card is the one whose PCI bus id is the torch device's (`rocm-smi --showbus --csv`);
    without a match the first card is sampled and the result says so."""
    import subprocess
    import torch
    import bench
    line = next(ln for ln in (ROOT / "profiles" / "r4_power_cap.txt").read_text().splitlines() if ln.startswith("[f32h K1024 random] device"))
    power_csv = line.split("] ", 1)[1].replace(" | ", "\n")
    hdr, row = power_csv.splitlines()[0], power_csv.splitlines()[1]
    two_cards = "\n".join([hdr, row.replace("card0", "card0", 1).replace("1400.0", "90.0", 1), row.replace("card0", "card3", 1)])

    class Done:
        def __init__(self, out):
            self.stdout = out

    def fake_run(cmd, *a, **k):
        return Done("device,PCI Bus\ncard0,0000:05:00.0\ncard3,0000:C5:00.0\n") if "--showbus" in cmd else Done(two_cards)

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(torch.cuda, "get_device_properties",
                        lambda d: SimpleNamespace(pci_domain_id=0, pci_bus_id=0xC5, pci_device_id=0))
    ps = bench.PowerSampler(device=0)
    assert ps.card == "card3" and ps.card_matched
    ps._poll()
    res = ps.result()
    assert res["card"] == "card3" and res["card_matched_by_pci_bus_id"] is True
    monkeypatch.setattr(torch.cuda, "get_device_properties",
                        lambda d: SimpleNamespace(pci_domain_id=0, pci_bus_id=0x11, pci_device_id=0))
    ps2 = bench.PowerSampler(device=0)
    assert ps2.card is None and not ps2.card_matched
    ps2._poll()
    assert ps2.result()["card"] == "card0" and ps2.result()["card_matched_by_pci_bus_id"] is False


def test_other_roof_names_the_binding_roof():
    """bench.py's second roof of a matrix class: the reduced contraction (268 GFLOP over 1.90 GB per launch) is below the ridge of
    its 1250 TFLOP/s peak, the f32h contraction (405 GFLOP over 2.39 GB) above the ridge of 833."""
    import bench
    mx = bench.other_roof({"flops": 268.168e9, "bytes": 1.90215e9, "ms": 0.985}, "mx")
    assert mx["binding_roof"] == "hbm" and abs(mx["alg_intensity_flop_per_byte"] - 141.0) < 0.1 and abs(mx["ridge_flop_per_byte"] - 156.2) < 0.1
    assert abs(mx["frac_of_hbm_peak"] - 1.90215e9 / 0.985e-3 / 8e12) < 1e-3
    h2 = bench.other_roof({"flops": 404.677e9, "bytes": 2.393908e9, "ms": 1.3288}, "f32h")
    assert h2["binding_roof"] == "mfma" and h2["ridge_flop_per_byte"] == 104.2
    assert bench.other_roof({"flops": 1e9, "bytes": 0.0, "ms": 1.0}, "f32h") == {}
