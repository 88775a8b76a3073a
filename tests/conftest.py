import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built_lib():
    """Make sure libdzn_hip.so exists (cross-compiles on CPU boxes)."""
    from diarizen_amd import _lib
    if not _lib.lib_path().exists():
        from diarizen_amd.build import build
        build()
    return _lib.load()


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("test marked gpu but no HIP device is visible")
    return torch.device("cuda:0")


def needs_bf16_mode(lib):
    """the bf16 engine mode is quarantined: only DZN_TUNING=1 builds of libdzn_hip.so carry it (dzn_version says so)"""
    if b"tuning build" not in lib.dzn_version():
        pytest.skip("bf16 engine mode: DZN_TUNING=1 builds only (quarantined, DESIGN.md §2)")


def pytest_sessionfinish(session, exitstatus):
    """CHECKED builds (DZN_HIP_LIB=.../libdzn_hip_checked.so, csrc/checked.h): a device-side bounds assertion that failed
    anywhere in the session fails the session, with the id / workgroup / detail of the first one."""
    try:
        from diarizen_amd import _lib
        if _lib._LIB is None or b"checked build" not in _lib._LIB.dzn_version():
            return
        import ctypes as C
        import torch
        if not torch.cuda.is_available():
            return
        w = (C.c_uint32 * 4)()
        n = _lib._LIB.dzn_checked_status(w, 0)
        line = f"[checked build] failed device-side checks: {n}" + (
            f" (first: id 0x{w[1]:x}, workgroup {w[2]}, detail {w[3]})" if n > 0 else "")
        print("\n" + line)
        os.makedirs("gpurun_out", exist_ok=True)
        with open("gpurun_out/checked_build_status.txt", "a") as f:
            f.write(line + "\n")
        if n != 0:
            session.exitstatus = 3
    except Exception as e:          # never mask the tests' own result
        print(f"\n[checked build] status query failed: {e}")
