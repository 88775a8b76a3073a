"""Build libdzn_hip.so (gfx950) in-tree with hipcc.

    python -m diarizen_amd.build            # incremental
    python -m diarizen_amd.build --force

hipcc cross-compiles for gfx950 without a GPU.  The shared object is written to
diarizen_amd/lib/ (git-ignored, but shipped to the GPU box by gpurun).  The library only
depends on the HIP runtime (libamdhip64): inside a PyTorch-ROCm process the already loaded
runtime of torch is reused, so device pointers / streams are shared with torch tensors.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent
CSRC = ROOT / "csrc"
OBJ = ROOT / "build"
LIBDIR = ROOT / "lib"
LIB = LIBDIR / "libdzn_hip.so"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wno-unused-result",
         "-ffp-contract=off",
         # the fused epilogue (4 x 6 accumulator blocks x erf-GELU ...) is fully unrolled by pragma; past the
         # default cost cap clang silently keeps the loop and the accumulators go to scratch (tests/test_host.py)
         "-mllvm", "-pragma-unroll-threshold=65536"]
if os.environ.get("DZN_TUNING"):     # probe / ablation kernel instantiations for scripts/bench_gemm_cfgs.py
    FLAGS.append("-DDZN_TUNING")


def _headers_mtime() -> float:
    hs = list(CSRC.glob("*.h")) + list((ROOT.parent / "include").glob("*.h"))
    return max(h.stat().st_mtime for h in hs)


def _compile(src: Path, force: bool, hdr_mtime: float) -> Path:
    obj = OBJ / (src.name + ".o")
    if (not force and obj.exists() and obj.stat().st_mtime > src.stat().st_mtime
            and obj.stat().st_mtime > hdr_mtime):
        return obj
    cmd = [HIPCC, *FLAGS, "-c", str(src), "-o", str(obj)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stderr[-4000:]}")
    return obj


def build_checked(force: bool = False, verbose: bool = False) -> Path:
    """python -m diarizen_amd.build --checked: the same sources with -DDZN_CHECKED (csrc/checked.h: device-side bounds
    assertions in the hand-scheduled kernels) into lib/libdzn_hip_checked.so, objects under build_checked/.  Select it with
    DZN_HIP_LIB=<path> (scripts/run_checked.sh runs the GPU kernel / segmentation tests under it)."""
    global OBJ, LIB
    saved = (OBJ, LIB, list(FLAGS))
    OBJ, LIB = ROOT / "build_checked", LIBDIR / "libdzn_hip_checked.so"
    FLAGS.append("-DDZN_CHECKED")
    try:
        return build(force, verbose, harness=False)
    finally:
        OBJ, LIB = saved[0], saved[1]
        FLAGS[:] = saved[2]


def build(force: bool = False, verbose: bool = False, harness: bool = True) -> Path:
    OBJ.mkdir(exist_ok=True)
    LIBDIR.mkdir(exist_ok=True)
    srcs = sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.cpp")))
    if not srcs:
        raise RuntimeError("no sources under csrc/")
    hdr = _headers_mtime()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, hdr), srcs))
    newest = max(o.stat().st_mtime for o in objs)
    if force or not LIB.exists() or LIB.stat().st_mtime < newest:
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(LIB),
               *map(str, objs)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
        if verbose:
            print(f"linked {LIB}")
    if harness:
        build_c_harness(force, verbose)
    return LIB


C_HARNESS_SRC = ROOT.parent / "tests" / "c_abi_smoke.c"
C_HARNESS = OBJ / "c_abi_smoke"


def build_c_harness(force: bool = False, verbose: bool = False):
    """tests/c_abi_smoke.c with gcc as plain C11 against include/dzn.h + libdzn_hip.so: proves the boundary is a C ABI
    (no C++ / torch types) and gives tests/test_properties_gpu.py a Python-free driver.  Optional: skipped when the
    source or gcc is absent."""
    import shutil
    gcc = shutil.which("gcc")
    if not C_HARNESS_SRC.exists() or gcc is None:
        return None
    if (not force and C_HARNESS.exists() and C_HARNESS.stat().st_mtime > C_HARNESS_SRC.stat().st_mtime
            and C_HARNESS.stat().st_mtime > _headers_mtime()):
        return C_HARNESS
    cmd = [gcc, "-std=c11", "-O1", "-Wall", f"-I{ROOT.parent / 'include'}", "-I/opt/rocm/include", str(C_HARNESS_SRC),
           "-o", str(C_HARNESS), f"-L{LIBDIR}", "-ldzn_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
           "-Wl,-rpath,$ORIGIN/../lib", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        # the harness is a test driver: its failure must not fail the product build (libdzn_hip.so is already linked)
        import warnings
        warnings.warn(f"c_abi_smoke.c did not build (tests/test_properties_gpu.py::test_c_abi_without_python will skip):\n"
                      f"{r.stderr[-1500:]}")
        return None
    if verbose:
        print(f"built {C_HARNESS}")
    return C_HARNESS


if __name__ == "__main__":
    if "--checked" in sys.argv:
        p = build_checked(force="--force" in sys.argv, verbose=True)
    else:
        p = build(force="--force" in sys.argv, verbose=True)
    print(p)
