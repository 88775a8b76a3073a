set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "amdgpu.ids" | tail -25
timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/r2e/bench_default.json 2> gpurun_out/r2e/bench_default.err
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r2e/bench_default.json"))
    print(d["value"], d["ms_per_step"], d.get("unprofiled_ms_per_step"))
    print("e2e", d.get("e2e")); print("other", d.get("other_fp32_modes")); print("cpu", d.get("cpu_baseline")); print("extra", d.get("roofline_extra")); print("roofline", d.get("roofline"))
except Exception as e:
    print("failed", e); print(open("gpurun_out/r2e/bench_default.err").read()[-2500:])
PY
