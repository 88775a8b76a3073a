cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2q
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_seg_gpu.py tests/test_properties_gpu.py tests/test_pipeline_gpu.py -m gpu -q -x 2>&1 | grep -v "amdgpu.ids" | tail -6
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-e2e > gpurun_out/r2q/bench.json 2> gpurun_out/r2q/bench.err
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r2q/bench.json"))
    print(d["value"], d["ms_per_step"], d.get("unprofiled_ms_per_step"), d["config"]["launches"])
    for k in d["kernels"][:12]: print("  ",k["kernel"],k["launches"],k["ms_total"],k.get("tflops"),k.get("gbs"))
except Exception as e:
    print("failed", e); print(open("gpurun_out/r2q/bench.err").read()[-2500:])
PY
