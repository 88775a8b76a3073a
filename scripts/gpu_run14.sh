cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2n
timeout 900 python -m pytest tests/test_seg_gpu.py -m gpu -q -x -s -k "conv01" 2>&1 | grep -v "amdgpu.ids" | tail -4
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-e2e > gpurun_out/r2n/bench.json 2> gpurun_out/r2n/bench.err
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r2n/bench.json"))
    print(d["value"], d["ms_per_step"], d.get("unprofiled_ms_per_step"))
    for k in d["kernels"][:8]: print("  ",k["kernel"],k["launches"],k["ms_total"],k.get("tflops"),k.get("gbs"))
except Exception as e:
    print("failed", e); print(open("gpurun_out/r2n/bench.err").read()[-2500:])
PY
