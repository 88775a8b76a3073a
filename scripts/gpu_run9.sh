cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2i
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_seg_gpu.py tests/test_properties_gpu.py tests/test_emb_gpu.py tests/test_decisions_gpu.py -m gpu -q -x 2>&1 | grep -v "amdgpu.ids" | tail -12
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-e2e > gpurun_out/r2i/bench.json 2> gpurun_out/r2i/bench.err
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r2i/bench.json"))
    print(d["value"], d["ms_per_step"], d.get("unprofiled_ms_per_step"))
    for k in d["kernels"][:16]: print("  ",k["kernel"],k["launches"],k["ms_total"],k.get("tflops"),k.get("gbs"))
except Exception as e:
    print("failed", e); print(open("gpurun_out/r2i/bench.err").read()[-2500:])
PY
