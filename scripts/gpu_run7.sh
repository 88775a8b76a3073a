cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "amdgpu.ids" | tail -25
tail -5 gpurun_out/dist_nccl_2.log; tail -5 gpurun_out/dist_gloo_2.log
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-e2e > gpurun_out/r2g/bench.json 2> gpurun_out/r2g/bench.err
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r2g/bench.json"))
    print(d["value"], d["ms_per_step"], d.get("unprofiled_ms_per_step"))
    for k in d["kernels"][:14]: print("  ",k["kernel"],k["launches"],k["ms_total"],k.get("tflops"),k.get("gbs"))
except Exception as e:
    print("failed", e); print(open("gpurun_out/r2g/bench.err").read()[-2500:])
PY
