cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2m
timeout 900 python -m pytest tests/test_seg_gpu.py -m gpu -q -x -s -k "conv01 or turn_taking or golden" 2>&1 | grep -v "amdgpu.ids" | tail -12
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-e2e > gpurun_out/r2m/bench.json 2> gpurun_out/r2m/bench.err
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r2m/bench.json"))
    print(d["value"], d["ms_per_step"], d.get("unprofiled_ms_per_step"))
    for k in d["kernels"][:12]: print("  ",k["kernel"],k["launches"],k["ms_total"],k.get("tflops"),k.get("gbs"))
except Exception as e:
    print("failed", e); print(open("gpurun_out/r2m/bench.err").read()[-2500:])
PY
DZN_NO_CONV01_FUSION=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-e2e > gpurun_out/r2m/bench_nofuse.json 2> gpurun_out/r2m/bench_nofuse.err
python - <<PY
import json
d=json.load(open("gpurun_out/r2m/bench_nofuse.json"))
print("nofuse", d["value"], d["ms_per_step"])
for k in d["kernels"]:
    if "conv0" in k["kernel"] or "128x80" in k["kernel"] or k["kernel"]=="layernorm": print("  ",k["kernel"],k["launches"],k["ms_total"],k.get("tflops"),k.get("gbs"))
PY
