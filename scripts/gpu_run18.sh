cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2r
timeout 1500 python -m pytest tests/test_seg_gpu.py tests/test_properties_gpu.py tests/test_pipeline_gpu.py -m gpu -q -x 2>&1 | grep -v "amdgpu.ids" | tail -5
for V in side noside; do
if [ $V = noside ]; then export DZN_NO_SIDE_WS=1; fi
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-e2e > gpurun_out/r2r/bench_$V.json 2> gpurun_out/r2r/bench_$V.err
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r2r/bench_$V.json"))
    print("$V", d["value"], d["ms_per_step"], d.get("unprofiled_ms_per_step"))
    for k in d["kernels"][:3]: print("  ",k["kernel"],k["launches"],k["ms_total"],k.get("tflops"),k.get("gbs"))
except Exception as e:
    print("failed", e); print(open("gpurun_out/r2r/bench_$V.err").read()[-2500:])
PY
done
