set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_seg_gpu.py tests/test_emb_gpu.py tests/test_pipeline_gpu.py -m gpu -q -x 2>&1 | grep -v "amdgpu.ids" | tail -15
for P in f32h; do
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --precision $P > gpurun_out/r2d/bench_$P.json 2> gpurun_out/r2d/bench_$P.err
DZN_PROFILE_SHAPES=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt --precision $P > gpurun_out/r2d/shapes_$P.json 2>> gpurun_out/r2d/bench_$P.err
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r2d/bench_$P.json"))
    print("$P", d["value"], d["ms_per_step"])
    ks=d.get("kernels",[])
    print("  sum kernel ms/step", sum(k["ms_total"] for k in ks)/d["steps"])
    for k in ks[:24]: print("  ",k["kernel"],k["launches"],k["ms_total"],k.get("tflops"),k.get("gbs"))
except Exception as e:
    print("$P failed", e); print(open("gpurun_out/r2d/bench_$P.err").read()[-1500:])
PY
done
