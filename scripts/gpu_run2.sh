set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "amdgpu.ids" | tail -60 > gpurun_out/r2b/pytest.log
tail -45 gpurun_out/r2b/pytest.log
for P in f32s f32h; do
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --precision $P > gpurun_out/r2b/bench_$P.json 2> gpurun_out/r2b/bench_$P.err
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r2b/bench_$P.json"))
    print("$P", d["value"], d["ms_per_step"])
    ks=d.get("kernels",[])
    print("  sum kernel ms/step", sum(k["ms_total"] for k in ks)/d["steps"])
    for k in ks[:16]: print("  ",k["kernel"],k["launches"],k["ms_total"],k.get("tflops"),k.get("gbs"))
except Exception as e:
    print("$P failed", e); print(open("gpurun_out/r2b/bench_$P.err").read()[-1500:])
PY
done
