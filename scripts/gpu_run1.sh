set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r2a/pytest.log
cat gpurun_out/r2a/pytest.log
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt > gpurun_out/r2a/bench_fold.json 2> gpurun_out/r2a/bench_fold.err
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-profile > gpurun_out/r2a/bench_fold_noprof.json 2>> gpurun_out/r2a/bench_fold.err
DZN_NO_LN_FOLD=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt > gpurun_out/r2a/bench_nofold.json 2> gpurun_out/r2a/bench_nofold.err
for f in fold fold_noprof nofold; do python - <<PY
import json
d=json.load(open("gpurun_out/r2a/bench_$f.json"))
print("$f", d["value"], d["ms_per_step"])
ks=d.get("kernels",[])
tot=sum(k["ms_total"] for k in ks)
print("  sum kernel ms/step", tot/d["steps"])
for k in ks[:14]: print("  ",k["kernel"],k["launches"],k["ms_total"],k.get("tflops"),k.get("gbs"))
PY
done
