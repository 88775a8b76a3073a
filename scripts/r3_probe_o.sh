cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r3o}; mkdir -p $O
timeout 900 python -m pytest tests/test_seg_gpu.py tests/test_ops_gpu.py -x -q -m gpu -s -k "conv01 or golden or gemm or ragged" > $O/tests.txt 2>&1; tail -5 $O/tests.txt; grep "LN in the" $O/tests.txt
timeout 600 python bench.py --no-cpu-baseline --no-alt --no-e2e > $O/bench_default.json 2> $O/bench_default.err; python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"])
for k in d.get("kernels", [])[:14]: print(k)
PY
DZN_CONV01_NO_LN=1 timeout 600 python bench.py --no-cpu-baseline --no-alt --no-e2e > $O/bench_noln.json 2>/dev/null; python - <<PY
import json
d=json.loads(open("$O/bench_noln.json").read().strip().splitlines()[-1])
print("no-ln", d["value"], d["ms_per_step"])
PY
