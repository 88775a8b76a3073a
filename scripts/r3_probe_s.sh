cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r3t}; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_properties_gpu.py tests/test_seg_gpu.py tests/test_f32h_grade_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; tail -4 $O/tests.txt
timeout 600 python bench.py --no-cpu-baseline --no-alt --no-e2e > $O/bench_default.json 2> $O/bench_default.err; python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["launches"])
for k in d.get("kernels", [])[:8]: print(k)
PY
DZN_GEMM_NO_PQ=1 timeout 600 python bench.py --no-cpu-baseline --no-alt --no-e2e > $O/bench_nopq.json 2>/dev/null; python - <<PY
import json
d=json.loads(open("$O/bench_nopq.json").read().strip().splitlines()[-1])
print("no-pq", d["value"], d["ms_per_step"])
PY
