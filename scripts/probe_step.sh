# quick check of a kernel change: the per-kernel + segmentation tests, then the default step (no comparison legs)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-probe}; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_seg_gpu.py tests/test_properties_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
timeout 600 python bench.py --no-cpu-baseline --no-alt --no-e2e ${2:-} > $O/bench_default.json 2> $O/bench_default.err; python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["launches"])
for k in d.get("kernels", [])[:8]: print({a: b for a, b in k.items() if a != "alg_bytes_per_launch"})
PY
