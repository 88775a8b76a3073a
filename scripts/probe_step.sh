# quick check of a kernel change: a subset of the GPU tests (default: per-kernel + segmentation + properties), then the
# default step without the comparison legs.   usage: probe_step.sh OUTDIR ["test files"] [bench flags]
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-probe}; mkdir -p $O
T=${2:-"tests/test_ops_gpu.py tests/test_seg_gpu.py tests/test_properties_gpu.py"}
timeout 1200 python -m pytest $T -x -q -m gpu > $O/tests.txt 2>&1; tail -5 $O/tests.txt
timeout 600 python bench.py --no-cpu-baseline --no-alt ${3:-} > $O/bench_default.json 2> $O/bench_default.err; python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["launches"], d.get("e2e"))
for k in d.get("kernels", [])[:6]: print({a: b for a, b in k.items() if a != "alg_bytes_per_launch"})
PY
