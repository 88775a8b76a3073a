# round 3: the whole GPU suite + smoke + the default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r3l}; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 | cut -c1-400 ) > $O/all_gpu_tests.log 2>&1; cat $O/all_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | cut -c1-300
timeout 300 python scripts/probe_kernel_class.py 374 stem conv01 layernorm conv3x3 2>&1 | tail -6
( time timeout 900 python bench.py --no-alt ) > $O/bench_default.json 2> $O/bench_default.err
python - <<PY
import json
try:
    d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
    print("default", d["value"], d["ms_per_step"], d.get("e2e"))
    for k in d["kernels"][:10]: print("   ", {a:b for a,b in k.items() if a!='alg_bytes_per_launch'})
except Exception as e:
    print("bench failed", e); print(open("$O/bench_default.err").read()[-2500:])
PY
