# round 3: the whole GPU suite + smoke + the bench's strong-scaling leg (one 4 h file) at world size 1
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r3full}; mkdir -p $O
export DZN_DECISION_WINDOWS=32
( time timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -15 | cut -c1-400 ) > $O/all_gpu_tests.log 2>&1; cat $O/all_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | cut -c1-300
timeout 900 python bench.py --steps 1 --warmup 1 --no-alt --no-cpu-baseline --strong-minutes 240 > $O/bench_with_strong_4h_leg.json 2> $O/bench_strong.err; python - <<PY
import json
d=json.loads(open("$O/bench_with_strong_4h_leg.json").read().strip().splitlines()[-1]); print(d["value"], d.get("strong_scaling_e2e"))
PY
