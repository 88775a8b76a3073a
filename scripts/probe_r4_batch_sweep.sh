# r4 probe: windows per launch x engine handles / streams (does keeping a launch's activations inside the 256 MB Infinity Cache,
# with the launch tails of several streams overlapping, beat one large launch?)
set -x
O=gpurun_out/r4bs; mkdir -p $O
for c in "192 3" "288 2" "128 4" "576 2" "576 1"; do
set -- $c
timeout 200 python bench.py --steps 2 --warmup 1 --batch $1 --streams $2 --no-alt --no-e2e --no-config1 --no-cpu-baseline --no-profile > $O/bench_b$1_s$2.json 2> $O/bench_b$1_s$2.err
python - <<PY
import json
try:
    d=json.loads(open("$O/bench_b$1_s$2.json").read().strip().splitlines()[-1])
    print("batch=$1 streams=$2", d["value"], d["ms_per_step"])
except Exception as e:
    print("batch=$1 streams=$2 failed", e, open("$O/bench_b$1_s$2.err").read()[-400:])
PY
done
