#!/bin/bash
# which earlier leg of the bench process costs the three-stream configs[1] leg its overlap (3.7 k in line vs 4.56 k alone)
mkdir -p gpurun_out
B="--steps 1 --warmup 1 --no-cpu-baseline --no-power"
run() { name=$1; shift; timeout 400 python bench.py $B "$@" > gpurun_out/c1_$name.json 2> gpurun_out/c1_$name.err; python - <<PY
import json
d=json.loads(open("gpurun_out/c1_$name.json").read().strip().splitlines()[-1])
print("$name", d["config1"]["by_streams"], d["value"])
PY
}
run headline_only --no-alt --no-e2e --no-profile
run headline_profiled --no-alt --no-e2e
run with_alt --no-e2e
run with_e2e --no-alt
GPU_MAX_HW_QUEUES=8 run all_hwq8
