#!/bin/bash
# r5 call 9: the host stage on its own stream / arena (csrc/linkage.hip, postprocess.DevicePost), DiariZenPipeline.diarize_many,
# bench steps with the host stage overlapped (A/B against --no-overlap on one box)
mkdir -p gpurun_out
set -x
timeout 900 python -m pytest tests/test_pipeline_gpu.py -q -x -k "diarize_many or host_stage_30min or rttm_equal" > gpurun_out/r5_overlap_tests.log 2>&1
tail -5 gpurun_out/r5_overlap_tests.log
timeout 600 python -m pytest tests/test_ops_gpu.py -q -k "linkage or cdist" >> gpurun_out/r5_overlap_tests.log 2>&1
tail -3 gpurun_out/r5_overlap_tests.log
B="--steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-e2e --no-config1"
timeout 400 python bench.py $B > gpurun_out/r5_bench_overlap.json 2> gpurun_out/r5_bench_overlap.err
timeout 400 python bench.py $B --no-overlap > gpurun_out/r5_bench_no_overlap.json 2> gpurun_out/r5_bench_no_overlap.err
timeout 400 python bench.py $B > gpurun_out/r5_bench_overlap2.json 2>> gpurun_out/r5_bench_overlap.err
python - <<'PY'
import json
for f in ("r5_bench_overlap", "r5_bench_no_overlap", "r5_bench_overlap2"):
    try:
        d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], d["device_value"], d["ms_per_step"], d["step_breakdown"], d.get("serial_value"))
    except Exception as e:
        print(f, "FAILED", e)
PY
tail -5 gpurun_out/r5_bench_overlap.err
