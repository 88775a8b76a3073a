# Last verification of the round-5 tree (after the attention change): GPU suite, smoke, checked build, the driver's bench command.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final_r5c
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/all_gpu_tests.log 2>&1
tail -4 $O/all_gpu_tests.log
timeout 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -5 $O/smoke.log
timeout 900 bash scripts/run_checked.sh > /dev/null 2>&1; cp gpurun_out/r5_checked_build.log $O/checked_build.log; tail -3 $O/checked_build.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_style.json 2> $O/bench_driver_style.err
python - <<PY
import json
d = json.loads(open("$O/bench_driver_style.json").read().strip().splitlines()[-1])
print("value", d["value"], "device", d["device_value"], "serial", d["serial_value"], "ms", d["ms_per_step"], d["step_breakdown"]["host_exposed_ms"], "e2e", d["e2e"]["audio_seconds_per_s"], d["e2e"]["corpus_audio_seconds_per_s"],
      "config1", d["config1"]["by_streams"], "two", d["two_streams"]["value"], "f16", d["reduced_precision_mode"]["f16"]["value"], "roof", d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"],
      "f32s/f32", d["other_fp32_modes"]["f32s"]["value"], d["fp32_mfma_mode"]["value"], "cpu", d["cpu_baseline"]["value"])
PY
tail -3 $O/bench_driver_style.err
