# Closing set of round 5 (MI355X, 1 GPU), final tree = the set of scripts/final_measure_r5.sh + the host stage off the critical path
# (own stream / arena, pipelined bench steps, DiariZenPipeline.diarize_many).  The kernels of the engine are the ones the PMC passes
# and per-shape tables of the first set were taken on (profiles/r5_pmc_*, r5_kernel_shapes_*), so those are not repeated.
# Outputs under gpurun_out/final_r5b/, copied to profiles/r5_final_* afterwards.
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final_r5b
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/all_gpu_tests.log 2>&1
tail -4 $O/all_gpu_tests.log
timeout 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -5 $O/smoke.log
timeout 900 bash scripts/run_checked.sh > /dev/null 2>&1; cp gpurun_out/r5_checked_build.log $O/checked_build.log; tail -3 $O/checked_build.log
timeout 900 python bench.py > $O/bench_f32h.json 2> $O/bench.err
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_style.json 2> $O/bench_driver_style.err
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-e2e --no-config1 > $O/bench_under_rocprof.json 2> $O/kt.err
cd $R
cp $(find $O/kt -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
find $O -name '*kernel_trace.csv' -delete; find $O -name '*agent_info.csv' -delete
( time DZN_BENCH_ONE_DEVICE=1 timeout 500 python bench.py --gpus 2 --steps 2 --warmup 1 --strong-minutes 30 --batch 192 --no-config1 ) > $O/bench_gpus2_one_device.json 2> $O/bench_gpus2.err
DZN_LINKAGE_DEBUG=1 timeout 600 python scripts/e2e_timing.py 240 576 > $O/e2e_4h.log 2>&1; grep -m1 "^timings" $O/e2e_4h.log; grep -m1 "^E2E_JSON" $O/e2e_4h.log | cut -c10- > $O/e2e_4h_1gpu.json
python - <<PY
import json
for f in ("bench_f32h", "bench_driver_style", "bench_under_rocprof", "bench_gpus2_one_device"):
    try:
        d = json.loads(open("$O/" + f + ".json").read().strip().splitlines()[-1])
        rp = d.get("reduced_precision_mode", {}).get("f16", {}) if isinstance(d.get("reduced_precision_mode"), dict) else {}
        print(f, "value", d["value"], "device", d.get("device_value"), "serial", d.get("serial_value"), "ms", d["ms_per_step"], d["step_breakdown"]["host_exposed_ms"],
              "e2e", (d.get("e2e") or {}).get("audio_seconds_per_s"), (d.get("e2e") or {}).get("corpus_audio_seconds_per_s"),
              "config1", (d.get("config1") or {}).get("audio_seconds_per_s"), "two", (d.get("two_streams") or {}).get("value"),
              "f16", rp.get("value"), (rp.get("roofline") or {}).get("achieved"), "roof", (d.get("roofline") or {}).get("achieved"), (d.get("roofline") or {}).get("frac"),
              "cpu", (d.get("cpu_baseline") or {}).get("value"), "power", (d.get("power") or {}).get("card"), (d.get("power") or {}).get("card_matched_by_pci_bus_id"))
    except Exception as e:
        print(f, "FAILED", e)
PY
head -6 $O/kernel_stats.csv | cut -c1-150
tail -3 $O/bench_driver_style.err
true
