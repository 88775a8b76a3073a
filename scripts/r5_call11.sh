#!/bin/bash
# r5 call 11: (a) smaller / more workgroups per CU for the short-K contraction class; (b) the driver's command on the final tree
mkdir -p gpurun_out
rm -f /tmp/small_ref.pt
for c in "" 128x64 64x64w2 128x32 128x32o4; do
if [ -z "$c" ]; then unset DZN_GEMM_CFG; else export DZN_GEMM_CFG=$c; fi
timeout 200 python scripts/bench_gemm_small_tiles.py /tmp/small_ref.pt >> gpurun_out/r5_small_tiles.txt 2>&1
done
unset DZN_GEMM_CFG
cat gpurun_out/r5_small_tiles.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r5_final_bench_driver_style.json 2> gpurun_out/r5_final_bench_driver_style.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5_final_bench_driver_style.json").read().strip().splitlines()[-1])
print("value", d["value"], "device", d["device_value"], "serial", d["serial_value"], "e2e", d["e2e"]["audio_seconds_per_s"], d["e2e"]["corpus_audio_seconds_per_s"],
      "config1", d["config1"]["by_streams"], "two", d["two_streams"]["value"], "f16", d["reduced_precision_mode"]["f16"]["value"], "roof", d["roofline"]["achieved"], d["roofline"]["frac"])
PY
tail -3 gpurun_out/r5_final_bench_driver_style.err
