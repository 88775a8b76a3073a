#!/bin/bash
mkdir -p gpurun_out
set -x
timeout 120 scripts/ubench/mx_probe > gpurun_out/r5_mx_probe2.txt 2>&1
for m in div mul; do
DZN_MX_CVTSCALE=$m timeout 300 python -m pytest tests/test_ops_gpu.py -q -s -k "mx_cross" > gpurun_out/r5_cvts_$m.log 2>&1
done
for m in none div mul; do
if [ $m = none ]; then unset DZN_MX_CVTSCALE; else export DZN_MX_CVTSCALE=$m; fi
timeout 200 python scripts/bench_gemm_mx.py 223839,1024,1024 223839,1024,256 2>&1 | grep "mx auto\|f32h " >> gpurun_out/r5_cvts_micro.txt
echo "---- DZN_MX_CVTSCALE=$m" >> gpurun_out/r5_cvts_micro.txt
done
unset DZN_MX_CVTSCALE
timeout 400 python bench.py --only-config1 > gpurun_out/r5_config1_only.json 2> gpurun_out/r5_config1_only.err
grep "^6\." gpurun_out/r5_mx_probe2.txt; tail -3 gpurun_out/r5_cvts_div.log; tail -3 gpurun_out/r5_cvts_mul.log; cat gpurun_out/r5_cvts_micro.txt; cut -c1-400 gpurun_out/r5_config1_only.json
