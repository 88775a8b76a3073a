#!/bin/bash
mkdir -p gpurun_out
set -x
timeout 600 python scripts/diag_host30.py > gpurun_out/r5_diag_host30.txt 2>&1
for m in 0x1 0x3f01; do
  DZN_F16_KEEP2=$m timeout 900 python -m pytest tests/test_seg_gpu.py -q -s -k "meets_the_reduced_bar" > gpurun_out/r5_seg_keep2_$m.log 2>&1
  DZN_F16_KEEP2=$m timeout 900 python -m pytest tests/test_pipeline_gpu.py -q -s -k "der_between" > gpurun_out/r5_der_keep2_$m.log 2>&1
  DZN_F16_KEEP2=$m timeout 600 python bench.py --precision f16 --no-alt --no-e2e --no-config1 --no-cpu-baseline --steps 4 --warmup 1 > gpurun_out/r5_step_f16_keep2_$m.json 2> gpurun_out/r5_step_f16_keep2_$m.err
done
head -30 gpurun_out/r5_diag_host30.txt
grep -n "host stage + RTTM" gpurun_out/r5_diag_host30.txt
for m in 0x1 0x3f01; do grep -h "max_abs_dlogp" gpurun_out/r5_seg_keep2_$m.log | cut -c1-170; grep -h "f16 vs fp32" gpurun_out/r5_der_keep2_$m.log; tail -1 gpurun_out/r5_der_keep2_$m.log; python -c "
import json;d=json.load(open('gpurun_out/r5_step_f16_keep2_$m.json'));print('$m','value',d['value'],'device_value',d.get('device_value'),[(k['kernel'],k['ms_total'],k.get('tflops')) for k in d['kernels'][:6]])"; done
