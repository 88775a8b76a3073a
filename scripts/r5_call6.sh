#!/bin/bash
mkdir -p gpurun_out
set -x
timeout 900 python -m pytest tests/test_ops_gpu.py -q -s -k "resblock" > gpurun_out/r5_resblock_np1.log 2>&1
timeout 900 python -m pytest tests/test_emb_gpu.py -q -s > gpurun_out/r5_emb3.log 2>&1
timeout 900 python -m pytest tests/test_pipeline_gpu.py -q -s -k "der_between" > gpurun_out/r5_der3.log 2>&1
for k in 0x3f01 0x7f01; do
DZN_F16_KEEP2=$k timeout 600 python bench.py --precision f16 --no-alt --no-e2e --no-config1 --no-cpu-baseline --steps 4 --warmup 1 > gpurun_out/r5_step_f16_np1_$k.json 2> gpurun_out/r5_step_f16_np1_$k.err
done
tail -8 gpurun_out/r5_resblock_np1.log | cut -c1-200; tail -4 gpurun_out/r5_emb3.log | cut -c1-200; grep -h "f16 vs fp32\|passed\|failed" gpurun_out/r5_der3.log | cut -c1-200
grep -h "embeddings vs the reference" gpurun_out/r5_emb3.log
for k in 0x3f01 0x7f01; do python -c "
import json;d=json.load(open('gpurun_out/r5_step_f16_np1_$k.json'));print('$k','value',d['value'],'device_value',d.get('device_value'),[(k['kernel'],k['ms_total'],k.get('tflops')) for k in d['kernels'][:9]])"; done
