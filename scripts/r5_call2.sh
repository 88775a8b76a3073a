#!/bin/bash
# round 5, GPU call 2: residual-prefetch A/B (micro + step), reduced-mode validation set, host-stage golden on the device backends,
# the new bench line
mkdir -p gpurun_out
set -x
for r in 0 1 2; do
  DZN_GEMM_RPF=$r timeout 200 python scripts/bench_gemm_mx.py 223839,1024,256 223839,1024,480 223839,1024,128 >> gpurun_out/r5_rpf_micro.txt 2>&1
done
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "gemm or mx" > gpurun_out/r5_ops_gemm.log 2>&1
timeout 900 python -m pytest tests/test_seg_gpu.py -q -s -k "f16 or turn_taking" > gpurun_out/r5_seg.log 2>&1
timeout 900 python -m pytest tests/test_pipeline_gpu.py -q -s -k "der_between or host_stage_30min" > gpurun_out/r5_pipe.log 2>&1
timeout 600 python -m pytest tests/test_emb_gpu.py -q -s -k "reduced" > gpurun_out/r5_emb_reduced.log 2>&1
for r in 0 1; do
  DZN_GEMM_RPF=$r timeout 600 python bench.py --no-alt --no-e2e --no-config1 --no-cpu-baseline --steps 4 --warmup 1 > gpurun_out/r5_step_rpf$r.json 2> gpurun_out/r5_step_rpf$r.err
done
timeout 1200 python bench.py > gpurun_out/r5_bench_b.json 2> gpurun_out/r5_bench_b.err
cat gpurun_out/r5_rpf_micro.txt; tail -3 gpurun_out/r5_ops_gemm.log; tail -12 gpurun_out/r5_seg.log; tail -12 gpurun_out/r5_pipe.log; tail -4 gpurun_out/r5_emb_reduced.log
python - <<'PY'
import json
for r in (0,1):
    try:
        d=json.load(open(f'gpurun_out/r5_step_rpf{r}.json'))
        print('RPF',r,'value',d['value'],'device_value',d.get('device_value'),'ms',d['ms_per_step'],[ (k['kernel'],k['ms_total'],k.get('tflops')) for k in d['kernels'][:4]])
    except Exception as e: print('rpf',r,e)
try:
    d=json.load(open('gpurun_out/r5_bench_b.json'))
    print({k:d.get(k) for k in ('value','device_value','e2e_value','fp32_mfma_value','step_breakdown')})
    print(d['reduced_precision_mode']['f16'].get('value'), d['reduced_precision_mode']['f16'].get('roofline'))
    print(d.get('cpu_baseline'))
except Exception as e: print('bench_b',e)
PY
