#!/bin/bash
mkdir -p gpurun_out
set -x
for pp in 0 1; do for uf in 4 8; do
  if [ $pp = 1 ]; then export DZN_CONV01_PP=1; else unset DZN_CONV01_PP; fi
  DZN_CONV01_UF=$uf timeout 600 python bench.py --no-alt --no-e2e --no-config1 --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r5_conv01_pp${pp}_uf${uf}.json 2> gpurun_out/r5_conv01_pp${pp}_uf${uf}.err
done; done
unset DZN_CONV01_PP
timeout 900 python -m pytest tests/test_pipeline_gpu.py -q -s -k "der_between or host_stage_30min" > gpurun_out/r5_pipe2.log 2>&1
timeout 900 python -m pytest tests/test_seg_gpu.py -q -s -k "conv01 or conv0_layernorm or meets_the_reduced" > gpurun_out/r5_seg2.log 2>&1
timeout 900 python -m pytest tests/test_emb_gpu.py -q -s -k "reduced" > gpurun_out/r5_emb2.log 2>&1
timeout 1500 bash scripts/run_checked.sh
python - <<'PY'
import json
for pp in (0,1):
    for uf in (4,8):
        try:
            d=json.load(open(f'gpurun_out/r5_conv01_pp{pp}_uf{uf}.json'))
            k=[k for k in d['kernels'] if k['kernel']=='conv01_fused'][0]
            print('PP',pp,'UF',uf,'conv01 ms/launch',round(k['ms_total']/k['launches'],3),'tflops',k.get('tflops'),'device_value',d.get('device_value'),'value',d.get('value'))
        except Exception as e: print(pp,uf,e)
PY
tail -12 gpurun_out/r5_pipe2.log | cut -c1-250; tail -4 gpurun_out/r5_seg2.log | cut -c1-200; tail -3 gpurun_out/r5_emb2.log | cut -c1-200
