#!/bin/bash
# A/B on ONE box: the r6 attention on pre-split K / V planes (default) against r5's in-kernel split (DZN_NO_ATT_PLANES=1)
mkdir -p gpurun_out
for mode in planes_pf planes_qb1 split planes_pf split; do
  unset DZN_NO_ATT_PLANES DZN_ATT_QB DZN_ATT_PF
  if [ $mode = planes_pf ]; then export DZN_ATT_PF=1; fi
  if [ $mode = split ]; then export DZN_NO_ATT_PLANES=1; fi
  if [ $mode = planes_qb1 ]; then export DZN_ATT_QB=1; fi
  python bench.py --steps 6 --warmup 2 --no-alt --no-e2e --no-config1 --no-cpu-baseline 2>/dev/null | python -c "
import json, sys
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
ks = {k['kernel']: k for k in d['kernels']}
def f(n):
    k = ks.get(n)
    return f\"{n}: {k['ms_total'] / d['steps']:.2f} ms/step {k.get('tflops')} TFLOP/s\" if k else n + ': -'
print('$mode', 'device_value', d['device_value'], 'ms/step', d['ms_per_step'], '|', f('attention_relpos_f32h'), '|', f('gemm_f32h_128x128'), '|', f('gemm_f32h_128x64'))
"
done 2>&1 | tee gpurun_out/r6_attention_planes_ab.txt
