#!/bin/bash
# what the planes attention kernel waits for: one ingredient removed at a time (DZN_ATT_ABL, WRONG results on purpose)
mkdir -p gpurun_out
for abl in 0 1 2 4 3 6 7; do
  for qb in 1 2; do
  DZN_ATT_ABL=$abl DZN_ATT_QB=$qb python bench.py --steps 4 --warmup 1 --no-alt --no-e2e --no-config1 --no-cpu-baseline --no-power 2>/dev/null | python -c "
import json, sys
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
ks = {k['kernel']: k for k in d['kernels']}
k = ks['attention_relpos_f32h']
print('abl $abl qb $qb: attention %.2f ms/step (%s TFLOP/s nominal)' % (k['ms_total'] / d['steps'], k.get('tflops')))
"
  done
done 2>&1 | tee gpurun_out/r6_attention_ablation.txt
