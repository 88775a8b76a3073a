#!/bin/bash
# r5 call 14: attention — request the next K / V tile in the middle of the current tile's compute (after the scores), same registers
mkdir -p gpurun_out
O=gpurun_out/r5_attention_midfetch.txt
rm -f $O
ATTM=$(pwd)/diarizen_amd/lib/libdzn_hip_attm.so
timeout 600 python -m pytest tests/test_ops_gpu.py -q -k "attention" 2>&1 | tail -1 | tee -a $O
DZN_HIP_LIB=$ATTM timeout 600 python -m pytest tests/test_ops_gpu.py -q -k "attention" 2>&1 | tail -1 | tee -a $O
for i in 1 2; do
echo "---- kernel microbench (scripts/bench_attention.py: B 256, L 399, 12 of 16 heads), pass $i" >> $O
echo "[kernel of the tree: fetch at the top of the tile]" >> $O; timeout 200 python scripts/bench_attention.py 2>&1 | grep prec >> $O
echo "[probe build: next tile requested after the scores]" >> $O; DZN_HIP_LIB=$ATTM timeout 200 python scripts/bench_attention.py 2>&1 | grep prec >> $O
done
B="--steps 4 --warmup 2 --no-cpu-baseline --no-alt --no-e2e --no-config1 --no-power"
for v in tree mid tree mid; do
case $v in
tree) unset DZN_HIP_LIB;;
mid) export DZN_HIP_LIB=$ATTM;;
esac
timeout 300 python bench.py $B > gpurun_out/attm_$v.json 2> gpurun_out/attm_$v.err
python - >> $O <<PY
import json
d=json.loads(open("gpurun_out/attm_$v.json").read().strip().splitlines()[-1])
print("$v", "value", d["value"], "device", d["device_value"], "ms", d["ms_per_step"], [(k["kernel"], round(k["ms_total"]/4,2), k.get("tflops")) for k in d["kernels"] if "attention" in k["kernel"]])
PY
done
unset DZN_HIP_LIB
cat $O
