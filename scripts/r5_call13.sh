#!/bin/bash
# r5 call 13: attention — skip the query-less wavefronts of the last query tile (B) and the masked key blocks of the last key tile (A)
mkdir -p gpurun_out
O=gpurun_out/r5_attention_skip.txt
rm -f $O
ATTB=$(pwd)/diarizen_amd/lib/libdzn_hip_attb.so
timeout 600 python -m pytest tests/test_ops_gpu.py -q -k "attention" 2>&1 | tail -2 | tee -a $O
DZN_HIP_LIB=$ATTB timeout 600 python -m pytest tests/test_ops_gpu.py -q -k "attention" 2>&1 | tail -2 | tee -a $O
for i in 1 2; do
echo "---- kernel microbench (scripts/bench_attention.py: B 256, L 399, 12 of 16 heads), pass $i" >> $O
echo "[r2-r4 kernel: DZN_ATT_NOSKIP=1]" >> $O; DZN_ATT_NOSKIP=1 timeout 200 python scripts/bench_attention.py 2>&1 | grep prec >> $O
echo "[wavefront skip only (probe build)]" >> $O; DZN_HIP_LIB=$ATTB timeout 200 python scripts/bench_attention.py 2>&1 | grep prec >> $O
echo "[wavefront skip + masked key blocks]" >> $O; timeout 200 python scripts/bench_attention.py 2>&1 | grep prec >> $O
done
B="--steps 4 --warmup 2 --no-cpu-baseline --no-alt --no-e2e --no-config1 --no-power"
for v in noskip attb both noskip attb both; do
case $v in
noskip) export DZN_ATT_NOSKIP=1; unset DZN_HIP_LIB;;
attb) unset DZN_ATT_NOSKIP; export DZN_HIP_LIB=$ATTB;;
both) unset DZN_ATT_NOSKIP; unset DZN_HIP_LIB;;
esac
timeout 300 python bench.py $B > gpurun_out/att_$v.json 2> gpurun_out/att_$v.err
python - >> $O <<PY
import json
d=json.loads(open("gpurun_out/att_$v.json").read().strip().splitlines()[-1])
print("$v", "value", d["value"], "device", d["device_value"], "ms", d["ms_per_step"], [(k["kernel"], round(k["ms_total"],1), k.get("tflops")) for k in d["kernels"] if "attention" in k["kernel"]])
PY
done
unset DZN_ATT_NOSKIP DZN_HIP_LIB
cat $O
