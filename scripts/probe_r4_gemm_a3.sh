# r4 GEMM probe: separate A ring of three LDS stages (128x128a3) vs the production 128x128w4 tile; then the pipeline A/B
set -x
O=gpurun_out/${1:-r4a3}; mkdir -p $O
DZN_GEMM_CFG=128x128a3 timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_f32h_grade_gpu.py -m gpu -q -x 2>&1 | tail -3 > $O/tests_forced_a3.txt
cat $O/tests_forced_a3.txt
timeout 600 python scripts/bench_gemm_cfgs.py 128x128w4,128x128a3 149226,1024,1024 149226,960,1024 149226,1920,1024 149226,4096,1024 149226,1024,4096 149226,384,1024 2>&1 | grep -v "f16 M" > $O/gemm_a3.txt
cat $O/gemm_a3.txt
for v in 0 1; do
if [ $v = 1 ]; then export DZN_GEMM_A3=1; else unset DZN_GEMM_A3; fi
timeout 300 python bench.py --steps 3 --warmup 1 --no-alt --no-e2e --no-config1 --no-cpu-baseline > $O/bench_a3_$v.json 2> $O/bench_a3_$v.err
python - <<PY
import json
d=json.loads(open("$O/bench_a3_$v.json").read().strip().splitlines()[-1])
print("A3=$v", d["value"], d["ms_per_step"], [(k["kernel"], round(k["ms_total"],1), k.get("tflops")) for k in d["kernels"][:4]])
PY
done
