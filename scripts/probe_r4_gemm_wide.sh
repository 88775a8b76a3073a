# r4 GEMM probe: 256 x 256 tiles, one wavefront per SIMD (gemm_wide.hip) vs the production 128 x 128 tile; then the pipeline A/B
# (DZN_GEMM_WIDE=1 sends the 128 x 128 class there)
set -x
O=gpurun_out/${1:-r4w}; mkdir -p $O
DZN_GEMM_CFG=wide timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_f32h_grade_gpu.py -m gpu -q -x 2>&1 | tail -5 > $O/tests_forced_wide.txt
cat $O/tests_forced_wide.txt
timeout 600 python scripts/bench_gemm_cfgs.py 128x128w4,wide 149226,1024,1024 149226,960,1024 149226,1920,1024 149226,4096,1024 149226,1024,4096 149226,768,1024 2>&1 | grep -v "f16 M" > $O/gemm_wide.txt
cat $O/gemm_wide.txt
for v in 0 1; do
if [ $v = 1 ]; then export DZN_GEMM_WIDE=1; else unset DZN_GEMM_WIDE; fi
timeout 300 python bench.py --steps 3 --warmup 1 --no-alt --no-e2e --no-config1 --no-cpu-baseline > $O/bench_wide_$v.json 2> $O/bench_wide_$v.err
python - <<PY
import json
d=json.loads(open("$O/bench_wide_$v.json").read().strip().splitlines()[-1])
print("WIDE=$v", d["value"], d["ms_per_step"], [(k["kernel"], round(k["ms_total"],1), k.get("tflops")) for k in d["kernels"][:4]])
PY
done
