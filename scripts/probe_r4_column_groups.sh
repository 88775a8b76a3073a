# r4: column groups of the split contraction (tile order that keeps an XCD's share of the weight planes inside its L2)
set -x
O=gpurun_out/${1:-r4cg}; mkdir -p $O
DZN_GEMM_NGROUPS=4 timeout 120 python -m pytest tests/test_ops_gpu.py -m gpu -q -x 2>&1 | tail -2
DZN_GEMM_NGROUPS=3 timeout 120 python -m pytest tests/test_seg_gpu.py -m gpu -q -x -k "golden_turn_taking and f32h" 2>&1 | tail -2
for v in 1 0 1 0; do
if [ $v = 1 ]; then export DZN_GEMM_NGROUPS=1; else unset DZN_GEMM_NGROUPS; fi
timeout 100 python bench.py --steps 3 --warmup 1 --no-alt --no-e2e --no-config1 --no-cpu-baseline --no-power > $O/bench_ng_$v.json 2> $O/bench_ng_$v.err
python - <<PY
import json
d=json.loads(open("$O/bench_ng_$v.json").read().strip().splitlines()[-1])
ks={k["kernel"]:k for k in d["kernels"]}
print("NGROUPS_forced_1=$v", d["value"], d["ms_per_step"], [(n, round(ks[n]["ms_total"],1), ks[n].get("tflops")) for n in ("gemm_f32h_128x128","gemm_f32h_128x64") if n in ks])
PY
done
