# r4: deferred layer-weighted sum (one pass over per-layer buffers) vs the read-modify-write in every FFN-output epilogue
set -x
O=gpurun_out/${1:-r4ws}; mkdir -p $O
timeout 900 python -m pytest tests/test_seg_gpu.py tests/test_properties_gpu.py -m gpu -q -x 2>&1 | tail -4 > $O/tests.txt; cat $O/tests.txt
timeout 300 python -m pytest tests/test_emb_gpu.py -m gpu -q -x -k "graph" 2>&1 | tail -2
for v in 1 0 1 0; do
if [ $v = 1 ]; then export DZN_NO_WS_DEFER=1; else unset DZN_NO_WS_DEFER; fi
timeout 300 python bench.py --steps 3 --warmup 1 --no-alt --no-e2e --no-config1 --no-cpu-baseline > $O/bench_nodefer_$v.json 2> $O/bench_nodefer_$v.err
python - <<PY
import json
d=json.loads(open("$O/bench_nodefer_$v.json").read().strip().splitlines()[-1])
ks={k["kernel"]:k for k in d["kernels"]}
print("NO_DEFER=$v", d["value"], d["ms_per_step"], [(n, round(ks[n]["ms_total"],1), ks[n].get("tflops")) for n in ("gemm_f32h_128x128","gemm_f32h_128x64","ws_sum","ws_accum") if n in ks])
PY
done
