# round 3: fused frontend, two tiles per workgroup in anti-phase
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r3k}; mkdir -p $O
( timeout 900 python -m pytest tests/test_seg_gpu.py tests/test_decisions_gpu.py tests/test_properties_gpu.py -m gpu -x -q 2>&1 | tail -12 | cut -c1-400 ) > $O/tests.log 2>&1
echo "== tests"; cat $O/tests.log
echo "== conv01 two-tile"; timeout 300 python scripts/probe_kernel_class.py 374 conv01 2>&1 | tail -3
echo "== conv01 r2 form"; DZN_CONV01_NO_PP=1 timeout 300 python scripts/probe_kernel_class.py 374 conv01 2>&1 | tail -3
for abl in 1 2; do echo "== two-tile abl=$abl"; DZN_CONV01_ABL=$abl timeout 300 python scripts/probe_kernel_class.py 374 conv01 2>&1 | tail -3; done
timeout 600 python bench.py --steps 3 --warmup 1 --no-alt --no-e2e --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<PY
import json
try:
    d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
    print("default", d["value"], d["ms_per_step"])
    for k in d["kernels"][:8]: print("   ", {a:b for a,b in k.items() if a!='alg_bytes_per_launch'})
except Exception as e:
    print("bench failed", e); print(open("$O/bench.err").read()[-2500:])
PY
