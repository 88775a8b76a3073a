# round 3: correctness of the batched epilogue + the ping-pong contraction through the per-kernel tests, tile probe, whole step
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r3b}; mkdir -p $O
( timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_seg_gpu.py tests/test_properties_gpu.py -m gpu -x -q 2>&1 | tail -8 | cut -c1-300 ) > $O/tests_auto.log 2>&1
echo "== tests auto"; cat $O/tests_auto.log
( DZN_GEMM_CFG=pp128 timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q 2>&1 | tail -8 | cut -c1-300 ) > $O/ops_pp128.log 2>&1
echo "== ops tests pp128"; cat $O/ops_pp128.log
timeout 600 python scripts/bench_gemm_cfgs.py pp128,pp128s4a2 149226,1024,1024 149226,1024,256 149226,256,1024 149226,1920,1024 149226,1024,1792 149226,1024,512 149226,1024,128 > $O/cfgs.txt 2>&1
echo "== cfg probe"; grep -v "f16 " $O/cfgs.txt | cut -c1-200
for cfg in auto pp128; do
  DZN_GEMM_CFG=$cfg DZN_PROFILE_SHAPES=1 timeout 600 python bench.py --steps 1 --warmup 1 --no-alt --no-e2e --no-cpu-baseline > $O/bench_shapes_$cfg.json 2> $O/bench_shapes_$cfg.err
  echo "== bench $cfg"; python - <<PY
import json
try:
    d=json.loads(open("$O/bench_shapes_$cfg.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"])
except Exception as e:
    print("bench failed", e); print(open("$O/bench_shapes_$cfg.err").read()[-1500:])
PY
done
