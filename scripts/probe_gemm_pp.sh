# round 3: correctness of the epilogue + the ping-pong contractions through the per-kernel tests, tile probe, whole step
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r3d}; mkdir -p $O
for cfg in pq128 pq192r3 pq128r3; do
( DZN_GEMM_CFG=$cfg timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q 2>&1 | tail -8 | cut -c1-300 ) > $O/ops_$cfg.log 2>&1
echo "== ops tests $cfg"; cat $O/ops_$cfg.log
done
timeout 600 python scripts/bench_gemm_cfgs.py pp128,pq128,pq128r3,pq192,pq192r3 149226,1024,1024 149226,960,1024 149226,1920,1024 149226,1024,1792 149226,1152,1024 149226,1024,256 149226,384,1024 > $O/cfgs.txt 2>&1
echo "== cfg probe"; grep -v "f16 " $O/cfgs.txt | cut -c1-200
for cfg in ${2:-pq128r3}; do
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
