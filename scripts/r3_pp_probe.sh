# round 3, GPU call 1: correctness of the ping-pong contraction through the existing per-kernel tests, tile probe, whole step
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3a; mkdir -p $O
for cfg in pp128 pp128s4a2; do
  ( DZN_GEMM_CFG=$cfg timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q 2>&1 | tail -8 | cut -c1-300 ) > $O/ops_$cfg.log 2>&1
  echo "== ops tests $cfg"; cat $O/ops_$cfg.log
done
( DZN_GEMM_CFG=pp128 timeout 600 python -m pytest tests/test_seg_gpu.py -m gpu -x -q 2>&1 | tail -8 | cut -c1-300 ) > $O/seg_pp128.log 2>&1
echo "== seg tests pp128"; cat $O/seg_pp128.log
timeout 600 python scripts/bench_gemm_cfgs.py pp128,pp128s4,pp128s4a2,pp64 149226,1024,1024 149226,1024,256 149226,256,1024 149226,1920,1024 149226,1024,1792 149226,960,1024 149226,1024,512 > $O/cfgs.txt 2>&1
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
