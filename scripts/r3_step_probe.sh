# round 3: whole-step comparisons of contraction heuristics + the new parity tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r3g}; mkdir -p $O
( timeout 900 python -m pytest tests/test_f32h_grade_gpu.py tests/test_emb_gpu.py tests/test_seg_gpu.py -m gpu -x -q 2>&1 | tail -12 | cut -c1-400 ) > $O/tests_new.log 2>&1
echo "== new tests"; cat $O/tests_new.log
for tag in auto pq; do
  if [ $tag = pq ]; then export DZN_GEMM_PQ=1; else unset DZN_GEMM_PQ; fi
  DZN_PROFILE_SHAPES=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-alt --no-e2e --no-cpu-baseline > $O/bench_shapes_$tag.json 2> $O/bench_shapes_$tag.err
  echo "== bench $tag"; python - <<PY
import json
try:
    d=json.loads(open("$O/bench_shapes_$tag.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"])
except Exception as e:
    print("bench failed", e); print(open("$O/bench_shapes_$tag.err").read()[-1500:])
PY
done
unset DZN_GEMM_PQ
