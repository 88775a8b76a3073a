cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2k
for B in 96 128 192 256 384; do
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-e2e --batch $B > gpurun_out/r2k/bench_b$B.json 2> gpurun_out/r2k/bench_b$B.err
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r2k/bench_b$B.json"))
    print("batch $B:", d["value"], d["ms_per_step"])
except Exception as e:
    print("batch $B failed", e)
PY
done
