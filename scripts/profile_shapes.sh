cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2w
DZN_PROFILE_SHAPES=1 timeout 600 python bench.py --steps 1 --warmup 1 --no-alt --no-e2e --no-cpu-baseline > gpurun_out/r2w/bench_shapes.json 2> gpurun_out/r2w/bench_shapes.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2w/bench_shapes.json'))
print(d['value'], d['ms_per_step'])
for k in d['kernels'][:45]:
    print(f"{k['kernel'][:60]:60s} n={k['launches']:4d} ms={k['ms_total']:8.2f} share={k['share_of_profiled']:.4f} tf={k.get('tflops','')} gbs={k.get('gbs','')}")
PY
