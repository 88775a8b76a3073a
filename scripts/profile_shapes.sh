# per-shape kernel table of the headline step (DZN_PROFILE_SHAPES=1: the in-situ profiler keys GEMM classes by M/N/K/z)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-shapes}; mkdir -p $O
DZN_PROFILE_SHAPES=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-alt --no-e2e --no-cpu-baseline > $O/bench_shapes.json 2> $O/bench_shapes.err
python - > $O/kernel_shapes.txt <<PY
import json
d=json.load(open('$O/bench_shapes.json'))
print(d['value'], d['ms_per_step'])
for k in d['kernels']:
    print(f"{k['kernel'][:64]:64s} launches={k['launches']:4d} ms={k['ms_total']:8.2f} share={k['share_of_profiled']:.4f} tflops={k.get('tflops','-')} alg_gbs={k.get('gbs','-')}")
PY
head -60 $O/kernel_shapes.txt
