cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time python bench.py --steps 10 --warmup 3 > gpurun_out/r6_bench_mid.json 2> gpurun_out/r6_bench_mid.err ) 2>&1 | tail -3
cut -c1-400 gpurun_out/r6_bench_mid.json
