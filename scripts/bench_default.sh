# the default bench line on the final tree (profiles/r2_bench_f32h_30min_b384.json)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final2
( time python bench.py > gpurun_out/final2/bench_f32h.json 2> gpurun_out/final2/bench.err ) 2>&1 | tail -3
cut -c1-200 gpurun_out/final2/bench_f32h.json
