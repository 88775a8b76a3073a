cd $GRAFT_REPO_ROOT
PART=a bash scripts/final_measure.sh 2>&1 | grep -v "^+" | tail -6 | cut -c1-400
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/final/bench_driver_style.json 2> gpurun_out/final/bench_driver_style.err; tail -4 gpurun_out/final/bench_driver_style.err; cut -c1-300 gpurun_out/final/bench_driver_style.json
