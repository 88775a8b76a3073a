# final tree of round 6: the whole GPU suite (serial) + smoke + the driver's bench command
cd $GRAFT_REPO_ROOT
O=gpurun_out/final_r6v; mkdir -p $O
export DZN_DECISION_WINDOWS=32
( time timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 | cut -c1-300 ) > $O/all_gpu_tests.log 2>&1; tail -6 $O/all_gpu_tests.log
timeout 400 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -5 | cut -c1-300 > $O/smoke.log; cat $O/smoke.log
( time timeout 700 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_style.json 2> $O/bench_driver_style.err
tail -3 $O/bench_driver_style.err; cut -c1-260 $O/bench_driver_style.json
