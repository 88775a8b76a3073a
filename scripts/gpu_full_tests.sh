cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2v
export DZN_DECISION_WINDOWS=32
( time timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 | cut -c1-400 ) > gpurun_out/r2v/all_gpu_tests.log 2>&1; cat gpurun_out/r2v/all_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6 | cut -c1-300
