# the whole GPU suite (serial, as the driver runs it: xdist only makes it slower and flaky) + smoke on the current tree
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
export DZN_DECISION_WINDOWS=32
( time timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 | cut -c1-400 ) > $O/${1:-r6_all_gpu_tests_b}.log 2>&1; cat $O/${1:-r6_all_gpu_tests_b}.log
timeout 400 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | cut -c1-300
