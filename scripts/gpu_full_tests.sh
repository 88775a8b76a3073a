# round 3: the whole GPU suite + smoke + the 4 h single-GPU pipeline timing
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r3full}; mkdir -p $O
export DZN_DECISION_WINDOWS=32
( time timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -15 | cut -c1-400 ) > $O/all_gpu_tests.log 2>&1; cat $O/all_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | cut -c1-300
DZN_LINKAGE_DEBUG=1 timeout 600 python scripts/e2e_timing.py 240 384 > $O/e2e_4h.log 2>&1; grep -m1 "^timings" $O/e2e_4h.log; grep -m1 "^E2E_JSON" $O/e2e_4h.log | cut -c10- > $O/e2e_4h_1gpu.json
