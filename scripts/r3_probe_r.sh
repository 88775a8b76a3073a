cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r3r}; mkdir -p $O
DZN_LINKAGE_DEBUG=1 timeout 600 python scripts/e2e_timing.py 240 384 > $O/e2e_4h.log 2>&1
grep -m1 "^timings" $O/e2e_4h.log; grep "^linkage" $O/e2e_4h.log; grep -A22 "Ordered by: internal time" $O/e2e_4h.log | cut -c1-150
