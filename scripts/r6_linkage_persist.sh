cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
DZN_LINKAGE_DEBUG=1 timeout 300 python scripts/bench_linkage.py 5000 20888 35790
} > gpurun_out/r6_linkage_persist.txt 2>&1
tail -40 gpurun_out/r6_linkage_persist.txt
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "linkage" 2>&1 | tail -5
