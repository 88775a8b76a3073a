cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r3p}; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "linkage" > $O/tests.txt 2>&1; tail -5 $O/tests.txt
timeout 300 python scripts/bench_linkage.py 5000 20000 35790 > $O/linkage.txt 2>&1; cat $O/linkage.txt
