cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2u
timeout 600 python scripts/bench_gemm_cfgs.py 128x128w4,abl1,abl2,abl3,abl4,abl5 149226,1024,1024 > gpurun_out/r2u/abl.log 2>&1
cat gpurun_out/r2u/abl.log | cut -c1-150
timeout 300 python -m pytest tests/test_ops_gpu.py -q -k "cdist or backends" 2>&1 | tail -12 | cut -c1-300
