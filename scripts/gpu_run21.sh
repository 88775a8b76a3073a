cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2t
timeout 600 python scripts/bench_gemm_cfgs.py 128x128w4,256x128s3,256x128w8s3,128x128s3,128x128s4,128x64,128x64s3,256x64s3 149226,1024,1024 149226,1536,1024 149226,1024,512 149226,256,1024 > gpurun_out/r2t/cfgs.log 2>&1
cat gpurun_out/r2t/cfgs.log | cut -c1-150
timeout 300 python -m pytest tests/test_ops_gpu.py -q -k "cdist or backends" 2>&1 | tail -3
