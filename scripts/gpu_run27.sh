cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2x
timeout 600 python scripts/bench_gemm_cfgs.py 128x64,128x128w4 149226,1024,128 149226,1024,256 149226,1024,384 149226,1024,512 149226,768,256 149226,1024,768 > gpurun_out/r2x/smallk.log 2>&1
grep f32h gpurun_out/r2x/smallk.log | cut -c1-150
