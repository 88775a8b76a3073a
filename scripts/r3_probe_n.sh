cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r3n}; mkdir -p $O
timeout 600 python scripts/bench_gemm_cfgs.py 128x64,128x128w4,pq128 149226,1024,128 149226,1024,256 149226,1024,384 149226,1024,512 149226,768,256 149226,512,256 > $O/cfgs.txt 2>&1
grep -v "f16 " $O/cfgs.txt | cut -c1-200
