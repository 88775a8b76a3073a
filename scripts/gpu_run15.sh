cd $GRAFT_REPO_ROOT
for CFG in 128x128 128x128w4 256x128w8; do
  export DZN_GEMM_CFG=$CFG
  timeout 120 python scripts/bench_gemm_h2.py 102144,960,1024 102144,1152,1024 102144,1024,1024 102144,576,1024 102144,1024,512 2>&1 | grep -v amdgpu.ids | grep f32h
done
