cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2f
timeout 600 python -m pytest tests/test_dist_gpu.py tests/test_properties_gpu.py -m gpu -q -k "two_ranks or device_postprocess or c_abi" 2>&1 | grep -v "amdgpu.ids" | tail -8
tail -30 gpurun_out/dist_nccl_2.log; tail -30 gpurun_out/dist_gloo_2.log
for CFG in auto 128x128 256x128 128x192 128x192o2; do
  if [ $CFG = auto ]; then unset DZN_GEMM_CFG; else export DZN_GEMM_CFG=$CFG; fi
  timeout 120 python scripts/bench_gemm_h2.py 102144,960,1024 102144,1152,1024 102144,1024,1024 102144,576,1024 102144,1024,256 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r2f/gemm_cfgs.log
done
