# BASELINE configs[1] (wavlm_base_s80_md, 5 s x 32): kernel classes of one batch, then the same batch with forced contraction tiles
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
timeout 300 python scripts/probe_config1.py 32 2>&1 | grep -v amdgpu.ids
for cfg in 128x64 128x32 128x128 128x128w4; do
  echo "== DZN_GEMM_CFG=$cfg"
  DZN_GEMM_CFG=$cfg timeout 300 python scripts/probe_config1.py 32 2>&1 | grep -v amdgpu.ids | grep "gemm_f32h_1\|sum of kernel\|wall"
done
} > gpurun_out/r6_config1_probe.txt 2>&1
cat gpurun_out/r6_config1_probe.txt
