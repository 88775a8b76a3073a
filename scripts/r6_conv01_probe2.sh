cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_seg_gpu.py -m gpu -x -q -k "conv01 or golden" 2>&1 | tail -3
{
for abl in 0 1 5 2; do
  echo "== DZN_CONV01_ABL=$abl (1 = no producer arithmetic; +4 no W loads; +8 no A reads; 2 = no consumer MFMA)"
  DZN_CONV01_ABL=$abl timeout 300 python scripts/probe_kernel_class.py 561 conv01 2>&1 | grep -v amdgpu.ids | tail -2 | head -1
done
} > gpurun_out/r6_conv01_probe2.txt 2>&1
cat gpurun_out/r6_conv01_probe2.txt
