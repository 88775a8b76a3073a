# conv01: in situ time per 561-window launch: producer/consumer kernel (r6 default) whole / without producers' arithmetic (ABL=1) /
# without the consumers' MFMAs (ABL=2), and the phase-alternating kernel (DZN_CONV01_WS=0)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_seg_gpu.py -m gpu -x -q -k "conv01 or golden" 2>&1 | tail -4
{
for ws in 1 0; do for abl in 0 1 2; do
  echo "== DZN_CONV01_WS=$ws DZN_CONV01_ABL=$abl"
  DZN_CONV01_WS=$ws DZN_CONV01_ABL=$abl timeout 300 python scripts/probe_kernel_class.py 561 conv01 2>&1 | grep -v amdgpu.ids | tail -2
done; done
} > gpurun_out/r6_conv01_probe.txt 2>&1
cat gpurun_out/r6_conv01_probe.txt
