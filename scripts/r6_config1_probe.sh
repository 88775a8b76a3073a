cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python scripts/probe_config1.py 32 > gpurun_out/r6_config1_probe.txt 2>&1
grep -v amdgpu.ids gpurun_out/r6_config1_probe.txt
