# Tile / pipeline-depth probes and the ablation of the dominant kernel (profiles/r2_gemm_cfg_probe.txt,
# profiles/r2_gemm_ablation.txt).  Needs the probe instantiations: build with DZN_TUNING=1 BEFORE sending the tree
#     DZN_TUNING=1 python -c "from diarizen_amd import build; build.build(force=True)"
# (and rebuild without it afterwards: the shipped library carries the production tiles only).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/probe
python scripts/bench_gemm_cfgs.py 128x128w4,256x128s3,256x128w8s3,128x64,128x64s3,256x64s3 149226,1024,1024 149226,1536,1024 149226,1024,512 149226,256,1024 > gpurun_out/probe/cfgs.log 2>&1
python scripts/bench_gemm_cfgs.py 128x128w4,abl1,abl2,abl3,abl4,abl5 149226,1024,1024 > gpurun_out/probe/abl.log 2>&1
python scripts/bench_gemm_cfgs.py 128x64,128x128w4 149226,1024,128 149226,1024,256 149226,1024,384 149226,1024,512 149226,768,256 149226,1024,768 > gpurun_out/probe/smallk.log 2>&1
cut -c1-150 gpurun_out/probe/cfgs.log gpurun_out/probe/abl.log gpurun_out/probe/smallk.log
