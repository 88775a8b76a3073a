# 4 h synthetic recording through DiariZenPipeline on one GPU (profiles/r2_e2e_4h_1gpu.json)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/e2e
timeout 600 python scripts/e2e_timing.py 240 384 > gpurun_out/e2e/e2e_4h.log 2>&1
grep -E "^timings|E2E_JSON" gpurun_out/e2e/e2e_4h.log | cut -c1-500
grep -E "linkage_centroid|cdist_cosine|clustering.py.*__call__|run_host_stage" gpurun_out/e2e/e2e_4h.log | head -8 | cut -c1-160
