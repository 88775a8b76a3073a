cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2s
export DZN_DECISION_WINDOWS=32
timeout 900 python -m pytest tests/test_ops_gpu.py -q -k "f16 or cdist or backends or linkage" 2>&1 | tail -8 > gpurun_out/r2s/ops.log; cat gpurun_out/r2s/ops.log
timeout 900 python -m pytest tests/test_seg_gpu.py tests/test_emb_gpu.py -q -s -k "f16 or reduced" 2>&1 | grep -v "^$" | tail -12 > gpurun_out/r2s/seg.log; cat gpurun_out/r2s/seg.log
timeout 900 python -m pytest tests/test_pipeline_gpu.py tests/test_decisions_gpu.py -q -s 2>&1 | grep -v "^$" | tail -12 | cut -c1-1500 > gpurun_out/r2s/pipe.log; cat gpurun_out/r2s/pipe.log
timeout 600 python bench.py --precision f16 --steps 2 --warmup 1 --no-e2e --no-cpu-baseline > gpurun_out/r2s/bench_f16.json 2> gpurun_out/r2s/bench_f16.err; tail -3 gpurun_out/r2s/bench_f16.err; cut -c1-400 gpurun_out/r2s/bench_f16.json
timeout 600 python scripts/e2e_timing.py 240 384 > gpurun_out/r2s/e2e_4h.log 2>&1; grep -E "timings|E2E_JSON" gpurun_out/r2s/e2e_4h.log | cut -c1-600; tail -45 gpurun_out/r2s/e2e_4h.log | cut -c1-160
