#!/bin/bash
mkdir -p gpurun_out
set -x
timeout 200 python scripts/bench_gemm_mx.py 223839,1024,1024 223839,1920,1024 223839,1024,256 > gpurun_out/r5_mx_256_micro.txt 2>&1
( time timeout 1800 python -m pytest tests -m gpu -q -x ) > gpurun_out/r5_all_gpu_tests.log 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" > gpurun_out/r5_smoke.log 2>&1
grep "mx \|f32h" gpurun_out/r5_mx_256_micro.txt; tail -8 gpurun_out/r5_all_gpu_tests.log; tail -3 gpurun_out/r5_smoke.log
