#!/bin/bash
# round 5, GPU call 1: pin the MX instruction semantics, first correctness + speed of gemm_mx.hip, reduced mode on the goldens,
# device outputs for the 30-min host-stage fixture, one default bench line
mkdir -p gpurun_out
set -x
timeout 120 scripts/ubench/mx_probe > gpurun_out/r5_mx_probe.txt 2>&1
timeout 600 python -m pytest tests/test_ops_gpu.py -k "mx" -q -s > gpurun_out/r5_mx_ops.log 2>&1
timeout 300 python scripts/bench_gemm_mx.py 223839,1024,1024 223839,1024,256 223839,1920,1024 223839,320,1024 223839,1024,480 > gpurun_out/r5_gemm_mx_bench.txt 2>&1
timeout 900 python -m pytest tests/test_seg_gpu.py -k "f16" -q -s > gpurun_out/r5_seg_f16.log 2>&1
timeout 600 python scripts/dump_device_outputs.py 30 gpurun_out/host30 > gpurun_out/r5_dump.log 2>&1
timeout 900 python bench.py > gpurun_out/r5_bench_a.json 2> gpurun_out/r5_bench_a.err
tail -3 gpurun_out/r5_mx_probe.txt; tail -5 gpurun_out/r5_mx_ops.log; cat gpurun_out/r5_gemm_mx_bench.txt; tail -5 gpurun_out/r5_seg_f16.log; cat gpurun_out/r5_dump.log | tail -2
