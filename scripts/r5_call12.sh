#!/bin/bash
# r5 call 12: first-round stagger of the narrow-tile contraction (DZN_GEMM_STAGGER="us[,mode]")
mkdir -p gpurun_out
rm -f /tmp/small_ref.pt gpurun_out/r5_stagger.txt
for c in "" 4 7 10 14 "7,1" "7,2" "14,2"; do
if [ -z "$c" ]; then unset DZN_GEMM_STAGGER; else export DZN_GEMM_STAGGER=$c; fi
echo "---- DZN_GEMM_STAGGER=$c" >> gpurun_out/r5_stagger.txt
timeout 200 python scripts/bench_gemm_small_tiles.py /tmp/small_ref.pt 2>&1 | grep -v amdgpu.ids >> gpurun_out/r5_stagger.txt
done
cat gpurun_out/r5_stagger.txt
