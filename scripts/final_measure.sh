set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/final
cd $R
python bench.py > gpurun_out/final/bench_f32s.json 2> gpurun_out/final/bench.err
python bench.py --precision bf16 --no-cpu-baseline --no-alt > gpurun_out/final/bench_bf16.json 2>> gpurun_out/final/bench.err; python bench.py --precision f32 --no-cpu-baseline --no-alt > gpurun_out/final/bench_f32.json 2>> gpurun_out/final/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/kt -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt > $R/gpurun_out/final/bench_under_rocprof.json 2> $R/gpurun_out/final/kt.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/final/pmc_fetch -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt --no-profile > /dev/null 2> $R/gpurun_out/final/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/final/pmc_write -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt --no-profile > /dev/null 2> $R/gpurun_out/final/pmc_write.err
rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $R/gpurun_out/final/pmc_mfma -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt --no-profile > /dev/null 2> $R/gpurun_out/final/pmc_mfma.err
cd $R
U=$(find gpurun_out/final/pmc_mfma -name '*counter_collection.csv' | head -1)
python scripts/pmc_mfma.py $U gpurun_out/final/pmc_mfma_util_f32s_b256.json
F=$(find gpurun_out/final/pmc_fetch -name '*counter_collection.csv' | head -1)
W=$(find gpurun_out/final/pmc_write -name '*counter_collection.csv' | head -1)
python scripts/pmc_traffic.py $F $W gpurun_out/final/pmc_traffic_f32s_b256.json
S=$(find gpurun_out/final/kt -name '*kernel_stats.csv' | head -1)
cp $S gpurun_out/final/kernel_stats.csv
# big raw traces do not need to come back
find gpurun_out/final -name '*kernel_trace.csv' -delete
find gpurun_out/final -name '*counter_collection.csv' -delete
find gpurun_out/final -name '*agent_info.csv' -delete
head -12 gpurun_out/final/kernel_stats.csv
cat gpurun_out/final/bench_f32s.json | cut -c1-1500
cat gpurun_out/final/bench_bf16.json | cut -c1-600
