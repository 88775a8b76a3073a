# Round-2 measurement set (MI355X, 1 GPU).  Outputs under gpurun_out/final/, copied to profiles/r2_* afterwards.
# PART=a : headline bench (+ other modes, e2e, cpu baseline), rocprofv3 kernel stats, the three PMC passes
# PART=b : BASELINE configs[1] (base-s80 5 s x 32, segmentation only), 4 h single-GPU end to end, decision parity 256
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
cd $R
if [ "${PART:-a}" = "a" ]; then
# 1. the three PMC passes FIRST: bench.py reads their per-kernel table (profiles/r2_pmc_*) for roofline.traffic
cd /tmp
for C in FETCH_SIZE WRITE_SIZE MfmaUtil; do
rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$C -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt --no-e2e --no-profile > /dev/null 2> $O/pmc_$C.err
done
cd $R
python scripts/pmc_summary.py $O/pmc_f32h_30min_b384.json $(find $O/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1) $(find $O/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1) $(find $O/pmc_MfmaUtil -name '*counter_collection.csv' | head -1)
cp $O/pmc_f32h_30min_b384.json $R/profiles/r2_pmc_f32h_30min_b384.json
# 2. the headline line (default command), 3. the same command under rocprofv3 --kernel-trace --stats, 4. the driver's command
python bench.py > $O/bench_f32h.json 2> $O/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-e2e > $O/bench_under_rocprof.json 2> $O/kt.err
cd $R
cp $(find $O/kt -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
find $O -name '*kernel_trace.csv' -delete; find $O -name '*counter_collection.csv' -delete; find $O -name '*agent_info.csv' -delete
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_style.json 2> $O/bench_driver_style.err
head -14 $O/kernel_stats.csv
cut -c1-1800 $O/bench_f32h.json
tail -4 $O/bench_driver_style.err; cut -c1-300 $O/bench_driver_style.json
else
python bench.py --model wavlm_base_s80_md --window 5 --batch 32 --stage seg --minutes 30 --steps 3 --warmup 1 --no-alt > $O/bench_base_s80_5s_b32.json 2> $O/bench_base.err
cut -c1-900 $O/bench_base_s80_5s_b32.json
python scripts/e2e_timing.py 240 256 > $O/e2e_4h.log 2>&1; grep -m1 "^timings" $O/e2e_4h.log
DZN_DECISION_WINDOWS=256 python -m pytest tests/test_decisions_gpu.py -m gpu -q 2>&1 | tail -3; cp gpurun_out/decision_parity.json $O/decision_parity_256.json
fi
