# Round-4 closing measurement on the final tree (deferred layer-weighted sum in): the headline line, the rocprofv3 kernel
# stats of the same command, and the FETCH_SIZE / WRITE_SIZE passes the bench's roofline.traffic is read from
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final_r4b
mkdir -p $O
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
timeout 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$C -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt --no-e2e --no-config1 --no-profile > /dev/null 2> $O/pmc_$C.err
done
cd $R
cc() { find $O/pmc_$1 -name '*counter_collection.csv' | head -1; }
python scripts/pmc_summary.py $O/pmc_f32h_30min_b576_final.json $(cc FETCH_SIZE) $(cc WRITE_SIZE) /nonexistent 2>&1 | tail -5
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-e2e --no-config1 > $O/bench_under_rocprof.json 2> $O/kt.err
cd $R
cp $(find $O/kt -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
find $O -name '*kernel_trace.csv' -delete; find $O -name '*counter_collection.csv' -delete; find $O -name '*agent_info.csv' -delete
( time timeout 400 python bench.py ) > $O/bench_f32h.json 2> $O/bench.err
tail -c 600 $O/bench_f32h.json; head -6 $O/kernel_stats.csv | cut -c1-200
