# Round-6 measurement set (MI355X, 1 GPU).  Outputs under gpurun_out/final_r6/, the judged ones copied to profiles/r6_*.
# PART=a : PMC passes of the headline command (FETCH_SIZE, WRITE_SIZE, MfmaUtil, clock / MFMA-busy) in f32h, the headline line,
#          rocprofv3 kernel stats of the same command, the driver's command
# PART=b : configs[3] 4 h on one GPU (pipeline timing + the strong-scaling leg at world size 1 with the projected N-GPU efficiency),
#          host-stage profile at 4 h
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final_r6
mkdir -p $O
cd $R
BARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-alt --no-e2e --no-config1 --no-profile"
if [ "${PART:-a}" = "a" ]; then
cd /tmp
for C in FETCH_SIZE WRITE_SIZE MfmaUtil "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; do
T=$(echo $C | cut -d' ' -f1)
timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$T -- python $R/bench.py $BARGS > /dev/null 2> $O/pmc_$T.err
done
cd $R
cc() { find $O/$1 -name '*counter_collection.csv' | head -1; }
python scripts/pmc_summary.py $O/pmc_f32h_30min_b576.json $(cc pmc_FETCH_SIZE) $(cc pmc_WRITE_SIZE) $(cc pmc_MfmaUtil) $(cc pmc_GRBM_GUI_ACTIVE) | tail -3
cp $O/pmc_f32h_30min_b576.json $R/profiles/r6_pmc_f32h_30min_b576.json
find $O -name '*kernel_trace.csv' -delete; find $O -name '*counter_collection.csv' -delete; find $O -name '*agent_info.csv' -delete
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-e2e --no-config1 > $O/bench_under_rocprof.json 2> $O/kt.err
cd $R
cp $(find $O/kt -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
cp $O/kernel_stats.csv $R/profiles/r6_kernel_stats_f32h_30min_b576.csv
find $O -name '*kernel_trace.csv' -delete; find $O -name '*agent_info.csv' -delete
timeout 900 python bench.py > $O/bench_f32h.json 2> $O/bench.err
( time timeout 700 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_style.json 2> $O/bench_driver_style.err
head -12 $O/kernel_stats.csv | cut -c1-160
cut -c1-1500 $O/bench_f32h.json
tail -3 $O/bench_driver_style.err; cut -c1-300 $O/bench_driver_style.json
else
DZN_LINKAGE_DEBUG=1 timeout 600 python scripts/e2e_timing.py 240 576 > $O/e2e_4h.log 2>&1; grep -m1 "^timings" $O/e2e_4h.log; grep -m1 "^E2E_JSON" $O/e2e_4h.log | cut -c10- > $O/e2e_4h_1gpu.json
timeout 600 python bench.py --steps 1 --warmup 1 --no-alt --no-cpu-baseline --no-config1 --strong-minutes 240 > $O/bench_with_strong_4h_leg.json 2> $O/bench_strong.err; python - <<PY
import json
d=json.loads(open("$O/bench_with_strong_4h_leg.json").read().strip().splitlines()[-1]); print(d.get("strong_scaling_e2e"))
PY
timeout 600 python scripts/host_stage_profile.py 240 > $O/host_stage_profile_4h.txt 2>&1; grep "host stage pass" $O/host_stage_profile_4h.txt
true
fi
