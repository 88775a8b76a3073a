# Round-5 measurement set (MI355X, 1 GPU).  Outputs under gpurun_out/final_r5/, copied to profiles/r5_* afterwards.
# PART=a : PMC passes of the headline command (FETCH_SIZE, WRITE_SIZE, MfmaUtil, clock / MFMA-busy) in f32h, the MfmaUtil / clock /
#          traffic passes of the reduced mode (f16), the headline line (+ other modes, reduced mode with its roofline, e2e, cpu
#          baseline), rocprofv3 kernel stats of the same command, the driver's command, per-shape classes
# PART=b : BASELINE configs[1], configs[3] 4 h on one GPU (pipeline timing + the strong-scaling leg at world size 1), the two-rank
#          rehearsal on one device, decision parity on 256 windows
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final_r5
mkdir -p $O
cd $R
BARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-alt --no-e2e --no-config1 --no-profile"
if [ "${PART:-a}" = "a" ]; then
cd /tmp
for C in FETCH_SIZE WRITE_SIZE MfmaUtil "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; do
T=$(echo $C | cut -d' ' -f1)
timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$T -- python $R/bench.py $BARGS > /dev/null 2> $O/pmc_$T.err
timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc16_$T -- python $R/bench.py --precision f16 $BARGS > /dev/null 2> $O/pmc16_$T.err
done
cd $R
cc() { find $O/$1 -name '*counter_collection.csv' | head -1; }
python scripts/pmc_summary.py $O/pmc_f32h_30min_b576.json $(cc pmc_FETCH_SIZE) $(cc pmc_WRITE_SIZE) $(cc pmc_MfmaUtil) $(cc pmc_GRBM_GUI_ACTIVE) | tail -3
python scripts/pmc_summary.py $O/pmc_f16_30min_b576.json $(cc pmc16_FETCH_SIZE) $(cc pmc16_WRITE_SIZE) $(cc pmc16_MfmaUtil) $(cc pmc16_GRBM_GUI_ACTIVE) | tail -3
cp $O/pmc_f32h_30min_b576.json $R/profiles/r5_pmc_f32h_30min_b576.json
cp $O/pmc_f16_30min_b576.json $R/profiles/r5_pmc_f16_30min_b576.json
find $O -name '*kernel_trace.csv' -delete; find $O -name '*counter_collection.csv' -delete; find $O -name '*agent_info.csv' -delete
timeout 900 python bench.py > $O/bench_f32h.json 2> $O/bench.err
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-e2e --no-config1 > $O/bench_under_rocprof.json 2> $O/kt.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt16 -- python $R/bench.py --precision f16 --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-e2e --no-config1 > $O/bench_f16_under_rocprof.json 2> $O/kt16.err
cd $R
cp $(find $O/kt -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
cp $(find $O/kt16 -name '*kernel_stats.csv' | head -1) $O/kernel_stats_f16.csv
find $O -name '*kernel_trace.csv' -delete; find $O -name '*agent_info.csv' -delete
( time timeout 700 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_style.json 2> $O/bench_driver_style.err
for P in f32h f16; do
DZN_PROFILE_SHAPES=1 timeout 300 python bench.py --precision $P --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-e2e --no-config1 > $O/bench_shapes_$P.json 2> $O/bench_shapes_$P.err
python - <<PY
import json
d=json.loads(open("$O/bench_shapes_$P.json").read().strip().splitlines()[-1])
tot=sum(k["ms_total"] for k in d["kernels"])
with open("$O/kernel_shapes_$P.txt","w") as f:
    f.write(f"{d['value']} {d.get('device_value')} {d['ms_per_step']}\n")
    for k in d["kernels"]:
        f.write(f"{k['kernel']:64s} launches={k['launches']:4d} ms={k['ms_total']:8.2f} share={k['ms_total']/tot:.4f} tflops={k.get('tflops','-')} alg_gbs={k.get('gbs','-')} bound={k.get('bound','-')} frac={k.get('frac_of_bound','-')}\n")
PY
done
head -12 $O/kernel_stats.csv | cut -c1-160
cut -c1-1800 $O/bench_f32h.json
tail -3 $O/bench_driver_style.err; cut -c1-300 $O/bench_driver_style.json
else
timeout 600 python bench.py --model wavlm_base_s80_md --window 5 --batch 32 --stage seg --minutes 30 --steps 3 --warmup 1 --no-alt > $O/bench_base_s80_5s_b32.json 2> $O/bench_base.err
cut -c1-600 $O/bench_base_s80_5s_b32.json
DZN_LINKAGE_DEBUG=1 timeout 600 python scripts/e2e_timing.py 240 576 > $O/e2e_4h.log 2>&1; grep -m1 "^timings" $O/e2e_4h.log; grep -m1 "^E2E_JSON" $O/e2e_4h.log | cut -c10- > $O/e2e_4h_1gpu.json
timeout 600 python bench.py --steps 1 --warmup 1 --no-alt --no-cpu-baseline --no-config1 --strong-minutes 240 > $O/bench_with_strong_4h_leg.json 2> $O/bench_strong.err; python - <<PY
import json
d=json.loads(open("$O/bench_with_strong_4h_leg.json").read().strip().splitlines()[-1]); print(d.get("strong_scaling_e2e"))
PY
( time DZN_BENCH_ONE_DEVICE=1 timeout 500 python bench.py --gpus 2 --steps 2 --warmup 1 --strong-minutes 30 --batch 192 --no-config1 ) > $O/bench_gpus2_one_device.json 2> $O/bench_gpus2.err
cut -c1-400 $O/bench_gpus2_one_device.json
timeout 900 python -m pytest tests/test_decisions_gpu.py -q -s > $O/decisions.log 2>&1; tail -3 $O/decisions.log
true
fi
