# round 3: what bounds the contraction — variants + PMC counters on two kernel forms
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r3e}; mkdir -p $O
cd $R
( timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q 2>&1 | tail -5 | cut -c1-300 ) > $O/ops_auto.log 2>&1; cat $O/ops_auto.log
( DZN_GEMM_CFG=pq192r3 timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q 2>&1 | tail -5 | cut -c1-300 ) > $O/ops_pq192r3.log 2>&1; cat $O/ops_pq192r3.log
timeout 600 python scripts/probe_gemm_bound.py auto,pq128r3 > $O/bound.txt 2>&1; grep cfg= $O/bound.txt
cd /tmp
for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
  T=$(echo $P | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/pmc_$T -- python $R/scripts/probe_gemm_bound.py auto,pq128r3 quick > /dev/null 2> $O/pmc_$T.err
  F=$(find $O/pmc_$T -name '*counter_collection.csv' | head -1)
  python - <<PY
import csv, collections
vals = collections.defaultdict(list)
try:
    for r in csv.DictReader(open("$F")):
        n = r["Kernel_Name"]
        if "gemm" not in n: continue
        n = n.split("<")[0].split("::")[-1] + "<" + n.split("<")[1][:24]
        vals[(n, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (n, c), v in sorted(vals.items()):
        print(f"{n:50s} {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
except Exception as e:
    print("pmc parse failed", "$T", e)
PY
  find $O/pmc_$T -name '*.csv' -size +2M -delete
done 2>&1 | tee $O/pmc.txt
