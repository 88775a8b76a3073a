# r4 attention probe: query blocks per workgroup (DZN_ATT_QW = 4 (r3) / 8 / 16): per-kernel tests, then the pipeline A/B
O=gpurun_out/${1:-r4f}; mkdir -p $O
for q in 4 8 16; do
  DZN_ATT_QW=$q timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "attention" 2>&1 | tail -1
  DZN_ATT_QW=$q timeout 300 python bench.py --steps 3 --warmup 1 --no-alt --no-e2e --no-config1 --no-cpu-baseline > $O/bench_qw$q.json 2> $O/bench_qw$q.err
  python - <<PY
import json
d=json.loads(open("$O/bench_qw$q.json").read().strip().splitlines()[-1])
print("QW=$q", d["value"], d["ms_per_step"], [(k["kernel"], round(k["ms_total"],1), k.get("tflops")) for k in d["kernels"] if "attention" in k["kernel"]])
PY
done
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/fetch_calib -- $GRAFT_REPO_ROOT/scripts/ubench/fetch_calib > $GRAFT_REPO_ROOT/$O/fetch_calib.txt 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections
f=glob.glob("$O/fetch_calib/**/*counter_collection.csv", recursive=True)
acc=collections.defaultdict(list)
for row in csv.DictReader(open(f[0])):
    if row.get("Counter_Name")=="FETCH_SIZE": acc[row["Kernel_Name"][:60]].append(float(row["Counter_Value"]))
for k,v in acc.items(): print("FETCH_SIZE", k, [round(x) for x in v])
PY
tail -2 $O/fetch_calib.txt
