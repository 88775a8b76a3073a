"""Summarise a rocprofv3 --pmc MfmaUtil pass per kernel: launch-count, mean and duration-free median of the
derived counter (percent of SIMD-cycles with the matrix pipe busy).   usage: pmc_mfma.py counters.csv out.json"""
import csv, collections, json, statistics, sys
path, outp = sys.argv[1:3]
vals = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    if r["Counter_Name"] != "MfmaUtil":
        continue
    n = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
    vals[n].append(float(r["Counter_Value"]))
out = {n: {"launches": len(v), "MfmaUtil_mean_pct": round(statistics.fmean(v), 2),
           "MfmaUtil_median_pct": round(statistics.median(v), 2)} for n, v in vals.items()}
json.dump(out, open(outp, "w"), indent=1, sort_keys=True)
for n, e in sorted(out.items(), key=lambda kv: -kv[1]["MfmaUtil_mean_pct"])[:12]:
    print(n, e)
