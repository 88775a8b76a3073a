"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel name substring."""
import csv, collections, sys
path, pat = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(path)))
agg = collections.defaultdict(float); disp = set()
for r in rows:
    if pat in r["Kernel_Name"]:
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
print("dispatches", len(disp))
for k, v in sorted(agg.items()):
    print(f"{k:32s} {v:.4g}  per-dispatch {v/len(disp):.4g}")
