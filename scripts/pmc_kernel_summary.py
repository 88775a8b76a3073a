"""Per-kernel means of the counters of rocprofv3 --pmc passes:  pmc_kernel_summary.py <dir with p*/ csv trees> <kernel-name substring>
Prints, for every kernel whose name contains the substring, launches and the mean of each counter (summed over dimensions per
dispatch), plus a few ratios when their counters are present."""
import collections
import csv
import glob
import sys
root, want = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))   # kernel -> counter -> dispatch -> sum
for f in glob.glob(f"{root}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if want not in n:
            continue
        n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
        acc[n][r["Counter_Name"]][(f, r["Dispatch_Id"])] += float(r["Counter_Value"])
for n, cs in acc.items():
    m = {c: sum(d.values()) / len(d) for c, d in cs.items()}
    print(f"== {n}: {max(len(d) for d in cs.values())} dispatches")
    for c in sorted(m):
        print(f"   {c:28s} {m[c]:16.0f}")
    g = m.get
    if g("SQ_BUSY_CYCLES") and g("SQ_VALU_MFMA_BUSY_CYCLES"):
        print(f"   MFMA busy / SQ busy (x4 SIMDs folded by the counter's own unit): {g('SQ_VALU_MFMA_BUSY_CYCLES') / g('SQ_BUSY_CYCLES'):.3f}")
    if g("SQ_WAVE_CYCLES") and g("SQ_WAIT_INST_ANY"):
        print(f"   wave-cycles waiting for an instruction's data / wave-cycles: {g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'):.3f}")
    if g("SQ_LDS_IDX_ACTIVE") and g("SQ_LDS_BANK_CONFLICT") is not None:
        print(f"   LDS bank-conflict cycles / LDS active cycles: {g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE'):.3f}")
    if g("GRBM_GUI_ACTIVE"):
        print(f"   GRBM_GUI_ACTIVE per dispatch {g('GRBM_GUI_ACTIVE'):.0f} cycles")
