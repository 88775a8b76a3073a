"""Merge rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, MfmaUtil; one pass each) into one table per kernel symbol:
    hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB   (gfx950: FETCH_SIZE counts 64 B per 128-B request on wide
                           coalesced reads -> doubled, MI355X_MICROARCH.md HBM section; both counters are in KiB)
    mfma_util_pct        = mean MfmaUtil (percent of SIMD cycles with the matrix pipe busy)
    eff_clock_ghz        = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration   (r3, optional 4th pass: the clock the chip held
                           while the kernel ran — it sits well under the 2.4 GHz peak on the matrix-heavy kernels)
    mfma_busy_frac       = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)
usage: pmc_summary.py out.json fetch.csv write.csv mfma.csv [clock.csv]"""
import collections
import csv
import json
import statistics
import sys

outp, fetch, write, mfma = sys.argv[1:5]


def clean(n):
    return n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]


out = collections.defaultdict(dict)
for tag, path in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write), ("MfmaUtil", mfma)):
    vals = collections.defaultdict(list)
    try:
        rows = csv.DictReader(open(path))
    except OSError:
        continue
    for r in rows:
        if r["Counter_Name"] == tag:
            vals[clean(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for n, v in vals.items():
        out[n]["launches"] = len(v)
        if tag == "MfmaUtil":
            out[n]["mfma_util_pct"] = round(statistics.fmean(v), 2)
        else:
            out[n][tag + "_KiB_per_launch"] = round(statistics.fmean(v), 1)
for n, e in out.items():
    if "FETCH_SIZE_KiB_per_launch" in e or "WRITE_SIZE_KiB_per_launch" in e:
        e["hbm_bytes_per_launch"] = int((2.0 * e.get("FETCH_SIZE_KiB_per_launch", 0.0) + e.get("WRITE_SIZE_KiB_per_launch", 0.0)) * 1024)
if len(sys.argv) > 5:
    act, busy, dur = (collections.defaultdict(list) for _ in range(3))
    try:
        rows = list(csv.DictReader(open(sys.argv[5])))
    except OSError:
        rows = []
    for r in rows:
        n = clean(r["Kernel_Name"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            act[n].append(float(r["Counter_Value"]))
            if "Start_Timestamp" in r and "End_Timestamp" in r:
                dur[n].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        elif r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
            busy[n].append(float(r["Counter_Value"]))
    for n in act:
        cyc = statistics.fmean(act[n]) / 8.0
        if dur[n]:
            out[n]["eff_clock_ghz"] = round(cyc / statistics.fmean(dur[n]), 3)
            out[n]["pmc_pass_duration_us"] = round(statistics.fmean(dur[n]) / 1e3, 1)
        if busy[n] and cyc > 0:
            out[n]["mfma_busy_frac"] = round(statistics.fmean(busy[n]) / (cyc * 1024.0), 4)
json.dump(out, open(outp, "w"), indent=1, sort_keys=True)
for n, e in sorted(out.items(), key=lambda kv: -kv[1].get("hbm_bytes_per_launch", 0) * kv[1].get("launches", 0))[:14]:
    print(n, e)
