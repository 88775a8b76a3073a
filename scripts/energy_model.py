"""Where the package power of the headline step goes, to first order (round 4).

    python scripts/energy_model.py [bench.json] [pmc.json]

Inputs: the bench line's per-kernel table (time per class), the PMC table (L2-miss traffic per launch = what crosses the fabric
to the Infinity Cache / HBM, FETCH_SIZE x 2 + WRITE_SIZE) and three measured anchors from profiles/r4_power_cap.txt:
  idle                                   239 W
  1 GiB copy, 5.5 TB/s over the fabric  1040 W   ->  (1040 - 239) / 5.5 = 146 W per TB/s of fabric traffic (an UPPER figure for
                                                     traffic the Infinity Cache answers; the copy streams from HBM)
  package cap                           1400 W
The model charges every kernel class idle + 146 W x its fabric TB/s and calls the rest of what the step draws (bench `power`,
~1315 W mean) "compute" (matrix pipe, VALU, LDS, L2).  It is an attribution, not a measurement per kernel — and an UPPER bound on
the fabric side: the per-TB/s anchor is an HBM-streaming copy, while much of the L2-miss traffic of the contractions is answered by the
256 MB Infinity Cache, where halving it changed neither time nor the step (profiles/r4_gemm_refetch_probe.txt)."""
from __future__ import annotations

import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
IDLE_W, W_PER_TBS, CAP_W = 239.0, (1040.0 - 239.0) / 5.5, 1400.0


def main():
    bench = Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "profiles" / "r4_final_bench_with_power_field.json"
    pmc = Path(sys.argv[2]) if len(sys.argv) > 2 else ROOT / "profiles" / "r4_pmc_f32h_30min_b576.json"
    sys.path.insert(0, str(ROOT))
    import bench as B           # PMC_SYMBOLS / pmc_lookup: the class -> kernel symbol map of the bench itself
    d = json.loads(bench.read_text().strip().splitlines()[-1])
    table = json.loads(pmc.read_text())
    steps = d["steps"]
    mean_w = (d.get("power") or {}).get("package_w", {}).get("mean")
    rows, tot_ms, tot_gb = [], 0.0, 0.0
    for k in d["kernels"]:
        per_launch = B.pmc_lookup(table, k["kernel"], "hbm_bytes_per_launch")
        ms = k["ms_total"] / steps
        gb = (per_launch or 0) * k["launches"] / steps / 1e9
        tot_ms += ms
        tot_gb += gb
        rows.append((ms, k["kernel"], gb, per_launch is not None))
    print(f"# {bench.name}: {d['value']} audio-s/s, {d['ms_per_step']} ms per step, package {mean_w} W mean of {CAP_W:.0f} W")
    print(f"# anchors: idle {IDLE_W:.0f} W, {W_PER_TBS:.0f} W per TB/s of fabric traffic")
    print(f"{'kernel class':28s} {'ms/step':>8s} {'share':>6s} {'fabric GB/step':>15s} {'TB/s':>6s} {'fabric W while it runs':>23s}")
    for ms, name, gb, known in sorted(rows, reverse=True)[:14]:
        tbs = gb / ms if ms > 0 else 0.0          # GB / ms = TB/s
        note = f"{W_PER_TBS * tbs:8.0f}" if known else "   (no PMC row)"
        print(f"{name:28s} {ms:8.1f} {100 * ms / tot_ms:5.1f}% {gb:15.1f} {tbs:6.2f} {note:>23s}")
    tbs = tot_gb / tot_ms
    fabric_w = W_PER_TBS * tbs
    print(f"{'whole step (profiled kernels)':28s} {tot_ms:8.1f} {'':6s} {tot_gb:15.1f} {tbs:6.2f} {fabric_w:23.0f}")
    if mean_w:
        dyn = mean_w - IDLE_W
        print(f"# dynamic power {dyn:.0f} W = fabric ~{fabric_w:.0f} W ({100 * fabric_w / dyn:.0f} %) + compute ~{dyn - fabric_w:.0f} W; "
              f"every 100 GB/step removed from the fabric frees ~{W_PER_TBS * 0.1 / (tot_ms / 1e3):.0f} W "
              f"(~{100 * W_PER_TBS * 0.1 / (tot_ms / 1e3) / (dyn - fabric_w):.1f} % of the compute budget)")


if __name__ == "__main__":
    main()
