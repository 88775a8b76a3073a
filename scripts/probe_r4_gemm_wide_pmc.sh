# r4: the 256 x 256 one-wavefront-per-SIMD contraction vs the 128 x 128 production tile: data-dependence (zero A), and the
# clock + matrix-pipe-busy counters of both forms run back to back (one PMC pass; K = 1024 and K = 4096)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r4wp}; mkdir -p $O
cd $R
timeout 300 python scripts/probe_gemm_bound.py 128x128w4,wide > $O/bound.txt 2>&1; grep cfg= $O/bound.txt
cd /tmp
P="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/pmc -- python $R/scripts/bench_gemm_cfgs.py 128x128w4,wide 149226,1024,1024 149226,1024,4096 > $O/pmc_run.txt 2> $O/pmc.err
F=$(find $O/pmc -name '*counter_collection.csv' | head -1)
python - <<PY | tee $O/pmc.txt
import csv, collections
vals = collections.defaultdict(list)
for r in csv.DictReader(open("$F")):
    n = r["Kernel_Name"]
    if "gemm_split_kernel<128, 128, 4, 1, 2, 2, 2" not in n and "gemm_wide" not in n: continue
    n = "wide256" if "gemm_wide" in n else "prod128"
    vals[(n, r["Counter_Name"])].append(float(r["Counter_Value"]))
# launches alternate shapes: the first half of each kernel's launches is K = 1024, the second K = 4096 (f32h only for wide)
for (n, c), v in sorted(vals.items()):
    h = len(v) // 2
    print(f"{n:8s} {c:28s} n={len(v):3d} mean_first_half={sum(v[:h])/max(h,1):.4g} mean_second_half={sum(v[h:])/max(len(v)-h,1):.4g}")
PY
find $O/pmc -name '*.csv' -size +2M -delete; find $O/pmc -name '*kernel_trace.csv' | head -1 | xargs -I{} cp {} $O/kernel_trace.csv
