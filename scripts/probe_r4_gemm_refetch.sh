# r4: which operand does the dominant contraction re-fetch over the fabric?  FETCH_SIZE per launch against the algorithmic
# reads (A + residual + weight planes) for N = 128 .. 2048 at M = 149226, K = 1024 (one column tile .. sixteen):
# A re-fetch grows with the number of column tiles from N = 256 on; weight-plane thrash needs the planes (N x K x 4 B) to
# approach the 4 MB L2, i.e. shows from N ~ 512-1024 on.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r4rf}; mkdir -p $O
cd /tmp
timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc -- python $R/scripts/bench_gemm_cfgs.py 128x128w4 149226,128,1024 149226,256,1024 149226,512,1024 149226,1024,1024 149226,2048,1024 > $O/run.txt 2> $O/pmc.err
F=$(find $O/pmc -name '*counter_collection.csv' | head -1)
python - <<PY | tee $O/refetch.txt
import csv, collections
vals = collections.OrderedDict()
for r in csv.DictReader(open("$F")):
    n = r["Kernel_Name"]
    if "gemm_split_kernel<128, 128, 4, 1, 2, 2, 2" not in n or r["Counter_Name"] != "FETCH_SIZE": continue
    g = r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", "?")
    vals.setdefault(g, []).append(float(r["Counter_Value"]))
M, K = 149226, 1024
for (g, v), N in zip(vals.items(), (128, 256, 512, 1024, 2048)):
    fetch = 2 * 1024 * sum(v) / len(v)
    alg = M * K * 4 + M * N * 4 + N * K * 4
    print(f"N={N:5d} grid={g:>9s} launches={len(v):3d} FETCH x2 = {fetch/1e9:6.3f} GB  algorithmic reads (A + R + W) = {alg/1e9:6.3f} GB  ratio {fetch/alg:5.2f}  excess {((fetch-alg)/1e9):6.3f} GB = {(fetch-alg)/(M*K*4):5.2f} x A = {(fetch-alg)/(N*K*4)/ (M/128):6.3f} x W per row block")
PY
find $O/pmc -name '*.csv' -size +1M -delete
