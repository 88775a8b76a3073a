# r4: the FETCH_SIZE pass of scripts/probe_r4_gemm_refetch.sh with the column-group tile order (DZN_GEMM_NGROUPS=auto)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r4rg}; mkdir -p $O
cd /tmp
DZN_GEMM_NGROUPS=auto timeout 45 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc -- python $R/scripts/bench_gemm_cfgs.py 128x128w4 149226,1024,1024 149226,2048,1024 > $O/run.txt 2> $O/pmc.err
F=$(find $O/pmc -name '*counter_collection.csv' | head -1)
python - <<PY | tee $O/refetch_groups.txt
import csv, collections
vals = collections.OrderedDict()
for r in csv.DictReader(open("$F")):
    n = r["Kernel_Name"]
    if "gemm_split_kernel<128, 128, 4, 1, 2, 2, 2" not in n or r["Counter_Name"] != "FETCH_SIZE": continue
    vals.setdefault(r["Grid_Size"], []).append(float(r["Counter_Value"]))
M, K = 149226, 1024
for (g, v), N in zip(vals.items(), (1024, 2048)):
    fetch = 2 * 1024 * sum(v) / len(v)
    alg = M * K * 4 + M * N * 4 + N * K * 4
    print(f"NGROUPS=auto N={N:5d} launches={len(v):3d} FETCH x2 = {fetch/1e9:6.3f} GB  algorithmic reads = {alg/1e9:6.3f} GB  ratio {fetch/alg:5.2f}")
PY
grep "cfg=" $O/run.txt | grep f32h
find $O/pmc -name '*.csv' -size +1M -delete
