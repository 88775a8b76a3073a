# round 3: float4 LayerNorm + cheaper profiler: tests, configs[1], default line
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r3m}; mkdir -p $O
( timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_seg_gpu.py tests/test_emb_gpu.py tests/test_properties_gpu.py -m gpu -x -q 2>&1 | tail -12 | cut -c1-400 ) > $O/tests.log 2>&1
echo "== tests"; cat $O/tests.log
timeout 300 python scripts/probe_kernel_class.py 374 layernorm stem gate row_stats 2>&1 | tail -6
timeout 600 python bench.py --model wavlm_base_s80_md --window 5 --batch 32 --minutes 2.7 --stage seg --steps 20 --warmup 3 --no-alt --no-e2e --no-cpu-baseline > $O/bench_cfg1.json 2> $O/bench_cfg1.err
timeout 600 python bench.py --model wavlm_base_s80_md --window 5 --batch 32 --minutes 2.7 --stage seg --steps 20 --warmup 3 --no-alt --no-e2e --no-cpu-baseline --no-profile > $O/bench_cfg1_noprof.json 2> $O/bench_cfg1_noprof.err
( time timeout 900 python bench.py --no-alt --no-e2e --no-cpu-baseline ) > $O/bench_default.json 2> $O/bench_default.err
python - <<PY
import json
for f in ("$O/bench_cfg1.json", "$O/bench_cfg1_noprof.json", "$O/bench_default.json"):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d.get("unprofiled_ms_per_step"), (d.get("roofline_extra") or {}).get("non_kernel_frac"))
        for k in d["kernels"][:8]: print("   ", {a:b for a,b in k.items() if a!='alg_bytes_per_launch'})
    except Exception as e:
        print("bench failed", f, e); print(open(f.replace(".json",".err")).read()[-1500:])
PY
