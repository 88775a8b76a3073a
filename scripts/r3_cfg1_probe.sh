# round 3: BASELINE configs[1] (base-s80, 5 s windows, batch 32, segmentation only) + streaming / base-model tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r3h}; mkdir -p $O
( timeout 900 python -m pytest tests/test_seg_gpu.py tests/test_pipeline_gpu.py tests/test_ops_gpu.py -m gpu -x -q 2>&1 | tail -12 | cut -c1-400 ) > $O/tests.log 2>&1
echo "== tests"; cat $O/tests.log
timeout 600 python bench.py --model wavlm_base_s80_md --window 5 --batch 32 --minutes 2.7 --stage seg --steps 20 --warmup 3 --no-alt --no-e2e --no-cpu-baseline > $O/bench_cfg1.json 2> $O/bench_cfg1.err
timeout 600 python bench.py --model wavlm_base_s80_md --window 5 --batch 32 --minutes 2.7 --stage seg --steps 20 --warmup 3 --no-alt --no-e2e --no-cpu-baseline --no-profile > $O/bench_cfg1_noprof.json 2> $O/bench_cfg1_noprof.err
python - <<PY
import json
for f in ("$O/bench_cfg1.json", "$O/bench_cfg1_noprof.json"):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["config"]["windows_per_step"], (d.get("roofline_extra") or {}).get("non_kernel_frac"))
        for k in d["kernels"][:12]: print("   ", k)
    except Exception as e:
        print("bench failed", f, e); print(open(f.replace(".json",".err")).read()[-1500:])
PY
