# round 3: BASELINE configs[1] (base-s80, 5 s windows, batch 32, segmentation only) + base-model tests + default line
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r3i}; mkdir -p $O
( timeout 900 python -m pytest tests/test_seg_gpu.py tests/test_pipeline_gpu.py tests/test_dist_gpu.py -m gpu -x -q 2>&1 | tail -12 | cut -c1-400 ) > $O/tests.log 2>&1
echo "== tests"; cat $O/tests.log
DZN_PROFILE_SHAPES=1 timeout 600 python bench.py --model wavlm_base_s80_md --window 5 --batch 32 --minutes 2.7 --stage seg --steps 10 --warmup 3 --no-alt --no-e2e --no-cpu-baseline > $O/bench_cfg1_shapes.json 2> $O/bench_cfg1_shapes.err
timeout 600 python bench.py --model wavlm_base_s80_md --window 5 --batch 32 --minutes 2.7 --stage seg --steps 20 --warmup 3 --no-alt --no-e2e --no-cpu-baseline > $O/bench_cfg1.json 2> $O/bench_cfg1.err
timeout 600 python bench.py --model wavlm_base_s80_md --window 5 --batch 32 --minutes 2.7 --stage seg --steps 20 --warmup 3 --no-alt --no-e2e --no-cpu-baseline --no-profile > $O/bench_cfg1_noprof.json 2> $O/bench_cfg1_noprof.err
python - <<PY
import json
for f in ("$O/bench_cfg1_shapes.json", "$O/bench_cfg1.json", "$O/bench_cfg1_noprof.json"):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["config"]["windows_per_step"], (d.get("roofline_extra") or {}).get("non_kernel_frac"))
        for k in d["kernels"][:14]: print("   ", {a:b for a,b in k.items() if a!='alg_bytes_per_launch'})
    except Exception as e:
        print("bench failed", f, e); print(open(f.replace(".json",".err")).read()[-1500:])
PY
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
python - <<PY
import json
try:
    d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
    print("default", d["value"], d["ms_per_step"], d.get("e2e"), {k:(v["value"],v["steps"]) for k,v in d.get("other_fp32_modes",{}).items()}, d["roofline"])
except Exception as e:
    print("bench failed", e); print(open("$O/bench_default.err").read()[-2500:])
PY
tail -4 $O/bench_default.err
