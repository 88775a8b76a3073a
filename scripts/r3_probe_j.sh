# round 3: base-model positional conv on the planes, GroupNorm tracker, fused-frontend phase probe
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r3j}; mkdir -p $O
( timeout 900 python -m pytest tests/test_seg_gpu.py tests/test_ops_gpu.py -m gpu -x -q 2>&1 | tail -12 | cut -c1-400 ) > $O/tests.log 2>&1
echo "== tests"; cat $O/tests.log
timeout 600 python bench.py --model wavlm_base_s80_md --window 5 --batch 32 --minutes 2.7 --stage seg --steps 20 --warmup 3 --no-alt --no-e2e --no-cpu-baseline > $O/bench_cfg1.json 2> $O/bench_cfg1.err
timeout 600 python bench.py --model wavlm_base_s80_md --window 5 --batch 32 --minutes 2.7 --stage seg --steps 20 --warmup 3 --no-alt --no-e2e --no-cpu-baseline --no-profile > $O/bench_cfg1_noprof.json 2> $O/bench_cfg1_noprof.err
python - <<PY
import json
for f in ("$O/bench_cfg1.json", "$O/bench_cfg1_noprof.json"):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["config"]["windows_per_step"], (d.get("roofline_extra") or {}).get("non_kernel_frac"), d.get("roofline"))
        for k in d["kernels"][:10]: print("   ", {a:b for a,b in k.items() if a!='alg_bytes_per_launch'})
    except Exception as e:
        print("bench failed", f, e); print(open(f.replace(".json",".err")).read()[-1500:])
PY
for abl in 0 1 2; do
  echo "== conv01 abl=$abl"; DZN_CONV01_ABL=$abl timeout 300 python scripts/probe_kernel_class.py 374 conv01 attention conv3x3 layernorm stem 2>&1 | tail -8
done
