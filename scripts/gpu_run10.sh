cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2j
DZN_PROFILE_SHAPES=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt --no-e2e > gpurun_out/r2j/shapes.json 2> gpurun_out/r2j/shapes.err
python - <<PY
import json
d=json.load(open("gpurun_out/r2j/shapes.json"))
ks=d["kernels"]; tot=sum(k["ms_total"] for k in ks)
print(d["value"], d["ms_per_step"], tot)
for k in ks[:45]:
    print(f"{k['kernel']:58s} n={k['launches']:5d} ms={k['ms_total']:9.2f} {100*k['ms_total']/tot:5.2f}% avg_us={1000*k['ms_total']/k['launches']:8.1f} tf={k.get('tflops','')}")
PY
