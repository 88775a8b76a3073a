cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2l
timeout 1500 python -m pytest tests/test_seg_gpu.py tests/test_decisions_gpu.py tests/test_pipeline_gpu.py tests/test_ops_gpu.py -m gpu -q -x 2>&1 | grep -v "amdgpu.ids" | tail -6
python - <<PY
import json
d=json.load(open("gpurun_out/decision_parity.json"))
for k,v in d["modes"].items(): print(k,v)
PY
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-e2e > gpurun_out/r2l/bench.json 2> gpurun_out/r2l/bench.err
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r2l/bench.json"))
    print(d["value"], d["ms_per_step"], d.get("unprofiled_ms_per_step"))
    for k in d["kernels"][:12]: print("  ",k["kernel"],k["launches"],k["ms_total"],k.get("tflops"),k.get("gbs"))
except Exception as e:
    print("failed", e); print(open("gpurun_out/r2l/bench.err").read()[-2500:])
PY
DZN_BENCH_ONE_DEVICE=1 DZN_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --steps 1 --warmup 1 --minutes 5 --batch 64 --scaling strong --no-cpu-baseline --no-alt --no-e2e --no-profile 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -3 | cut -c1-600
