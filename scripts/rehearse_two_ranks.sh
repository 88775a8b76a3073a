# two ranks on the ONE visible GPU (gloo staging: RCCL refuses duplicate devices) through the whole bench, incl. the
# strong-scaling end-to-end leg (shared PCM file written by the ranks, range decode, gather with the partition assertion)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-rehearse}; mkdir -p $O
DZN_BENCH_ONE_DEVICE=1 DZN_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --steps 1 --warmup 1 --minutes 3 --batch 64 --strong-minutes 12 --no-alt \
  > $O/bench_2ranks.json 2> $O/bench_2ranks.err
tail -3 $O/bench_2ranks.err | cut -c1-300
python - <<PY
import json
d=json.loads(open("$O/bench_2ranks.json").read().strip().splitlines()[-1])
print(d["n_gpus"], d["value"], d["scaling"]); print(d.get("strong_scaling_e2e"))
PY
timeout 600 python bench.py --steps 1 --warmup 1 --minutes 3 --batch 64 --strong-minutes 12 --no-alt --no-cpu-baseline > $O/bench_1rank.json 2> $O/bench_1rank.err
python - <<PY
import json
d=json.loads(open("$O/bench_1rank.json").read().strip().splitlines()[-1])
print(d["n_gpus"], d["value"]); print(d.get("strong_scaling_e2e"))
PY
