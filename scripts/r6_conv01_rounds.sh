# conv01_ws_kernel fully persistent (1 workgroup per CU for the whole launch) vs 4 / 16 successive workgroups per CU: the headline
# steps on one stream and on two (bench.py --streams 2)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for r in 1 4 16; do for st in 1 2; do
  DZN_CONV01_ROUNDS=$r timeout 600 python bench.py --steps 3 --warmup 1 --streams $st --no-cpu-baseline --no-alt --no-e2e --no-config1 --no-power 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = {x['kernel']: x for x in d.get('kernels', [])}
c = k.get('conv01_fused', {})
print('rounds $r streams $st value', d['value'], 'device_value', d.get('device_value'), 'conv01 ms/launch', round(c.get('ms_total', 0) / max(c.get('launches', 1), 1), 3))"
done; done
} > gpurun_out/r6_conv01_rounds.txt 2>&1
cat gpurun_out/r6_conv01_rounds.txt
