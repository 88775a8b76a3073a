# r4 GEMM probe: 32x32x16 MFMA forms vs the 16x16x32 production tiles (one process per shape set), then the pipeline A/B
# needs a DZN_TUNING build of gemm_split.hip linked as diarizen_amd/lib/libdzn_hip_tuning.so:
#   hipcc <FLAGS of diarizen_amd/build.py> -DDZN_TUNING -c diarizen_amd/csrc/gemm_split.hip -o /tmp/gs_tuning.o
#   hipcc -shared -fPIC --offload-arch=gfx950 -o diarizen_amd/lib/libdzn_hip_tuning.so $(ls diarizen_amd/build/*.o | grep -v gemm_split.hip.o) /tmp/gs_tuning.o
# (DZN_TUNING=1 python -m diarizen_amd.build builds the whole library that way)
set -x
O=gpurun_out/${1:-r4b}; mkdir -p $O
export DZN_HIP_LIB=$PWD/diarizen_amd/lib/libdzn_hip_tuning.so
timeout 900 python scripts/bench_gemm_cfgs.py 128x128w4,m32_128x128,m32_128x128p,m32_128x128w22,m32_256x128,m32_256x128w8 149226,1024,1024 149226,960,1024 149226,1920,1024 149226,384,1024 2>&1 | grep -v "f16 M" > $O/gemm_m32_long_k.txt
timeout 600 python scripts/bench_gemm_cfgs.py 128x64,m32_128x64,m32_128x64p,m32_128x128 149226,1024,256 149226,1024,512 149226,1024,128 2>&1 | grep -v "f16 M" > $O/gemm_m32_short_k.txt
unset DZN_HIP_LIB
cat $O/gemm_m32_long_k.txt $O/gemm_m32_short_k.txt
for v in 0 1 3; do
DZN_GEMM_M32=$v timeout 300 python bench.py --steps 3 --warmup 1 --no-alt --no-e2e --no-config1 --no-cpu-baseline > $O/bench_m32_$v.json 2> $O/bench_m32_$v.err
python - <<PY
import json
d=json.loads(open("$O/bench_m32_$v.json").read().strip().splitlines()[-1])
print("M32=$v", d["value"], d["ms_per_step"], [(k["kernel"], round(k["ms_total"],1), k.get("tflops")) for k in d["kernels"][:4]])
PY
done
