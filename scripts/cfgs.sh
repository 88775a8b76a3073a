run() { shape=$1; shift; for c in "$@"; do r=$(DZN_GEMM_CFG=$c timeout 100 python scripts/bench_gemm_split.py $shape 2>&1 | grep "f32s" | awk '{print $5,$6,$7,$8}'); echo "$shape $c: $r"; done; }
run 1021440,64,576 128x64 128x64v 128x64vs3 256x64w8
run 204288,1024,256 128x64 128x64v 128x128 128x128v 256x64w8
run 2042880,32,288 128x32 128x32s4 256x32w8
run 817152,160,1536 128x96 128x160 128x192 128x64v
run 1021440,128,1152 128x128 128x128v 128x64v
run 102144,960,1024 128x128 128x96 128x192 128x128v
