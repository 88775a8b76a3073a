# in-loop operand split of the fp16x2 contraction on plain VALU instructions (split.h DZN_SPLIT_PLAIN): kernel classes in situ,
# then the contraction / model parity tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for i in 1 2; do
python scripts/probe_kernel_class.py 561 gemm_f32h 2>&1 | grep -v amdgpu.ids | tail -6
done
} > gpurun_out/${1:-r6_split_plain_ab}.txt 2>&1
cat gpurun_out/${1:-r6_split_plain_ab}.txt
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_f32h_grade_gpu.py -m gpu -x -q -k "gemm or f32h or split" 2>&1 | tail -4
