# A/B of the whole-row epilogue (gemm_epilogue.h TR): kernel classes in situ with DZN_EPI_TR=0 / 1, then parity tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for v in 0 1 0 1; do
  echo "== DZN_EPI_TR=$v"
  DZN_EPI_TR=$v python scripts/probe_kernel_class.py 561 gemm_f32h 2>&1 | tail -8
done
} > gpurun_out/r6_epi_tr_ab.txt 2>&1
cat gpurun_out/r6_epi_tr_ab.txt
python -m pytest tests/test_ops_gpu.py tests/test_seg_gpu.py tests/test_f32h_grade_gpu.py -m gpu -x -q -n 4 2>&1 | tail -5
