#!/bin/bash
# the GPU kernel / segmentation / embedding tests under the CHECKED library (csrc/checked.h): every DZN_CHECK in the hand-scheduled
# kernels is live; tests/conftest.py fails the session if one fired.  Build first: python -m diarizen_amd.build --checked
export DZN_HIP_LIB="$(pwd)/diarizen_amd/lib/libdzn_hip_checked.so"
mkdir -p gpurun_out
rm -f gpurun_out/checked_build_status.txt
python -m pytest tests/test_ops_gpu.py tests/test_seg_gpu.py tests/test_emb_gpu.py -m gpu -q -x -k "not linkage and not vbx and not cdist and not clustering" 2>&1 | tail -15 > gpurun_out/${DZN_CHECKED_LOG:-r6_checked_build.log}
cat gpurun_out/checked_build_status.txt >> gpurun_out/${DZN_CHECKED_LOG:-r6_checked_build.log}
tail -6 gpurun_out/${DZN_CHECKED_LOG:-r6_checked_build.log}
