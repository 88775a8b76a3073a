#!/bin/bash
# hardware counters of the fused front end (conv01_ws_kernel / conv01_fused_kernel) inside one 561-window forward:
# three rocprofv3 --pmc passes (kernel trace only), summarised per kernel by scripts/pmc_kernel_summary.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_conv01; rm -rf $O; mkdir -p $O
CMD="python $R/scripts/probe_kernel_class.py 561 conv01"
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/p1 -- $CMD > /dev/null 2> $O/p1.err
timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O/p2 -- $CMD > /dev/null 2> $O/p2.err
timeout 400 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_WAVES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM --kernel-trace --output-format csv -d $O/p3 -- $CMD > /dev/null 2> $O/p3.err
for e in $O/p*.err; do tail -n 2 $e; done
python $R/scripts/pmc_kernel_summary.py $O conv01 | tee $R/gpurun_out/${1:-r6_conv01_pmc}.txt
rm -rf $O/p1 $O/p2 $O/p3
