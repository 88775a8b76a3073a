#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_att; rm -rf $O; mkdir -p $O
python $R/scripts/bench_attention_pmc.py
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/p1 -- python $R/scripts/bench_attention_pmc.py > /dev/null 2> $O/p1.err
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/p2 -- python $R/scripts/bench_attention_pmc.py > /dev/null 2> $O/p2.err
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INSTS_MFMA SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_FLAT --kernel-trace --output-format csv -d $O/p3 -- python $R/scripts/bench_attention_pmc.py > /dev/null 2> $O/p3.err
for e in $O/p*.err; do tail -n 2 $e; done
python $R/scripts/bench_attention_pmc.py --summarise $O | tee $R/gpurun_out/r6_attention_pmc.txt
for p in p1 p2 p3; do f=$(find $O/$p -name "*counter_collection.csv" | head -1); echo "== $p $f"; head -3 "$f"; wc -l "$f"; done > $R/gpurun_out/r6_attention_pmc_debug.txt 2>&1; rm -rf $O/p1 $O/p2 $O/p3
