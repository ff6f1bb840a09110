#!/bin/bash
# tools/gpu_split_pmc.sh [lib]: SQ counters of the f16x3 conv shapes of tools/split_bench.py (two passes, --kernel-trace only)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/sb_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
[ -n "$1" ] && export PRG_HIP_LIB=$GRAFT_REPO_ROOT/$1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/a -o r -- python $GRAFT_REPO_ROOT/tools/split_bench.py f16x3 > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/b -o r -- python $GRAFT_REPO_ROOT/tools/split_bench.py f16x3 > $OUT/b.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT/a/r_counter_collection.csv conv > $OUT/a_summary.txt
python tools/pmc_summary.py $OUT/b/r_counter_collection.csv conv > $OUT/b_summary.txt
cat $OUT/a_summary.txt $OUT/b_summary.txt
rm -rf $OUT/a $OUT/b
