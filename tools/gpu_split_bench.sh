#!/bin/bash
# tools/gpu_split_bench.sh [dtype] [lib]: per-launch kernel times of the conv shapes (rocprofv3 kernel trace)
cd $GRAFT_REPO_ROOT
DT=${1:-f16x3}
OUT=$GRAFT_REPO_ROOT/gpurun_out/sb_$DT
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
[ -n "$2" ] && export PRG_HIP_LIB=$GRAFT_REPO_ROOT/$2
rocprofv3 --kernel-trace --output-format csv -d $OUT -o r -- python $GRAFT_REPO_ROOT/tools/split_bench.py $DT > $OUT/run.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/split_bench.py --summarise $OUT/r_kernel_trace.csv | tee $OUT/summary.txt
tail -3 $OUT/run.log
rm -f $OUT/r_kernel_trace.csv
