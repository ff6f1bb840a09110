#!/bin/bash
# usage: tools/pmc_insts.sh <tag>   (GPU box, through gpurun): instruction-class counts per kernel of the headline workload
# (one pass, --kernel-trace only): how many VALU / SALU / LDS / VMEM instructions each kernel issues per MFMA.
set -e
TAG=$1
ROOT=$GRAFT_REPO_ROOT
ARGS="--steps 1 --warmup 0 --streams 1 --sampling-steps 2 --no-cpu-baseline --no-roofline --no-e2e-files --no-drift --no-configs4 --no-parity-mode"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES --kernel-trace --output-format csv -d $ROOT/gpurun_out/${TAG}_INSTS -o r -- python $ROOT/bench.py $ARGS > $ROOT/gpurun_out/${TAG}_INSTS.log 2>&1 || true
python $ROOT/tools/pmc_summary.py $ROOT/gpurun_out/${TAG}_INSTS/r_counter_collection.csv > $ROOT/gpurun_out/${TAG}_INSTS_summary.txt
head -40 $ROOT/gpurun_out/${TAG}_INSTS_summary.txt
