#!/bin/bash
# tools/gpu_prof_mode.sh <dtype> <tag>: rocprofv3 kernel stats of 10 ancestral transitions at B=64, 128x128 in a precision mode
cd $GRAFT_REPO_ROOT
DT=$1; TAG=$2
mkdir -p gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof -o r -- python $GRAFT_REPO_ROOT/bench.py --dtype $DT --timesteps 10 --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-roofline --no-e2e-files --no-drift --no-configs4 --no-parity-mode > $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py gpurun_out/$TAG/prof/r_kernel_trace.csv auto > gpurun_out/$TAG/prof_summary.txt 2>&1
cp gpurun_out/$TAG/prof/r_kernel_stats.csv gpurun_out/$TAG/kernel_stats.csv
rm -rf gpurun_out/$TAG/prof
head -70 gpurun_out/$TAG/prof_summary.txt
