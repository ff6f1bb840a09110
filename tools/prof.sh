#!/bin/bash
# usage: tools/prof.sh <tag> [bench args...]   (run on the GPU box through gpurun)
set -e
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --sampling-steps 20 --no-cpu-baseline --no-roofline --no-e2e-files --no-drift --no-configs4 "$@" > $OUT.log 2>&1 || true
python $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT/r_kernel_trace.csv auto
