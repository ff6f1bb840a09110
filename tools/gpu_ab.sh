#!/bin/bash
# same-box A/B of two library builds: tools/gpu_ab.sh <filter>   (old = pointreggpt_amd/libprg_old.so)
cd $GRAFT_REPO_ROOT
bash tools/prof.sh ab_new > gpurun_out/ab_new_summary.txt 2>&1
cd $GRAFT_REPO_ROOT
PRG_HIP_LIB=$GRAFT_REPO_ROOT/pointreggpt_amd/libprg_old.so bash tools/prof.sh ab_old > gpurun_out/ab_old_summary.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_seq.py gpurun_out/ab_new/r_kernel_trace.csv gpurun_out/ab_old/r_kernel_trace.csv "${1:-conv}" > gpurun_out/ab_seq.txt 2>&1
grep -E "true|sum" gpurun_out/ab_seq.txt
rm -rf gpurun_out/ab_new gpurun_out/ab_old
