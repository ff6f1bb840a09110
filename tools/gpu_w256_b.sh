#!/bin/bash
cd $GRAFT_REPO_ROOT
PRG_CONV_W256=1 bash tools/prof.sh w256_on > gpurun_out/w256_on_summary.txt 2>&1
cd $GRAFT_REPO_ROOT
PRG_CONV_W256=0 bash tools/prof.sh w256_off > gpurun_out/w256_off_summary.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_seq.py gpurun_out/w256_on/r_kernel_trace.csv gpurun_out/w256_off/r_kernel_trace.csv conv > gpurun_out/w256_seq.txt 2>&1
cat gpurun_out/w256_seq.txt
cp gpurun_out/w256_on/r_kernel_trace.csv gpurun_out/w256_on_trace.csv; cp gpurun_out/w256_off/r_kernel_trace.csv gpurun_out/w256_off_trace.csv; rm -rf gpurun_out/w256_on gpurun_out/w256_off
