#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 900 -k "fast_paths or benchmark_batch or golden or unet" 2>&1 | tail -4
bash tools/prof.sh la_a > gpurun_out/la_a_summary.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_seq.py gpurun_out/la_a/r_kernel_trace.csv la_ | awk '{print}' 
head -2 gpurun_out/la_a_summary.txt
rm -rf gpurun_out/la_a
