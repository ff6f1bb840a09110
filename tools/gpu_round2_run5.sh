#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -s --timeout 1500 -k "fast_paths or dim64 or chain8 or benchmark_batch" > gpurun_out/r2_t5.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_t5.log
grep -v "^   \|^$" gpurun_out/r2_t5.log | tail -40
bash tools/prof.sh r2_prof128c > gpurun_out/r2_prof128c_summary.txt 2>&1
head -30 $GRAFT_REPO_ROOT/gpurun_out/r2_prof128c_summary.txt
