#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -s --timeout 1500 -k "fast_paths or dim64 or chain8 or benchmark_batch or cli" > gpurun_out/r2_t8.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_t8.log
grep -v "^   \|^$" gpurun_out/r2_t8.log | grep -v "no_ws\|no_fused\|no_kshift" | tail -16
bash tools/prof.sh r2_prof128f > gpurun_out/r2_prof128f_summary.txt 2>&1
cd $GRAFT_REPO_ROOT; head -20 gpurun_out/r2_prof128f_summary.txt; grep "c64\|gn_coeff" gpurun_out/r2_prof128f_summary.txt | tail -6
