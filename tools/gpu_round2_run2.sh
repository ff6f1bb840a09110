#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s -x --timeout 1500 > gpurun_out/r2_t2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_t2.log
python __graft_entry__.py smoke > gpurun_out/r2_smoke2.log 2>&1
bash tools/prof.sh r2_prof128b > gpurun_out/r2_prof128b_summary.txt 2>&1
cd $GRAFT_REPO_ROOT
bash tools/prof.sh r2_prof256b --size 256 --batch 16 > gpurun_out/r2_prof256b_summary.txt 2>&1
cd $GRAFT_REPO_ROOT
grep -v "^   \|^$" gpurun_out/r2_t2.log | tail -40; tail -3 gpurun_out/r2_smoke2.log
