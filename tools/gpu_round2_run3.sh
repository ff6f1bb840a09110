#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s --timeout 1500 > gpurun_out/r2_t3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_t3.log
python __graft_entry__.py smoke > gpurun_out/r2_smoke3.log 2>&1
python bench.py --steps 1 --warmup 1 > gpurun_out/r2_bench3.json 2> gpurun_out/r2_bench3.err
grep -v "^   \|^$" gpurun_out/r2_t3.log | tail -60; tail -3 gpurun_out/r2_smoke3.log; tail -5 gpurun_out/r2_bench3.err
