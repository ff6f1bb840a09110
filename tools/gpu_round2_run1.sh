#!/bin/bash
# first GPU call of round 2: parity tests with the new fixtures (verbose), smoke, 256x256 DDIM bench + kernel trace, default bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s -x --timeout 1500 > gpurun_out/r2_t1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_t1.log
python __graft_entry__.py smoke > gpurun_out/r2_smoke1.log 2>&1
python bench.py --size 256 --sampling-steps 250 --batch 16 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2_bench256_b16.json 2> gpurun_out/r2_bench256_b16.err
bash tools/prof.sh r2_prof256 --size 256 --batch 16 > gpurun_out/r2_prof256_summary.txt 2>&1
cd $GRAFT_REPO_ROOT
python bench.py --steps 1 --warmup 1 > gpurun_out/r2_bench_default0.json 2> gpurun_out/r2_bench_default0.err
tail -40 gpurun_out/r2_t1.log
