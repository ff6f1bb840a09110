#!/bin/bash
# round 4: first run of the f16x3 mode — conv / U-Net / chain parity, then its cost
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
timeout 900 python -m pytest tests/test_gpu_f16x3.py -x -q -m gpu -s > gpurun_out/r4b/tests.log 2>&1
tail -5 gpurun_out/r4b/tests.log
python bench.py --dtype f16x3 --timesteps 100 --steps 1 --warmup 1 --streams 1 --no-e2e-files --no-drift --no-configs4 --no-cpu-baseline --profile-transitions 10 > gpurun_out/r4b/bench_f16x3_t100.json 2> gpurun_out/r4b/bench_f16x3_t100.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4b/prof -o r -- python $GRAFT_REPO_ROOT/bench.py --dtype f16x3 --timesteps 10 --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-roofline --no-e2e-files --no-drift --no-configs4 > $GRAFT_REPO_ROOT/gpurun_out/r4b/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py gpurun_out/r4b/prof/r_kernel_trace.csv 24 > gpurun_out/r4b/prof_summary.txt 2>&1
rm -f gpurun_out/r4b/prof/*trace.csv
cut -c1-400 gpurun_out/r4b/bench_f16x3_t100.json; tail -3 gpurun_out/r4b/bench_f16x3_t100.err; head -60 gpurun_out/r4b/prof_summary.txt
