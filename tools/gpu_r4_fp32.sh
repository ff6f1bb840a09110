#!/bin/bash
# round 4, first GPU call: what does the parity mode (fp32) cost?  + the f16 split probe
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
tools/micro/f16_split_probe > gpurun_out/r4a/f16_split_probe.txt 2>&1
python bench.py --dtype fp32 --timesteps 100 --steps 1 --warmup 1 --streams 1 --no-e2e-files --no-drift --no-configs4 --no-cpu-baseline --profile-transitions 10 > gpurun_out/r4a/bench_fp32_t100.json 2> gpurun_out/r4a/bench_fp32_t100.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4a/prof -o r -- python $GRAFT_REPO_ROOT/bench.py --dtype fp32 --timesteps 10 --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-roofline --no-e2e-files --no-drift --no-configs4 > $GRAFT_REPO_ROOT/gpurun_out/r4a/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py gpurun_out/r4a/prof/r_kernel_trace.csv 40 > gpurun_out/r4a/prof_summary.txt 2>&1
rm -f gpurun_out/r4a/prof/*trace.csv
cat gpurun_out/r4a/f16_split_probe.txt; cat gpurun_out/r4a/bench_fp32_t100.json | cut -c1-1500; head -50 gpurun_out/r4a/prof_summary.txt
