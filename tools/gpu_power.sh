#!/bin/bash
# samples socket power and shader clock while the benchmark's sampler loop runs (evidence for DESIGN.md 4.4)
cd $GRAFT_REPO_ROOT
( for i in $(seq 1 60); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk" | tr '\n' ' '; echo; sleep 0.5; done ) > gpurun_out/power_trace.txt &
SAMP=$!
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e-files --no-drift > gpurun_out/power_bench.json 2> gpurun_out/power_bench.err
wait $SAMP
awk '{for(i=1;i<=NF;i++){ if($i=="(W):") p=$(i+1); if($i ~ /^\(/ && $(i-1)=="S:") c=$i }} {print p, c}' gpurun_out/power_trace.txt | sort -n | tail -12
