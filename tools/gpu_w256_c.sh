#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 900 -k "one_at_a_time or fast_paths or benchmark_batch" -s 2>&1 | tail -40 > gpurun_out/w256_c_tests.log
tail -4 gpurun_out/w256_c_tests.log
bash tools/prof.sh w256_c > gpurun_out/w256_c_summary.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_seq.py gpurun_out/w256_c/r_kernel_trace.csv conv > gpurun_out/w256_c_seq.txt 2>&1
grep -E "c64|sum" gpurun_out/w256_c_seq.txt
head -3 gpurun_out/w256_c_summary.txt
rm -rf gpurun_out/w256_c
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e-files --no-drift > gpurun_out/w256_c_bench.json 2> gpurun_out/w256_c_bench.err
python - <<PY
import json
r = json.load(open("gpurun_out/w256_c_bench.json"))
print("pairs/s", round(r["value"], 3), "ms/step", round(r["ms_per_step"], 1), "conv TF/s", round(r["roofline"]["achieved"], 1), "frac", round(r["roofline"]["frac"], 3))
PY
