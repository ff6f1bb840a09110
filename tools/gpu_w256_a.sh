#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 900 -k "one_at_a_time or fast_paths or benchmark_batch" -s 2>&1 | tail -40 > gpurun_out/w256_a_tests.log
tail -15 gpurun_out/w256_a_tests.log
for v in 1 0; do
  PRG_CONV_W256=$v timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e-files --no-drift > gpurun_out/w256_a_bench_$v.json 2> gpurun_out/w256_a_bench_$v.err
  python - <<PY
import json
r = json.load(open("gpurun_out/w256_a_bench_$v.json"))
print("W256=$v pairs/s", round(r["value"], 3), "ms/step", round(r["ms_per_step"], 1), "conv TF/s", round(r["roofline"]["achieved"], 1), "frac", round(r["roofline"]["frac"], 3))
PY
done
