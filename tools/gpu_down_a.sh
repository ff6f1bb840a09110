#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 900 -k "downsample or one_at_a_time or fast_paths or benchmark_batch" 2>&1 | tail -15
for v in 1 0; do
  PRG_CONV_DOWN_W256=$v timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e-files --no-drift > gpurun_out/down_bench_$v.json 2> gpurun_out/down_bench_$v.err
  python - <<PY
import json
r = json.load(open("gpurun_out/down_bench_$v.json"))
print("DOWN=$v pairs/s", round(r["value"], 3), "ms/step", round(r["ms_per_step"], 1), "conv TF/s", round(r["roofline"]["achieved"], 1), "frac", round(r["roofline"]["frac"], 3))
PY
done
bash tools/prof.sh down_on > gpurun_out/down_on_summary.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_seq.py gpurun_out/down_on/r_kernel_trace.csv conv | grep -E "^ +(14|28|42) |sum"
rm -rf gpurun_out/down_on
