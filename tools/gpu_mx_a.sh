#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 1200 -k "mxfp8" -s 2>&1 | tail -15
for dt in mxfp8 bf16; do
  timeout 600 python bench.py --dtype $dt --steps 3 --warmup 1 --no-cpu-baseline --no-e2e-files --no-drift > gpurun_out/mx_bench_$dt.json 2> gpurun_out/mx_bench_$dt.err
  python - <<PY
import json
r = json.load(open("gpurun_out/mx_bench_$dt.json"))
print("$dt pairs/s", round(r["value"], 3), "ms/step", round(r["ms_per_step"], 1), "conv TF/s", round(r["roofline"]["achieved"], 1), "frac", round(r["roofline"]["frac"], 3))
PY
done
bash tools/prof.sh mx_on --dtype mxfp8 > gpurun_out/mx_on_summary.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_seq.py gpurun_out/mx_on/r_kernel_trace.csv conv | grep -E "w256|sum"
rm -rf gpurun_out/mx_on
