#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -s --timeout 1200 -k "one_at_a_time or mxfp8" 2>&1 | grep -v "^$" | tail -6
python bench.py --dtype mxfp8 --steps 1 --warmup 1 --no-cpu-baseline --no-e2e-files --no-drift > gpurun_out/r2_bench_mxfp8_128.json 2> gpurun_out/r2_bench_mxfp8_128.err
python bench.py --dtype mxfp8 --size 256 --sampling-steps 250 --batch 16 --steps 1 --warmup 1 --no-cpu-baseline --no-e2e-files --no-drift > gpurun_out/r2_bench_mxfp8_256.json 2> gpurun_out/r2_bench_mxfp8_256.err
python - <<'PY'
import json
for n in ("128","256"):
    try:
        r=json.load(open(f"gpurun_out/r2_bench_mxfp8_{n}.json")); print(n, "pairs/s", r["value"], "conv TF/s", r["roofline"]["achieved"], "frac", r["roofline"]["frac"], "share", r["roofline"]["share_of_step_time"])
    except Exception as e: print(n, "failed", e); print(open(f"gpurun_out/r2_bench_mxfp8_{n}.err").read()[-800:])
PY
bash tools/prof.sh r2_prof_mx --dtype mxfp8 > gpurun_out/r2_prof_mx_summary.txt 2>&1; cd $GRAFT_REPO_ROOT; head -14 gpurun_out/r2_prof_mx_summary.txt
