#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q --timeout 1500 2>&1 | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --dtype mxfp8 > gpurun_out/full_bench_mxfp8.json 2> gpurun_out/full_bench_mxfp8.err
python bench.py --dtype mxfp8 --size 256 --sampling-steps 250 --batch 16 --no-cpu-baseline > gpurun_out/full_bench256_mxfp8.json 2> gpurun_out/full_bench256_mxfp8.err
python bench.py --no-cpu-baseline > gpurun_out/full_bench_bf16.json 2> gpurun_out/full_bench_bf16.err
bash tools/prof.sh full_mx128 --dtype mxfp8 > gpurun_out/full_mx128_summary.txt 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json
for f in ("full_bench_mxfp8", "full_bench256_mxfp8", "full_bench_bf16"):
    r = json.load(open(f"gpurun_out/{f}.json"))
    print(f, "pairs/s", round(r["value"], 3), "ms/step", round(r["ms_per_step"], 1), "conv TF/s", round(r["roofline"]["achieved"], 1), "frac", round(r["roofline"]["frac"], 3), "e2e", round(r.get("e2e_files", {}).get("value", 0), 3), "drift", r.get("bf16_drift"))
PY
