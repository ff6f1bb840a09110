#!/bin/bash
# round-end evidence run: full GPU suite, smoke, default bench, 256 bench, rocprof stats, PMC passes
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q --timeout 1500 > gpurun_out/r2_final_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_final_tests.log; tail -3 gpurun_out/r2_final_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/r2_final_bench.json 2> gpurun_out/r2_final_bench.err
python bench.py --size 256 --sampling-steps 250 --batch 16 --no-cpu-baseline > gpurun_out/r2_final_bench256.json 2> gpurun_out/r2_final_bench256.err
bash tools/prof.sh r2_final_prof128 > gpurun_out/r2_final_prof128_summary.txt 2>&1
cd $GRAFT_REPO_ROOT
bash tools/prof.sh r2_final_prof256 --size 256 --batch 16 > gpurun_out/r2_final_prof256_summary.txt 2>&1
cd $GRAFT_REPO_ROOT
bash tools/pmc.sh r2_final_pmc > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json
for f in ("r2_final_bench", "r2_final_bench256"):
    r = json.load(open(f"gpurun_out/{f}.json"))
    print(f, "pairs/s", round(r["value"], 3), "ms/step", round(r["ms_per_step"], 1), "conv TF/s", round(r["roofline"]["achieved"], 1), "frac", round(r["roofline"]["frac"], 3),
          "share", round(r["roofline"]["share_of_step_time"], 3), "traffic", r["roofline"]["traffic"], "e2e", round(r.get("e2e_files", {}).get("value", 0), 3), "cpu", r.get("cpu_baseline", {}).get("value"))
PY
head -8 gpurun_out/r2_final_prof128_summary.txt
