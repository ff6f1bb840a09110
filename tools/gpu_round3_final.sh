#!/bin/bash
# round-end evidence run (GPU box, through gpurun): full GPU suite, smoke, the driver's bench command, rocprofv3 kernel
# traces with one and two lanes, the three PMC passes.  Everything lands in gpurun_out/r3_final_*; copy into profiles/.
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -rA > gpurun_out/r3_final_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_final_tests.log; tail -3 gpurun_out/r3_final_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/r3_final_smoke.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r3_final_bench.json 2> gpurun_out/r3_final_bench.err
bash tools/prof.sh r3_final_prof_1lane --streams 1 > gpurun_out/r3_final_prof_1lane_summary.txt 2>&1
cd $GRAFT_REPO_ROOT
bash tools/prof.sh r3_final_prof_2lanes --streams 2 --steps 2 > gpurun_out/r3_final_prof_2lanes_summary.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_seq.py gpurun_out/r3_final_prof_1lane/r_kernel_trace.csv conv > gpurun_out/r3_final_conv_per_launch.txt 2>&1
bash tools/pmc.sh r3_final_pmc > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json
r = json.load(open("gpurun_out/r3_final_bench.json"))
print("pairs/s", round(r["value"], 3), "ms/step", round(r["ms_per_step"], 1), "conv TF/s", round(r["roofline"]["achieved"], 1), "frac", round(r["roofline"]["frac"], 3),
      "share", round(r["roofline"]["share_of_step_time"], 3), "traffic", r["roofline"]["traffic"], "e2e", round(r.get("e2e_files", {}).get("value", 0), 3),
      "cpu", r.get("cpu_baseline", {}).get("value"), "configs4", round(r.get("configs4", {}).get("value", 0), 3))
PY
head -8 gpurun_out/r3_final_prof_1lane_summary.txt
