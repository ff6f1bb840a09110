#!/bin/bash
cd $GRAFT_REPO_ROOT
python bench.py --steps 1 --warmup 1 > gpurun_out/r2_bench6.json 2> gpurun_out/r2_bench6.err
bash tools/prof.sh r2_prof128d > gpurun_out/r2_prof128d_summary.txt 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json
r = json.load(open("gpurun_out/r2_bench6.json"))
print("pairs/s", r["value"], "ms/step", r["ms_per_step"], "conv TF/s", r["roofline"]["achieved"], "share", r["roofline"]["share_of_step_time"])
print("e2e", r["e2e_files"]["value"], r["e2e_files"]["vs_device_only"])
for k, v in r["roofline_mem"]["kernels"].items(): print("  %-70s %6.1f us %7.0f GB/s" % (k, v["avg_us"], v["GBps"]))
print("drift", r["bf16_drift"]["inpainted_depth_m"], r["bf16_drift"]["xyz_m_points_kept_by_both"], r["bf16_drift"].get("saturated_fraction_fp32"))
PY
head -32 gpurun_out/r2_prof128d_summary.txt
