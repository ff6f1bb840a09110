#!/bin/bash
cd $GRAFT_REPO_ROOT
for b in 64 32 16 128; do
  timeout 900 python bench.py --batch $b --steps 2 --warmup 1 --no-cpu-baseline --no-e2e-files --no-drift > gpurun_out/batch_$b.json 2> gpurun_out/batch_$b.err
  python - <<PY
import json
r = json.load(open("gpurun_out/batch_$b.json"))
print("B=$b pairs/s", round(r["value"], 3), "ms/step", round(r["ms_per_step"], 1), "conv TF/s", round(r["roofline"]["achieved"], 1), "share", round(r["roofline"]["share_of_step_time"], 3))
PY
done
