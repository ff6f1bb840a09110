#!/bin/bash
# round 5, call 9: merged producers of the persistent 64-channel split kernel — tests, micro-bench A/B, parity legs A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests/test_gpu_f16x3.py -m gpu -q -x > $O/r5c9_tests.log 2>&1; echo rc=$? >> $O/r5c9_tests.log; tail -3 $O/r5c9_tests.log
for V in 1 0; do
  echo "== PRG_SPLIT_P64_MERGED=$V"; PRG_SPLIT_P64_MERGED=$V bash tools/gpu_split_bench.sh f16x3 2>&1 | grep -E "^L0|^L1"
done | tee $O/r5c9_p64_merged_bench.txt
ARGS="--steps 1 --warmup 1 --no-roofline --no-cpu-baseline --no-e2e-files --no-drift --no-configs4"
for R in 1 2; do
python bench.py $ARGS > $O/r5c9_pm_new_$R.json 2> $O/r5c9_pm_new_$R.err
PRG_SPLIT_P64_MERGED=0 python bench.py $ARGS > $O/r5c9_pm_old_$R.json 2> $O/r5c9_pm_old_$R.err
done
python - <<'PY'
import json
for r in (1, 2):
  for k in ("new", "old"):
    try:
        j = json.load(open(f"gpurun_out/r5c9_pm_{k}_{r}.json")); pm = j["parity_mode"]; f = pm["f16x3"]; g = pm["f16x3_256_ddim250"]
        print(r, k, "headline", round(j["value"], 3), "f16x3", round(f["pairs_per_s"], 3), "one lane", round(f["one_lane"]["pairs_per_s"], 3), "256:", round(g["pairs_per_s"], 3))
    except Exception as e:
        print(k, "failed", e)
PY
