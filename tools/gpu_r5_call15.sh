#!/bin/bash
# round 5, call 15: split kernels — a chunk's first MFMAs start from the constant zero (no accumulator clearing in the flush):
# f16x3 tests, micro-bench A/B against the previous build (pointreggpt_amd/libprg_old.so), parity legs A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests/test_gpu_f16x3.py -m gpu -q -x > $O/r5c15_tests.log 2>&1; echo rc=$? >> $O/r5c15_tests.log; tail -2 $O/r5c15_tests.log
OLD=pointreggpt_amd/libprg_old.so
for R in 1 2; do
echo "== new (run $R)"; bash tools/gpu_split_bench.sh f16x3 2>&1 | grep -E "^L0|^L1|^L2|^L3|^mid|^up1"
echo "== old (run $R)"; bash tools/gpu_split_bench.sh f16x3 $OLD 2>&1 | grep -E "^L0|^L1|^L2|^L3|^mid|^up1"
done | cut -c1-75 | tee $O/r5c15_split_bench_ab.txt
ARGS="--steps 1 --warmup 1 --no-roofline --no-cpu-baseline --no-e2e-files --no-drift --no-configs4"
for R in 1 2; do
python bench.py $ARGS > $O/r5c15_pm_new_$R.json 2> $O/r5c15_pm_new_$R.err
PRG_HIP_LIB=$GRAFT_REPO_ROOT/$OLD python bench.py $ARGS > $O/r5c15_pm_old_$R.json 2> $O/r5c15_pm_old_$R.err
done
python - <<'PY'
import json
for r in (1, 2):
  for k in ("new", "old"):
    try:
        j = json.load(open(f"gpurun_out/r5c15_pm_{k}_{r}.json")); pm = j["parity_mode"]; f = pm["f16x3"]; g = pm["f16x3_256_ddim250"]
        print(r, k, "headline", round(j["value"], 3), "f16x3", round(f["pairs_per_s"], 3), "one lane", round(f["one_lane"]["pairs_per_s"], 3), "256:", round(g["pairs_per_s"], 3))
    except Exception as e:
        print(k, "failed", e)
PY
