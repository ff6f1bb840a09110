#!/bin/bash
# round 5, call 10: split8 with v_fma_mix_f32 — f16x3 tests, micro-bench, parity legs
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests/test_gpu_f16x3.py -m gpu -q -x -rA > $O/r5c10_tests.log 2>&1; echo rc=$? >> $O/r5c10_tests.log; tail -3 $O/r5c10_tests.log
grep -E "f16x3 \(B=" $O/r5c10_tests.log | cut -c1-120
bash tools/gpu_split_bench.sh f16x3 2>&1 | grep -E "^L0|^L1|^L2|^L3|^mid|^up1|^qkv|^res|^down" | tee $O/r5c10_split_bench.txt
ARGS="--steps 1 --warmup 1 --no-roofline --no-cpu-baseline --no-e2e-files --no-drift --no-configs4"
for R in 1 2; do
python bench.py $ARGS > $O/r5c10_pm_$R.json 2> $O/r5c10_pm_$R.err
done
python - <<'PY'
import json
for r in (1, 2):
    try:
        j = json.load(open(f"gpurun_out/r5c10_pm_{r}.json")); pm = j["parity_mode"]; f = pm["f16x3"]; g = pm["f16x3_256_ddim250"]
        print(r, "headline", round(j["value"], 3), "f16x3", round(f["pairs_per_s"], 3), "one lane", round(f["one_lane"]["pairs_per_s"], 3), "256:", round(g["pairs_per_s"], 3))
    except Exception as e:
        print("failed", e)
PY
