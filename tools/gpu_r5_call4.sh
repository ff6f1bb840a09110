#!/bin/bash
# round 5, call 4: f16x3 tests with the MFMA stem + split bottleneck attention, the MX cap experiment, parity-mode legs
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests/test_gpu_f16x3.py -m gpu -q -x -rA > $O/r5c4_tests_f16x3.log 2>&1; echo "pytest rc=$?" >> $O/r5c4_tests_f16x3.log
grep -E "passed|failed|rc=|f16x3 \(B=|mfma vs scalar|on the (mfma|scalar)" $O/r5c4_tests_f16x3.log | tail -30
bash tools/gpu_r5_mxcap.sh > $O/r5_mxcap_summary.txt 2>&1
cat $O/r5_mxcap_summary.txt
ARGS="--steps 1 --warmup 1 --no-roofline --no-cpu-baseline --no-e2e-files --no-drift --no-configs4"
python bench.py $ARGS > $O/r5c4_pm_default.json 2> $O/r5c4_pm_default.err
PRG_SPLIT_STEM=0 PRG_SPLIT_FULLATTN=0 python bench.py $ARGS > $O/r5c4_pm_scalar.json 2> $O/r5c4_pm_scalar.err
python - <<'PY'
import json
for k in ("default", "scalar"):
    try:
        j = json.load(open(f"gpurun_out/r5c4_pm_{k}.json"))
        pm = j["parity_mode"]
        f = pm["f16x3"]; g = pm["f16x3_256_ddim250"]
        print(k, "headline", round(j["value"], 3), "fp32", round(pm["fp32"]["pairs_per_s"], 3), "f16x3", round(f["pairs_per_s"], 3), "lanes", f["streams"],
              "one lane", f.get("one_lane"), "256:", round(g["pairs_per_s"], 3))
    except Exception as e:
        print(k, "failed", e)
PY
