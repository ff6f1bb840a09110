#!/bin/bash
# round 5, call 3: (1) MX-activations-in-memory cap experiment (tools/gpu_r5_mxcap.sh), (2) f16x3 launch-by-launch listing,
# (3) f16x3 parity-mode leg on two lanes, with and without the sub-pixel Upsample form
cd $GRAFT_REPO_ROOT
O=gpurun_out
bash tools/gpu_r5_mxcap.sh > $O/r5_mxcap_summary.txt 2>&1
cat $O/r5_mxcap_summary.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/r5c3_f16x3 -o r -- python $GRAFT_REPO_ROOT/bench.py --dtype f16x3 --timesteps 10 --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-roofline --no-e2e-files --no-drift --no-configs4 --no-parity-mode > $GRAFT_REPO_ROOT/$O/r5c3_f16x3.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_seq.py $O/r5c3_f16x3/r_kernel_trace.csv _ > $O/r5c3_f16x3_all_per_launch.txt 2>&1
rm -rf $O/r5c3_f16x3
ARGS="--steps 1 --warmup 1 --no-roofline --no-cpu-baseline --no-e2e-files --no-drift --no-configs4"
python bench.py $ARGS > $O/r5c3_pm_default.json 2> $O/r5c3_pm_default.err
PRG_SPLIT_UP2X2=1 python bench.py $ARGS > $O/r5c3_pm_up2x2.json 2> $O/r5c3_pm_up2x2.err
python - <<'PY'
import json
for k in ("default", "up2x2"):
    try:
        j = json.load(open(f"gpurun_out/r5c3_pm_{k}.json"))
        pm = j["parity_mode"]
        f = pm["f16x3"]; g = pm["f16x3_256_ddim250"]
        print(k, "headline", round(j["value"], 3), "fp32", round(pm["fp32"]["pairs_per_s"], 3), "f16x3", round(f["pairs_per_s"], 3), "lanes", f["streams"],
              "one lane", f.get("one_lane"), "256:", round(g["pairs_per_s"], 3))
    except Exception as e:
        print(k, "failed", e)
PY
grep -c . $O/r5c3_f16x3_all_per_launch.txt
