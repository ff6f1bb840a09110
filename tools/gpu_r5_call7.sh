#!/bin/bash
# round 5, call 7: one-sweep la_ctx_split — tests, chain spread, per-launch A/B, parity-mode legs
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests/test_gpu_f16x3.py -m gpu -q -x -rA > $O/r5c7_tests.log 2>&1; echo rc=$? >> $O/r5c7_tests.log
grep -E "passed|failed|rc=|one sweep|one_sweep|two_sweeps|f16x3 \(B=" $O/r5c7_tests.log | tail -16
cd /tmp && export TMPDIR=/tmp
for V in 1 0; do
  PRG_SPLIT_LA_ONLINE=$V rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/r5c7_la$V -o r -- python $GRAFT_REPO_ROOT/bench.py --dtype f16x3 --timesteps 10 --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-roofline --no-e2e-files --no-drift --no-configs4 --no-parity-mode > $GRAFT_REPO_ROOT/$O/r5c7_la$V.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/prof_seq.py $O/r5c7_la0/r_kernel_trace.csv $O/r5c7_la1/r_kernel_trace.csv la_ > $O/r5c7_la_two_vs_one_sweep.txt 2>&1; cat $O/r5c7_la_two_vs_one_sweep.txt
rm -rf $O/r5c7_la0 $O/r5c7_la1
ARGS="--steps 1 --warmup 1 --no-roofline --no-cpu-baseline --no-e2e-files --no-drift --no-configs4"
for R in 1 2; do
python bench.py $ARGS > $O/r5c7_pm_one_$R.json 2> $O/r5c7_pm_one_$R.err
PRG_SPLIT_LA_ONLINE=0 python bench.py $ARGS > $O/r5c7_pm_two_$R.json 2> $O/r5c7_pm_two_$R.err
done
python - <<'PY'
import json
for r in (1, 2):
  for k in ("one", "two"):
    try:
        j = json.load(open(f"gpurun_out/r5c7_pm_{k}_{r}.json")); pm = j["parity_mode"]; f = pm["f16x3"]; g = pm["f16x3_256_ddim250"]
        print(r, k, "sweep(s): f16x3", round(f["pairs_per_s"], 3), "one lane", round(f["one_lane"]["pairs_per_s"], 3), "256:", round(g["pairs_per_s"], 3), "headline", round(j["value"], 3))
    except Exception as e:
        print(k, "failed", e)
PY
