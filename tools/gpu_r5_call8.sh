#!/bin/bash
# round 5, call 8: leaner epilogues (split p64 / ws addressing; shared epilogue: shift instead of division, cached tables) —
# tests, then same-box A/B against the previous build (pointreggpt_amd/libprg_old.so): per-launch listings and pairs/s
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests -m gpu -q -x > $O/r5c8_tests.log 2>&1; echo rc=$? >> $O/r5c8_tests.log; tail -3 $O/r5c8_tests.log
OLD=$GRAFT_REPO_ROOT/pointreggpt_amd/libprg_old.so
bash tools/prof.sh r5c8_bf16_new --streams 1 --no-parity-mode > $O/r5c8_bf16_new_summary.txt 2>&1
cd $GRAFT_REPO_ROOT
PRG_HIP_LIB=$OLD bash tools/prof.sh r5c8_bf16_old --streams 1 --no-parity-mode > $O/r5c8_bf16_old_summary.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_seq.py $O/r5c8_bf16_old/r_kernel_trace.csv $O/r5c8_bf16_new/r_kernel_trace.csv igemm > $O/r5c8_bf16_igemm_old_vs_new.txt 2>&1; cat $O/r5c8_bf16_igemm_old_vs_new.txt
head -2 $O/r5c8_bf16_old_summary.txt; head -2 $O/r5c8_bf16_new_summary.txt
rm -rf $O/r5c8_bf16_old $O/r5c8_bf16_new
cd /tmp && export TMPDIR=/tmp
for V in new old; do
  [ $V = old ] && export PRG_HIP_LIB=$OLD
  rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/r5c8_f16x3_$V -o r -- python $GRAFT_REPO_ROOT/bench.py --dtype f16x3 --timesteps 10 --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-roofline --no-e2e-files --no-drift --no-configs4 --no-parity-mode > $GRAFT_REPO_ROOT/$O/r5c8_f16x3_$V.log 2>&1
done
unset PRG_HIP_LIB
cd $GRAFT_REPO_ROOT
python tools/prof_seq.py $O/r5c8_f16x3_old/r_kernel_trace.csv $O/r5c8_f16x3_new/r_kernel_trace.csv conv > $O/r5c8_f16x3_conv_old_vs_new.txt 2>&1; cat $O/r5c8_f16x3_conv_old_vs_new.txt
rm -rf $O/r5c8_f16x3_old $O/r5c8_f16x3_new
ARGS="--steps 2 --warmup 1 --no-roofline --no-cpu-baseline --no-e2e-files --no-drift --no-configs4"
for R in 1 2; do
python bench.py $ARGS > $O/r5c8_pm_new_$R.json 2> $O/r5c8_pm_new_$R.err
PRG_HIP_LIB=$OLD python bench.py $ARGS > $O/r5c8_pm_old_$R.json 2> $O/r5c8_pm_old_$R.err
done
python - <<'PY'
import json
for r in (1, 2):
  for k in ("new", "old"):
    try:
        j = json.load(open(f"gpurun_out/r5c8_pm_{k}_{r}.json")); pm = j["parity_mode"]; f = pm["f16x3"]; g = pm["f16x3_256_ddim250"]
        print(r, k, "headline", round(j["value"], 3), "f16x3", round(f["pairs_per_s"], 3), "one lane", round(f["one_lane"]["pairs_per_s"], 3), "256:", round(g["pairs_per_s"], 3))
    except Exception as e:
        print(k, "failed", e)
PY
