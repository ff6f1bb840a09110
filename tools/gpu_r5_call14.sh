#!/bin/bash
# round 5, call 14: bf16 c64 / w256 accumulators starting at the bias — full GPU suite, per-launch A/B against the previous build
# (pointreggpt_amd/libprg_old.so), alternating headline runs
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests -m gpu -q -x > $O/r5c14_tests.log 2>&1; echo rc=$? >> $O/r5c14_tests.log; tail -3 $O/r5c14_tests.log
OLD=$GRAFT_REPO_ROOT/pointreggpt_amd/libprg_old.so
bash tools/prof.sh r5c14_new --streams 1 --no-parity-mode > $O/r5c14_new_summary.txt 2>&1
cd $GRAFT_REPO_ROOT
PRG_HIP_LIB=$OLD bash tools/prof.sh r5c14_old --streams 1 --no-parity-mode > $O/r5c14_old_summary.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_seq.py $O/r5c14_old/r_kernel_trace.csv $O/r5c14_new/r_kernel_trace.csv conv > $O/r5c14_conv_old_vs_new.txt 2>&1; cat $O/r5c14_conv_old_vs_new.txt
head -2 $O/r5c14_old_summary.txt; head -2 $O/r5c14_new_summary.txt
rm -rf $O/r5c14_old $O/r5c14_new
ARGS="--steps 6 --warmup 2 --no-roofline --no-cpu-baseline --no-e2e-files --no-drift --no-configs4 --no-parity-mode"
for R in 1 2 3; do
python bench.py $ARGS > $O/r5c14_b_new_$R.json 2> $O/r5c14_b_new_$R.err
PRG_HIP_LIB=$OLD python bench.py $ARGS > $O/r5c14_b_old_$R.json 2> $O/r5c14_b_old_$R.err
done
python - <<'PY'
import json
for k in ("new", "old"):
    v = []
    for r in (1, 2, 3):
        try: v.append(round(json.load(open(f"gpurun_out/r5c14_b_{k}_{r}.json"))["value"], 3))
        except Exception as e: v.append(str(e))
    print(k, v)
PY
