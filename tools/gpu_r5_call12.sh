#!/bin/bash
# round 5, call 12: attn_split.hip split1 with / without v_fma_mix_f32 (libprg_nomix.so = attn_split.hip built -DPRG_SPLIT_MIX=0), same box
cd $GRAFT_REPO_ROOT
O=gpurun_out
OLD=pointreggpt_amd/libprg_nomix.so
python -m pytest tests/test_gpu_f16x3.py -m gpu -q -x > $O/r5c12_tests.log 2>&1; echo rc=$? >> $O/r5c12_tests.log; tail -3 $O/r5c12_tests.log
cd /tmp && export TMPDIR=/tmp
for V in new old; do
  [ $V = old ] && export PRG_HIP_LIB=$GRAFT_REPO_ROOT/$OLD
  rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/r5c12_f16x3_$V -o r -- python $GRAFT_REPO_ROOT/bench.py --dtype f16x3 --timesteps 10 --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-roofline --no-e2e-files --no-drift --no-configs4 --no-parity-mode > $GRAFT_REPO_ROOT/$O/r5c12_f16x3_$V.log 2>&1
done
unset PRG_HIP_LIB
cd $GRAFT_REPO_ROOT
python tools/prof_seq.py $O/r5c12_f16x3_old/r_kernel_trace.csv $O/r5c12_f16x3_new/r_kernel_trace.csv _split > $O/r5c12_f16x3_attn_nomix_vs_mix.txt 2>&1; grep -E "la_|full_attn|sum" $O/r5c12_f16x3_attn_nomix_vs_mix.txt
rm -rf $O/r5c12_f16x3_old $O/r5c12_f16x3_new
ARGS="--steps 1 --warmup 1 --no-roofline --no-cpu-baseline --no-e2e-files --no-drift --no-configs4"
for R in 1 2; do
python bench.py $ARGS > $O/r5c12_pm_new_$R.json 2> $O/r5c12_pm_new_$R.err
PRG_HIP_LIB=$GRAFT_REPO_ROOT/$OLD python bench.py $ARGS > $O/r5c12_pm_old_$R.json 2> $O/r5c12_pm_old_$R.err
done
python - <<'PY'
import json
for r in (1, 2):
  for k in ("new", "old"):
    try:
        j = json.load(open(f"gpurun_out/r5c12_pm_{k}_{r}.json")); pm = j["parity_mode"]; f = pm["f16x3"]; g = pm["f16x3_256_ddim250"]
        print(r, k, "headline", round(j["value"], 3), "f16x3", round(f["pairs_per_s"], 3), "one lane", round(f["one_lane"]["pairs_per_s"], 3), "256:", round(g["pairs_per_s"], 3))
    except Exception as e:
        print(k, "failed", e)
PY
