#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "mx" -rA > $O/r5c6_tests_mx.log 2>&1; echo rc=$? >> $O/r5c6_tests_mx.log; grep -E "passed|failed|rc=" $O/r5c6_tests_mx.log | tail -5
bash tools/prof.sh r5c6_mx --streams 1 --no-parity-mode --size 256 --batch 16 --dtype mxfp8 > $O/r5c6_mx_summary.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_seq.py $O/r5c6_mx/r_kernel_trace.csv conv > $O/r5c6_mx_conv.txt 2>&1; tail -30 $O/r5c6_mx_conv.txt; head -3 $O/r5c6_mx_summary.txt
rm -rf $O/r5c6_mx
ARGS="--steps 1 --warmup 1 --no-roofline --no-cpu-baseline --no-e2e-files --no-drift --no-parity-mode --c4-steps 4"
for R in 1 2; do
  python bench.py $ARGS > $O/r5c6_new_$R.json 2> $O/r5c6_new_$R.err
  PRG_MX_UP=1 PRG_W256_MIN_TILES=128 python bench.py $ARGS > $O/r5c6_old_$R.json 2> $O/r5c6_old_$R.err
done
python - <<'PY'
import json
for r in (1, 2):
    for k in ("new", "old"):
        try:
            j = json.load(open(f"gpurun_out/r5c6_{k}_{r}.json")); c = j["configs4"]
            print(f"run {r} {k}: configs4 mxfp8 {c['value']:.3f} pairs/s, bf16 same shape {c['bf16_same_shape']['value']:.3f}, ratio {c['mxfp8_over_bf16_same_shape']:.3f}; headline {j['value']:.3f}")
        except Exception as e:
            print(r, k, "failed", e)
PY
