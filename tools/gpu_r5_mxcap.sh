#!/bin/bash
# round 5: upper bound of "MX activations in memory" for configs[4].  libprg_mxcap.so = this tree with
# conv_w256.hip compiled -DPRG_W256_EXP=64: the plain (no-prologue) MX launches gather HALF the bytes and do NO quantisation
# arithmetic (meaningless numbers, finite).  Same box, alternating runs, the configs[4] leg of bench.py (B = 16, 256x256, 250-step
# DDIM) with its bf16 same-shape comparator.
cd $GRAFT_REPO_ROOT
O=gpurun_out
ARGS="--steps 1 --warmup 1 --no-roofline --no-cpu-baseline --no-e2e-files --no-drift --no-parity-mode --c4-steps 4"
for R in 1 2; do
  python bench.py $ARGS > $O/r5_mxcap_real_$R.json 2> $O/r5_mxcap_real_$R.err
  PRG_HIP_LIB=$GRAFT_REPO_ROOT/pointreggpt_amd/libprg_mxcap.so python bench.py $ARGS > $O/r5_mxcap_cap_$R.json 2> $O/r5_mxcap_cap_$R.err
done
python - <<'PY'
import json
for r in (1, 2):
    for k in ("real", "cap"):
        try:
            j = json.load(open(f"gpurun_out/r5_mxcap_{k}_{r}.json"))
            c = j["configs4"]
            print(f"run {r} {k:4s}: configs4 mxfp8 {c['value']:.3f} pairs/s ({c['ms_per_step']:.1f} ms/batch), bf16 same shape {c['bf16_same_shape']['value']:.3f}, "
                  f"ratio {c['mxfp8_over_bf16_same_shape']:.3f}; headline {j['value']:.3f}")
        except Exception as e:
            print(r, k, "failed", e)
PY
