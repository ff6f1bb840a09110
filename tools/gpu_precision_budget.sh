#!/bin/bash
# round 5: which contraction class of the f16x3 mode needs more than 22 bits on the 256x256 chain (G21b), and what the
# per-channel power-of-two weight scale (PRG_SPLIT_WSCALE, conv_split.hip) buys.  Runs on the GPU box through gpurun.
cd $GRAFT_REPO_ROOT
O=gpurun_out/precision_budget.txt
: > $O
run() {  # fixture batch env...
  local fx=$1 nb=$2; shift 2
  env "$@" python tools/chain_run.py $fx f16x3 $nb 2>&1 | grep -E "^CHAIN|Error|error" >> $O
}
for FX in G21b_ddim250_256 G20_ddim250_128; do
  run $FX 1 PRG_SPLIT_WSCALE=0
  run $FX 1 PRG_SPLIT_WSCALE=1
  for M in 1 2 4 8 15; do
    run $FX 1 PRG_SPLIT_WSCALE=0 PRG_SPLIT_EXACT=$M
    run $FX 1 PRG_SPLIT_WSCALE=1 PRG_SPLIT_EXACT=$M
  done
  run $FX 1 PRG_SPLIT_WSCALE=1 PRG_SPLIT_ATTN=0
  run $FX 1 PRG_SPLIT_WSCALE=1 PRG_SPLIT_ATTN=0 PRG_SPLIT_EXACT=15
  run $FX 1 PRG_SPLIT_WSCALE=1 PRG_SPLIT_UP2X2=1
done
run G22_chain1000_ancestral_128 1 PRG_SPLIT_WSCALE=1
run G19_chain1000_ancestral_64 1 PRG_SPLIT_WSCALE=1
python - <<'PY'
import json
for l in open("gpurun_out/precision_budget.txt"):
    if l.startswith("CHAIN "):
        r = json.loads(l[6:])
        print(f"{r['fixture']:30s} {str(r['env']):70s} xyz {r.get('xyz_linf_m', float('nan')):.3e} mean {r['depth_mean_m']:.3e} same_mask {r['same_valid_mask']} {r['seconds']} s")
    else:
        print(l.rstrip())
PY
