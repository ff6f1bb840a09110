#!/bin/bash
# round 5: how much the point-XYZ L-infinity of the long chains moves between EQUAL-PRECISION variants of the f16x3 arithmetic
# (kernel choices that only re-order float32 sums): the spread the literal 1e-4 m assertion has to live with.
cd $GRAFT_REPO_ROOT
O=gpurun_out/chain_spread.txt
: > $O
run() {  # fixture batch env...
  local fx=$1 nb=$2; shift 2
  env "$@" python tools/chain_run.py $fx f16x3 $nb 2>&1 | grep -E "^CHAIN|Error|error" >> $O
}
for FX in G21b_ddim250_256 G20_ddim250_128 G22_chain1000_ancestral_128; do
  run $FX 1 PRG_X=0
  run $FX 1 PRG_SPLIT_UP2X2=1
  run $FX 1 PRG_SPLIT_STEM=0
  run $FX 1 PRG_SPLIT_FULLATTN=0
  run $FX 1 PRG_SPLIT_P64=0
  run $FX 1 PRG_SPLIT_WS=0
  run $FX 1 PRG_SPLIT_ATTN_C128=0
  run $FX 1 PRG_SPLIT_UP2X2=1 PRG_SPLIT_P64=0
  run $FX 1 PRG_SPLIT_UP2X2=1 PRG_SPLIT_STEM=0
done
python - <<'PY'
import json
for l in open("gpurun_out/chain_spread.txt"):
    if l.startswith("CHAIN "):
        r = json.loads(l[6:])
        print(f"{r['fixture']:30s} {str(r['env']):60s} xyz {r.get('xyz_linf_m', float('nan')):.3e} mean {r['depth_mean_m']:.3e} same_mask {r['same_valid_mask']} {r['seconds']} s")
    else:
        print(l.rstrip())
PY
