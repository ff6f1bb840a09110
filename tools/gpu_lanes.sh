#!/bin/bash
# tools/gpu_lanes.sh <tag> [dtype] [timesteps]: pairs/s of the driver's command shape (--steps 20 --warmup 5) with 2 lanes dealt round-robin
# (the default), and 2 / 3 / 4 lanes pulling batches from a shared counter (--dynamic-lanes); same box, back to back
cd $GRAFT_REPO_ROOT
DT=${2:-bf16}; TS=${3:-1000}
O=gpurun_out/$1_lanes_$DT.txt; : > $O
A="--dtype $DT --timesteps $TS --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-e2e-files --no-drift --no-configs4 --no-parity-mode"
for CFG in "2" "2 --dynamic-lanes" "3 --dynamic-lanes" "4 --dynamic-lanes" "3 --dynamic-lanes" "2 --dynamic-lanes"; do
  python bench.py $A --streams $CFG > $O.out 2> $O.err
  R=$(tail -1 $O.out | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('%.3f pairs/s, %.1f ms per step' % (j['value'], j['ms_per_step']))" 2>/dev/null || (grep -v '^{' $O.err | tail -3))
  echo "$DT lanes $CFG :  $R" | tee -a $O
done
rm -f $O.out $O.err
