#!/bin/bash
# tools/gpu_f16x3_env_ab.sh <tag> <ENV=VAL of arm B> [rounds]: same-box A/B of the whole f16x3 pipeline (bench.py --dtype f16x3, 100 ancestral
# transitions, B = 64, 128x128, two lanes) with an environment switch off / on, alternating.
cd $GRAFT_REPO_ROOT
T=$1; KV=$2; N=${3:-2}
K=${KV%%=*}; V=${KV#*=}
O=gpurun_out/${T}_f16x3_env_ab.txt
: > $O
ARGS="--dtype f16x3 --timesteps 100 --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-e2e-files --no-drift --no-configs4 --no-parity-mode"
for i in $(seq 1 $N); do
  for ARM in 0 $V; do
    R=$(env $K=$ARM python bench.py $ARGS 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('%.4f pairs/s at 100 transitions, %.1f ms per step' % (j['value'], j['ms_per_step']))")
    echo "round $i  $K=$ARM  $R" | tee -a $O
  done
done
