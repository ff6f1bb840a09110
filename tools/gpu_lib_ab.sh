#!/bin/bash
# tools/gpu_lib_ab.sh <tag> <variant lib name> <rounds> [pytest -k expression]: same-box A/B of the product library against
# pointreggpt_amd/libprg_<variant>.so (tools/variant_lib.sh): bf16 conv micro-bench (tools/split_bench.py), then the bf16 pipeline
# (bench.py, 200 transitions), arms alternating; first the parity tests named by the -k expression on the product library.
cd $GRAFT_REPO_ROOT
T=$1; V=$2; N=${3:-2}; K=$4
O=gpurun_out/${T}_ab_$V.txt; : > $O
if [ -n "$K" ]; then timeout 1500 python -m pytest tests -m gpu -q -x -k "$K" 2>&1 | tail -3 | tee -a $O; fi
for i in $(seq 1 $N); do
  for ARM in product $V; do
    echo "== $ARM (round $i)" >> $O
    if [ $ARM = product ]; then unset PRG_HIP_LIB; L=""; else export PRG_HIP_LIB=$GRAFT_REPO_ROOT/pointreggpt_amd/libprg_$V.so; L=pointreggpt_amd/libprg_$V.so; fi
    bash tools/gpu_split_bench.sh bf16 $L 2>/dev/null | grep -E "^L[0-9]" >> $O
    R=$(python bench.py --timesteps 200 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-e2e-files --no-drift --no-configs4 --no-parity-mode 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('%.3f pairs/s at 200 transitions' % j['value'])")
    echo "bench bf16 $ARM: $R" >> $O
  done
done
unset PRG_HIP_LIB
cat $O
