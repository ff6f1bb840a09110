#!/bin/bash
# tools/gpu_c64w_ab.sh <tag> [rounds]: same-box A/B of conv3x3_c64w_kernel (PRG_CONV_C64W=1; default 0) against conv3x3_c64_kernel (=0): bf16
# micro-bench shapes, the kernel's tests, then the whole bf16 pipeline (bench.py, 200 transitions)
cd $GRAFT_REPO_ROOT
T=$1; N=${2:-2}
O=gpurun_out/${T}_c64w_ab.txt; : > $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "c64w or block_pair_h16_against or kernels_one_at_a_time" 2>&1 | tail -4 | tee -a $O
for i in $(seq 1 $N); do
  for ARM in 0 1; do
    echo "== PRG_CONV_C64W=$ARM (round $i)" >> $O
    PRG_CONV_C64W=$ARM bash tools/gpu_split_bench.sh bf16 2>/dev/null | grep -E "L0 64|L1 64" >> $O
    R=$(PRG_CONV_C64W=$ARM python bench.py --timesteps 200 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-e2e-files --no-drift --no-configs4 --no-parity-mode 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('%.3f pairs/s at 200 transitions' % j['value'])")
    echo "bench bf16 PRG_CONV_C64W=$ARM: $R" >> $O
  done
done
cat $O
