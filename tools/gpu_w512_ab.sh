#!/bin/bash
# tools/gpu_w512_ab.sh <tag> [rounds]: same-box A/B of the one-wave-per-SIMD f16x3 kernel (PRG_SPLIT_W512=1) against the wave-specialised
# one on the micro-bench shapes (rocprofv3 kernel trace), then its bit-identity test.
cd $GRAFT_REPO_ROOT
T=$1; N=${2:-2}
O=gpurun_out/${T}_w512_ab.txt
: > $O
for i in $(seq 1 $N); do
  echo "== conv3x3_split_ws_kernel (round $i)" >> $O
  PRG_SPLIT_W512=0 bash tools/gpu_split_bench.sh f16x3 2>/dev/null | grep -E "L2 |L3 |mid |up1 " >> $O
  echo "== conv3x3_split_w512_kernel, PRG_SPLIT_W512=2 (round $i)" >> $O
  PRG_SPLIT_W512=2 bash tools/gpu_split_bench.sh f16x3 2>/dev/null | grep -E "L2 |L3 |mid |up1 " >> $O
done
cat $O
timeout 900 python -m pytest tests/test_gpu_f16x3.py -q -x -k one_wave_per_simd 2>&1 | tail -5 | tee -a $O
