#!/bin/bash
# tools/gpu_c64_exp.sh <tag>: timing experiments of conv3x3_c64_kernel (variant libraries built with -DPRG_C64_EXP=n), bf16 micro-bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1_c64_exp.txt; : > $O
for V in "" libprg_c64exp512.so "" libprg_c64exp512.so libprg_c64exp1.so; do
  echo "== ${V:-product}" >> $O
  if [ -n "$V" ]; then bash tools/gpu_split_bench.sh bf16 pointreggpt_amd/$V 2>/dev/null | grep -E "L0 |L1 " >> $O; else bash tools/gpu_split_bench.sh bf16 2>/dev/null | grep -E "L0 |L1 " >> $O; fi
done
cat $O
