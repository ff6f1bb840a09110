#!/bin/bash
# tools/gpu_c64w_exp.sh <tag>: timing experiments of conv3x3_c64w_kernel (variant libraries built with -DPRG_C64W_EXP=n), bf16 micro-bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1_c64w_exp.txt; : > $O
for V in "" libprg_c64wexp1.so libprg_c64wexp2.so libprg_c64wexp4.so libprg_c64wexp6.so; do
  echo "== ${V:-product}" >> $O
  if [ -n "$V" ]; then bash tools/gpu_split_bench.sh bf16 pointreggpt_amd/$V 2>/dev/null | grep -E "L0 64|L1 64" >> $O; else bash tools/gpu_split_bench.sh bf16 2>/dev/null | grep -E "L0 64|L1 64" >> $O; fi
done
cat $O
