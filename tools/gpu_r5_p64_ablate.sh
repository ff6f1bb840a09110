#!/bin/bash
# round 5: timing ablations of conv3x3_split_p64_kernel (libprg_exp{11..15}.so = conv_split.hip built -DPRG_SPLIT_EXP=n: 11 no MFMAs,
# 12 no epilogue stores, 13 no halo staging (loads, prologue, split, LDS writes), 14 no fragment reads, 15 no weight DMA); results garbage
cd $GRAFT_REPO_ROOT
O=gpurun_out
: > $O/r5_p64_ablate.txt
for n in 0 11 12 13 14 15; do
  if [ $n = 0 ]; then LIB=pointreggpt_amd/libprg_hip.so; else LIB=pointreggpt_amd/libprg_exp$n.so; fi
  echo "== PRG_SPLIT_EXP=$n" >> $O/r5_p64_ablate.txt
  bash tools/gpu_split_bench.sh f16x3 $LIB 2>&1 | grep -E "^L0|^L1" >> $O/r5_p64_ablate.txt
done
cat $O/r5_p64_ablate.txt
