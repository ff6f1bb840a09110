#!/bin/bash
# round 5: timing ablations of conv3x3_split_ws_kernel (libprg_exp{21,23,24,25}.so = conv_split.hip built -DPRG_SPLIT_EXP=n: 21 no MFMAs,
# 23 no halo staging, 24 no fragment reads, 25 no weight DMA); results garbage
cd $GRAFT_REPO_ROOT
O=gpurun_out
: > $O/r5_ws_ablate.txt
for n in 0 21 23 24 25; do
  if [ $n = 0 ]; then LIB=pointreggpt_amd/libprg_hip.so; else LIB=pointreggpt_amd/libprg_exp$n.so; fi
  echo "== PRG_SPLIT_EXP=$n" >> $O/r5_ws_ablate.txt
  bash tools/gpu_split_bench.sh f16x3 $LIB 2>&1 | grep -E "^L2|^L3|^mid|^up1" >> $O/r5_ws_ablate.txt
done
cat $O/r5_ws_ablate.txt
