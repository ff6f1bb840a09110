#!/bin/bash
# tools/gpu_c64_exp.sh <tag> [variant libs...]: timing experiments of conv3x3_c64_kernel (variant libraries: tools/variant_lib.sh <name> conv_c64.hip
# -DPRG_C64_EXP=<n>), bf16 micro-bench, two rounds on one box.  Default variants: c64old (1024), c64exp1536 (1024 + 512), c64exp3072 (1024 + 2048)
cd $GRAFT_REPO_ROOT
T=$1; shift
VS=${@:-"c64old c64exp1536 c64exp3072"}
O=gpurun_out/${T}_c64_exp.txt; : > $O
for i in 1 2; do
  for V in product $VS; do
    echo "== $V (round $i)" >> $O
    if [ $V = product ]; then bash tools/gpu_split_bench.sh bf16 2>/dev/null | grep -E "L0 |L1 64" >> $O; else bash tools/gpu_split_bench.sh bf16 pointreggpt_amd/libprg_$V.so 2>/dev/null | grep -E "L0 |L1 64" >> $O; fi
  done
done
cat $O
