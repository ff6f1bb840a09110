#!/bin/bash
# round 5: configs[4] shape (B = 16, 256 x 256), conv launches of one evaluation side by side: bf16 | mxfp8 (same box, same positions)
cd $GRAFT_REPO_ROOT
O=gpurun_out
bash tools/prof.sh r5_mxseq_bf16 --streams 1 --no-parity-mode --size 256 --batch 16 > $O/r5_mxseq_bf16_summary.txt 2>&1
cd $GRAFT_REPO_ROOT
bash tools/prof.sh r5_mxseq_mx --streams 1 --no-parity-mode --size 256 --batch 16 --dtype mxfp8 > $O/r5_mxseq_mx_summary.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_seq.py $O/r5_mxseq_bf16/r_kernel_trace.csv $O/r5_mxseq_mx/r_kernel_trace.csv conv > $O/r5_mxseq_conv_bf16_vs_mxfp8.txt 2>&1
cat $O/r5_mxseq_conv_bf16_vs_mxfp8.txt
head -4 $O/r5_mxseq_bf16_summary.txt; head -4 $O/r5_mxseq_mx_summary.txt
rm -rf $O/r5_mxseq_bf16 $O/r5_mxseq_mx
