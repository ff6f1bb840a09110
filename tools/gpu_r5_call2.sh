#!/bin/bash
# round 5, call 2: full GPU suite on the new binary, then same-box A/B of the 1x1 conv changes (swizzled LDS rows, 256x128 tiles)
# and of the MaskUnet stem against round 4's library (pointreggpt_amd/libprg_old.so).
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests -m gpu -q -x -rA > $O/r5c2_tests.log 2>&1; echo "pytest rc=$?" >> $O/r5c2_tests.log; tail -5 $O/r5c2_tests.log
grep -E "MaskUnet stem|small-weight|8 ranks|f16x3 \(B=" $O/r5c2_tests.log | head -20
for V in 0 1 2; do
  PRG_IGEMM_BM256=$V bash tools/prof.sh r5c2_ig$V --streams 1 --no-parity-mode > $O/r5c2_ig${V}_summary.txt 2>&1
  cd $GRAFT_REPO_ROOT
done
PRG_HIP_LIB_ALLOW_MISSING=1 PRG_HIP_LIB=$GRAFT_REPO_ROOT/pointreggpt_amd/libprg_old.so bash tools/prof.sh r5c2_old --streams 1 --no-parity-mode > $O/r5c2_old_summary.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_seq.py $O/r5c2_old/r_kernel_trace.csv $O/r5c2_ig0/r_kernel_trace.csv igemm > $O/r5c2_seq_old_vs_ig0.txt 2>&1
python tools/prof_seq.py $O/r5c2_ig0/r_kernel_trace.csv $O/r5c2_ig1/r_kernel_trace.csv igemm > $O/r5c2_seq_ig0_vs_ig1.txt 2>&1
python tools/prof_seq.py $O/r5c2_ig0/r_kernel_trace.csv $O/r5c2_ig2/r_kernel_trace.csv igemm > $O/r5c2_seq_ig0_vs_ig2.txt 2>&1
cat $O/r5c2_seq_old_vs_ig0.txt; tail -12 $O/r5c2_seq_ig0_vs_ig1.txt; tail -3 $O/r5c2_seq_ig0_vs_ig2.txt
for T in old ig0 ig1 ig2; do head -3 $O/r5c2_${T}_summary.txt; grep -E "stem" $O/r5c2_${T}_summary.txt | head -3; done
rm -rf $O/r5c2_old $O/r5c2_ig0 $O/r5c2_ig1 $O/r5c2_ig2
