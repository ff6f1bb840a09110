#!/bin/bash
# tools/gpu_split_ab.sh <tag> <variant-lib> [rounds]: same-box A/B of the f16x3 conv micro-bench (tools/split_bench.py under rocprofv3):
# the product library and pointreggpt_amd/<variant-lib> alternately, `rounds` times each; then the f16x3 kernel tests with the variant.
cd $GRAFT_REPO_ROOT
T=$1; V=$2; N=${3:-2}
O=gpurun_out/${T}_split_ab.txt
: > $O
for i in $(seq 1 $N); do
  echo "== base (round $i)" >> $O
  bash tools/gpu_split_bench.sh f16x3 2>/dev/null | grep -E "us " >> $O
  echo "== $V (round $i)" >> $O
  bash tools/gpu_split_bench.sh f16x3 pointreggpt_amd/$V 2>/dev/null | grep -E "us " >> $O
done
cat $O
PRG_HIP_LIB=$GRAFT_REPO_ROOT/pointreggpt_amd/$V python -m pytest tests/test_gpu_f16x3.py -q -x -k "split_conv or persistent or kernels or subpixel or upsample" 2>&1 | tail -3 | tee -a $O
