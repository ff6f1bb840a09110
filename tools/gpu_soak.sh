#!/bin/bash
cd $GRAFT_REPO_ROOT
fail=0
for i in 1 2 3 4 5 6 7 8; do
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 900 -k "one_at_a_time or downsample or mxfp8_conv or benchmark_batch" -p no:cacheprovider 2>&1 | tail -1 > gpurun_out/soak_$i.txt
  cat gpurun_out/soak_$i.txt
  grep -q failed gpurun_out/soak_$i.txt && fail=1
done
echo "soak fail=$fail"
