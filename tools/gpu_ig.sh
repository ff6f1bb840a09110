#!/bin/bash
# round 4: two-stage igemm prefetch — parity tests that exercise it, per-launch listing (bf16), f16x3 kernel summary
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests -m gpu -q -x -k "unet or conv or fast_paths or mxfp8 or f16x3 or maskunet" > $O/ig_tests.log 2>&1; echo "pytest rc=$?" >> $O/ig_tests.log; tail -4 $O/ig_tests.log
bash tools/prof.sh ig_new --streams 1 --no-parity-mode > $O/ig_new_summary.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_seq.py $O/ig_new/r_kernel_trace.csv conv > $O/ig_seq.txt 2>&1
grep -E "igemm|sum" $O/ig_seq.txt
head -3 $O/ig_new_summary.txt
rm -rf $O/ig_new
bash tools/gpu_prof_mode.sh f16x3 igfix2 | head -14
