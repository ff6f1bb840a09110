#!/bin/bash
# round 4: Upsample convs as four 2x2-tap sub-pixel convolutions (conv_w256.hip MODE 2) against the nine-tap gather form: per-launch A/B,
# bench A/B (alternating), full GPU tests
cd $GRAFT_REPO_ROOT
O=gpurun_out
PRG_UP2X2=1 bash tools/prof.sh up_on --streams 1 --no-parity-mode > $O/up_on_summary.txt 2>&1
cd $GRAFT_REPO_ROOT
PRG_UP2X2=0 bash tools/prof.sh up_off --streams 1 --no-parity-mode > $O/up_off_summary.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_seq.py $O/up_on/r_kernel_trace.csv $O/up_off/r_kernel_trace.csv conv > $O/up_ab_seq.txt 2>&1
grep -E "^ +(68|78|88) |sum" $O/up_ab_seq.txt
rm -rf $O/up_on $O/up_off
BA="--steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-e2e-files --no-drift --no-configs4 --no-parity-mode"
rm -f $O/up_ab.txt
for i in 1 2; do
  PRG_UP2X2=1 python bench.py $BA 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.readlines()[-1]); print('up2x2=1', round(r['value'],3))" | tee -a $O/up_ab.txt
  PRG_UP2X2=0 python bench.py $BA 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.readlines()[-1]); print('up2x2=0', round(r['value'],3))" | tee -a $O/up_ab.txt
done
python -m pytest tests -m gpu -q -x > $O/up_fulltests.log 2>&1; echo "pytest rc=$?" >> $O/up_fulltests.log; tail -4 $O/up_fulltests.log
