#!/bin/bash
# round 4: h16 (f16 h1 tensors + packed-f16 prologue) against the bf16 form, same box: tests, bench A/B (alternating), per-launch A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out
tools/micro/valu_rates | tee $O/h16_valu_rates.txt
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fast_paths or bf16 or unet" > $O/h16_tests.log 2>&1; echo "pytest rc=$?" >> $O/h16_tests.log; tail -5 $O/h16_tests.log
BA="--steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-e2e-files --no-drift --no-configs4 --no-parity-mode"
for i in 1 2; do
  PRG_H16=7 python bench.py $BA 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.readlines()[-1]); print('h16=1', round(r['value'],3))" | tee -a $O/h16_ab.txt
  PRG_H16=0 python bench.py $BA 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.readlines()[-1]); print('h16=0', round(r['value'],3))" | tee -a $O/h16_ab.txt
done
PRG_H16=7 bash tools/prof.sh h16_on --streams 1 --no-parity-mode > $O/h16_on_summary.txt 2>&1
cd $GRAFT_REPO_ROOT
PRG_H16=0 bash tools/prof.sh h16_off --streams 1 --no-parity-mode > $O/h16_off_summary.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_seq.py $O/h16_on/r_kernel_trace.csv $O/h16_off/r_kernel_trace.csv conv > $O/h16_ab_seq.txt 2>&1
tail -70 $O/h16_ab_seq.txt
head -3 $O/h16_on_summary.txt $O/h16_off_summary.txt
rm -rf $O/h16_on $O/h16_off
