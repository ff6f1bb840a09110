#!/bin/bash
# round-4 evidence run (GPU box, through gpurun): full GPU suite, smoke, the driver's bench command, rocprofv3 kernel summaries of
# the three precision modes at the headline shape, per-launch listing of the bf16 convs, the PMC passes, the f16x3 conv
# micro-bench + counters, the f16 split probe.  Everything lands in gpurun_out/r4_final_*; copy into profiles/r04_end_*.
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests -m gpu -q -rA > $O/r4_final_tests.log 2>&1; echo "pytest rc=$?" >> $O/r4_final_tests.log; tail -3 $O/r4_final_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/r4_final_smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r4_final_bench.json 2> $O/r4_final_bench.err
bash tools/prof.sh r4_final_prof_bf16_1lane --streams 1 --no-parity-mode > $O/r4_final_prof_bf16_1lane_summary.txt 2>&1
cd $GRAFT_REPO_ROOT
cp $O/r4_final_prof_bf16_1lane/r_kernel_stats.csv $O/r4_final_kernel_stats_bf16_1lane.csv
python tools/prof_seq.py $O/r4_final_prof_bf16_1lane/r_kernel_trace.csv conv > $O/r4_final_conv_per_launch_bf16.txt 2>&1
rm -rf $O/r4_final_prof_bf16_1lane
for DT in fp32 f16x3; do
  bash tools/gpu_prof_mode.sh $DT r4_final_prof_$DT > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT
done
bash tools/pmc.sh r4_final_pmc > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
rm -rf $O/r4_final_pmc_FETCH_SIZE $O/r4_final_pmc_WRITE_SIZE $O/r4_final_pmc_SQ
bash tools/gpu_split_bench.sh f16x3 > $O/r4_final_split_conv_bench.txt 2>&1
bash tools/gpu_split_pmc.sh > $O/r4_final_split_conv_pmc.txt 2>&1
tools/micro/f16_split_probe > $O/r4_final_f16_split_probe.txt 2>&1
python - <<'PY'
import json
r = json.load(open("gpurun_out/r4_final_bench.json"))
pm = r.get("parity_mode", {})
print("pairs/s", round(r["value"], 3), "ms/step", round(r["ms_per_step"], 1), "conv TF/s", round(r["roofline"]["achieved"], 1), "frac", round(r["roofline"]["frac"], 3),
      "traffic", r["roofline"]["traffic"], "e2e", round(r.get("e2e_files", {}).get("value", 0), 3), "configs4", round(r.get("configs4", {}).get("value", 0), 3))
for k in ("fp32", "f16x3"):
    if k in pm:
        print(k, round(pm[k]["pairs_per_s"], 3), "pairs/s", round(pm[k]["ms_per_transition"], 2), "ms/transition", "conv", round(pm[k]["roofline"]["achieved"], 1), "TF/s", round(pm[k]["roofline"]["frac"], 3))
print("f16x3/fp32", pm.get("f16x3_vs_fp32"))
PY
head -12 $O/r4_final_prof_f16x3/prof_summary.txt
