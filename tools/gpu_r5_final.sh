#!/bin/bash
# round-5 evidence run (GPU box, through gpurun): full GPU suite, smoke, the driver's bench command, rocprofv3 kernel summaries of the
# precision modes at the headline shape (+ bf16 / mxfp8 at the configs[4] shape), launch-by-launch listings, the three PMC passes and the
# HBM-traffic JSON, the f16x3 conv micro-bench + counters, power / clock / MFMA-busy per mode.  Everything lands in gpurun_out/r5_final_*;
# copy into profiles/r05_end_*.
cd $GRAFT_REPO_ROOT
O=gpurun_out
T=r5_final
python -m pytest tests -m gpu -q -rA > $O/${T}_tests.log 2>&1; echo "pytest rc=$?" >> $O/${T}_tests.log; tail -3 $O/${T}_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/${T}_smoke.log
bash tools/pmc.sh ${T}_pmc > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
rm -rf $O/${T}_pmc_FETCH_SIZE $O/${T}_pmc_WRITE_SIZE $O/${T}_pmc_SQ
cp $O/${T}_pmc_conv_hbm_traffic.json profiles/conv_hbm_traffic.json      # (bench.py reads it when the conv sources' hash matches)
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${T}_bench.json 2> $O/${T}_bench.err
bash tools/prof.sh ${T}_prof_bf16_1lane --streams 1 --no-parity-mode > $O/${T}_summary_by_shape_128_bf16_1lane.txt 2>&1
cd $GRAFT_REPO_ROOT
cp $O/${T}_prof_bf16_1lane/r_kernel_stats.csv $O/${T}_kernel_stats_ddim20_b64_128_bf16_1lane.csv
python tools/prof_seq.py $O/${T}_prof_bf16_1lane/r_kernel_trace.csv conv > $O/${T}_conv_per_launch_128_bf16.txt 2>&1
python tools/prof_seq.py $O/${T}_prof_bf16_1lane/r_kernel_trace.csv _ > $O/${T}_all_per_launch_128_bf16.txt 2>&1
rm -rf $O/${T}_prof_bf16_1lane
for DT in fp32 f16x3; do
  bash tools/gpu_prof_mode.sh $DT ${T}_prof_$DT > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/${T}_f16x3_trace -o r -- python $GRAFT_REPO_ROOT/bench.py --dtype f16x3 --timesteps 10 --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-roofline --no-e2e-files --no-drift --no-configs4 --no-parity-mode > $GRAFT_REPO_ROOT/$O/${T}_f16x3_trace.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_seq.py $O/${T}_f16x3_trace/r_kernel_trace.csv _ > $O/${T}_all_per_launch_128_f16x3.txt 2>&1
rm -rf $O/${T}_f16x3_trace
bash tools/prof.sh ${T}_prof_256_bf16 --streams 1 --no-parity-mode --size 256 --batch 16 > $O/${T}_summary_by_shape_256_b16_bf16.txt 2>&1
cd $GRAFT_REPO_ROOT
bash tools/prof.sh ${T}_prof_256_mx --streams 1 --no-parity-mode --size 256 --batch 16 --dtype mxfp8 > $O/${T}_summary_by_shape_256_b16_mxfp8.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_seq.py $O/${T}_prof_256_bf16/r_kernel_trace.csv $O/${T}_prof_256_mx/r_kernel_trace.csv conv > $O/${T}_conv_per_launch_256_b16_bf16_vs_mxfp8.txt 2>&1
rm -rf $O/${T}_prof_256_bf16 $O/${T}_prof_256_mx
bash tools/gpu_split_bench.sh f16x3 > $O/${T}_split_conv_bench.txt 2>&1
bash tools/gpu_split_pmc.sh > $O/${T}_split_conv_pmc.txt 2>&1
bash tools/gpu_r5_power_modes.sh > $O/${T}_power_modes.log 2>&1
python - <<'PY'
import json
r = json.load(open("gpurun_out/r5_final_bench.json"))
pm = r.get("parity_mode", {})
print("pairs/s", round(r["value"], 3), "ms/step", round(r["ms_per_step"], 1), "conv TF/s", round(r["roofline"]["achieved"], 1), "frac", round(r["roofline"]["frac"], 3),
      "traffic", r["roofline"]["traffic"], "e2e", round(r.get("e2e_files", {}).get("value", 0), 3), "configs4", round(r.get("configs4", {}).get("value", 0), 3),
      "bf16 same shape", round(r.get("configs4", {}).get("bf16_same_shape", {}).get("value", 0), 3))
for k in ("fp32", "f16x3", "f16x3_256_ddim250"):
    if k in pm:
        print(k, round(pm[k]["pairs_per_s"], 3), "pairs/s", round(pm[k]["ms_per_transition"], 2), "ms/transition", "lanes", pm[k].get("streams"), "one lane", pm[k].get("one_lane"),
              "conv", round(pm[k].get("roofline", {}).get("achieved", 0), 1), "TF/s", round(pm[k].get("roofline", {}).get("frac", 0), 3))
print("f16x3/fp32", pm.get("f16x3_vs_fp32"), "tolerance", json.dumps(pm.get("tolerance", {}))[:600])
PY
head -12 $O/${T}_prof_f16x3/prof_summary.txt
head -8 $O/${T}_summary_by_shape_128_bf16_1lane.txt
tail -25 $O/${T}_power_modes.log
