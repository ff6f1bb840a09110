#!/bin/bash
# round 5 baseline/evidence call: GPU suite, smoke, the driver's bench command, one-lane rocprofv3 summaries of bf16 and f16x3,
# per-launch conv listing, split conv micro-bench.  Output under gpurun_out/$TAG_*.
cd $GRAFT_REPO_ROOT
O=gpurun_out
TAG=${1:-r5_base}
python -m pytest tests -m gpu -q -rA > $O/${TAG}_tests.log 2>&1; echo "pytest rc=$?" >> $O/${TAG}_tests.log; tail -3 $O/${TAG}_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/${TAG}_smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
bash tools/prof.sh ${TAG}_prof_bf16_1lane --streams 1 --no-parity-mode > $O/${TAG}_summary_by_shape_128_bf16_1lane.txt 2>&1
cd $GRAFT_REPO_ROOT
cp $O/${TAG}_prof_bf16_1lane/r_kernel_stats.csv $O/${TAG}_kernel_stats_bf16_1lane.csv
python tools/prof_seq.py $O/${TAG}_prof_bf16_1lane/r_kernel_trace.csv conv > $O/${TAG}_conv_per_launch_128_bf16.txt 2>&1
python tools/prof_seq.py $O/${TAG}_prof_bf16_1lane/r_kernel_trace.csv _ > $O/${TAG}_all_per_launch_128_bf16.txt 2>&1
rm -rf $O/${TAG}_prof_bf16_1lane
bash tools/gpu_prof_mode.sh f16x3 ${TAG}_prof_f16x3 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
bash tools/gpu_split_bench.sh f16x3 > $O/${TAG}_split_conv_bench.txt 2>&1
python - <<PY
import json
r = json.load(open("gpurun_out/${TAG}_bench.json"))
pm = r.get("parity_mode", {})
print("pairs/s", round(r["value"], 3), "ms/step", round(r["ms_per_step"], 1), "conv TF/s", round(r["roofline"]["achieved"], 1), "frac", round(r["roofline"]["frac"], 3),
      "traffic", r["roofline"]["traffic"], "e2e", round(r.get("e2e_files", {}).get("value", 0), 3), "configs4", json.dumps(r.get("configs4", {}))[:600])
print(json.dumps(pm)[:1500])
PY
head -40 $O/${TAG}_prof_f16x3/prof_summary.txt
head -30 $O/${TAG}_summary_by_shape_128_bf16_1lane.txt
cat $O/${TAG}_split_conv_bench.txt | tail -20
