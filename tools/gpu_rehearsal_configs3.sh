#!/bin/bash
# BASELINE configs[3] rehearsal on ONE GPU: one rank's share of the 10k-pair job (-start 0 -stop 1250, B = 64, 128x128,
# 1000-step DDNM, bf16, synthetic scenes and weights) through the two CLIs, wall time per stage.  Usage (GPU box):
#   bash tools/gpu_rehearsal_configs3.sh <tag> [stop]     -> gpurun_out/<tag>.json
TAG=$1; STOP=${2:-1250}
ROOT=$GRAFT_REPO_ROOT
WORK=$(mktemp -d /tmp/prg_rehearsal_XXXX)
cd $WORK
export PYTHONPATH=$ROOT
t0=$(date +%s.%N)
python $ROOT/generate_dataset.py --resume synthetic:1 --synthetic 0 --image_size 128 --timesteps 1000 --sampling_timesteps 1000 \
  --batch_size 64 --dtype bf16 --dataset_name ds -start 0 -stop $STOP > gen.log 2>&1
rc1=$?
t1=$(date +%s.%N)
python $ROOT/generate_gt.py --dataset_name ds -start 0 -stop $STOP --disable_tqdm > gt.log 2>&1
rc2=$?
t2=$(date +%s.%N)
python - <<PY > $ROOT/gpurun_out/$TAG.json
import json, os
n = $STOP
root = "ds"
files = sum(len(f) for _d, _s, f in os.walk(root))
size = sum(os.path.getsize(os.path.join(d, f)) for d, _s, fs in os.walk(root) for f in fs)
gt = os.path.join(root, "metadata", "gt.log")
lines = sum(1 for _ in open(gt)) if os.path.exists(gt) else 0
gen, gts = $t1 - $t0, $t2 - $t1
print(json.dumps({"what": "configs[3] rehearsal: one of eight ranks' share of the 10k-pair dataset on one MI355X, both CLIs, files on local disk",
                  "scenes": n, "rc": [$rc1, $rc2], "generate_dataset_s": gen, "generate_gt_s": gts, "pairs_per_s_generate": n / gen,
                  "pairs_per_s_with_gt": n / (gen + gts), "files": files, "bytes": size, "gt_log_lines": lines,
                  "projected_10k_pairs_8_gpus_s": (gen + gts) * (10000 / 8) / n,
                  "includes": "process start, weight synthesis, library load, graph capture (python start-up ~10 s)"}))
PY
tail -2 gen.log; tail -2 gt.log; cat $ROOT/gpurun_out/$TAG.json
rm -rf $WORK
