#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -s --timeout 1500 > gpurun_out/r2_t7.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_t7.log
grep -v "^   \|^$" gpurun_out/r2_t7.log | tail -12
python __graft_entry__.py smoke 2>&1 | tail -2
bash tools/prof.sh r2_prof128e > gpurun_out/r2_prof128e_summary.txt 2>&1
cd $GRAFT_REPO_ROOT; head -24 gpurun_out/r2_prof128e_summary.txt
python bench.py --size 256 --sampling-steps 250 --batch 16 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2_bench256_v3.json 2> gpurun_out/r2_bench256_v3.err
python -c "
import json; r=json.load(open('gpurun_out/r2_bench256_v3.json')); print('256: pairs/s', r['value'], 'conv TF', r['roofline']['achieved'], 'e2e', r.get('e2e_files',{}).get('value'))"
