#!/bin/bash
# usage: tools/pmc.sh <tag>   (GPU box, through gpurun).  Three SEPARATE counter passes of the same short workload
# (bench.py --steps 1 --warmup 0 --sampling-steps 2: two transitions + two MaskUnet evaluations at B=64, 128x128, bf16),
# each with --kernel-trace only, as MI355X_MICROARCH.md's HBM / rocprofv3 section prescribes.
set -e
TAG=$1
ROOT=$GRAFT_REPO_ROOT
ARGS="--steps 1 --warmup 0 --streams 1 --sampling-steps 2 --no-cpu-baseline --no-roofline --no-e2e-files --no-drift --no-configs4 --no-parity-mode"
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $ROOT/gpurun_out/${TAG}_$C -o r -- python $ROOT/bench.py $ARGS > $ROOT/gpurun_out/${TAG}_$C.log 2>&1 || true
  python $ROOT/tools/pmc_summary.py $ROOT/gpurun_out/${TAG}_$C/r_counter_collection.csv conv > $ROOT/gpurun_out/${TAG}_${C}_summary.txt
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $ROOT/gpurun_out/${TAG}_SQ -o r -- python $ROOT/bench.py $ARGS > $ROOT/gpurun_out/${TAG}_SQ.log 2>&1 || true
python $ROOT/tools/pmc_summary.py $ROOT/gpurun_out/${TAG}_SQ/r_counter_collection.csv > $ROOT/gpurun_out/${TAG}_SQ_summary.txt
python $ROOT/tools/hbm_traffic.py $ROOT/gpurun_out/${TAG}_FETCH_SIZE/r_counter_collection.csv $ROOT/gpurun_out/${TAG}_WRITE_SIZE/r_counter_collection.csv $ROOT/gpurun_out/${TAG}_conv_hbm_traffic.json
