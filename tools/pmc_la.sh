set -e
ROOT=$GRAFT_REPO_ROOT
ARGS="--steps 1 --warmup 0 --streams 1 --sampling-steps 2 --no-cpu-baseline --no-roofline --no-e2e-files --no-drift --no-configs4"
cd /tmp && export TMPDIR=/tmp
export PRG_LA_PSUM=0
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $ROOT/gpurun_out/la_SQ1 -o r -- python $ROOT/bench.py $ARGS > $ROOT/gpurun_out/la_SQ1.log 2>&1 || true
python $ROOT/tools/pmc_summary.py $ROOT/gpurun_out/la_SQ1/r_counter_collection.csv la_ | head -20
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace --output-format csv -d $ROOT/gpurun_out/la_SQ2 -o r -- python $ROOT/bench.py $ARGS > $ROOT/gpurun_out/la_SQ2.log 2>&1 || true
python $ROOT/tools/pmc_summary.py $ROOT/gpurun_out/la_SQ2/r_counter_collection.csv la_ | head -20
