#!/bin/bash
# tools/gpu.sh <tag> <stage> [<stage> ...]      (run on the GPU box:  gpurun --timeout N -- 'bash tools/gpu.sh r6a tests bench')
# One parametrised evidence script (replaces the per-call gpu_r5_call*.sh of round 5).  Everything lands in gpurun_out/<tag>_*;
# copy what is to be judged into profiles/r06_*.  Stages:
#   tests            full `-m gpu` suite (log with -rA)          tests:<expr>   pytest -k <expr>
#   smoke            __graft_entry__.smoke()
#   bench            the driver's command: bench.py --gpus 1 --steps 20 --warmup 5 (contract line + sidecar)
#   benchq           the same with --steps 4 --warmup 2 (quick)
#   prof:<dtype>     rocprofv3 --kernel-trace --stats of 20 DDIM transitions (bf16 / mxfp8) or 10 ancestral ones (fp32 / f16x3), one lane
#   pmc              the three PMC passes (FETCH_SIZE, WRITE_SIZE, SQ_*) + conv_hbm_traffic.json
#   splitpmc         SQ counters of the f16x3 conv micro-bench, wave-specialised kernel vs the one-wave-per-SIMD kernel
#   split            tools/split_bench.py (f16x3 conv micro-bench) + its counters
#   hostsat          tools/host_saturation.py (8-rank host-side replay, GPU idle)
#   power            power / clock / MFMA-busy per precision mode
#   sh:<script>      bash tools/<script>
cd $GRAFT_REPO_ROOT
O=gpurun_out
T=$1; shift
mkdir -p $O
for ST in "$@"; do
  cd $GRAFT_REPO_ROOT
  echo "=== stage $ST ($(date +%T))"
  case $ST in
    tests) python -m pytest tests -m gpu -q -rA > $O/${T}_tests.log 2>&1; echo "pytest rc=$?" >> $O/${T}_tests.log; tail -4 $O/${T}_tests.log;;
    tests:*) python -m pytest tests -m gpu -q -rA -k "${ST#tests:}" > $O/${T}_tests_k.log 2>&1; echo "pytest rc=$?" >> $O/${T}_tests_k.log; grep -E "passed|failed|rc=|^FAILED|^ERROR" $O/${T}_tests_k.log | tail -12;;
    smoke) python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/${T}_smoke.log;;
    bench|benchq)
      A="--gpus 1 --steps 20 --warmup 5"; [ $ST = benchq ] && A="--gpus 1 --steps 4 --warmup 2"
      T0=$(date +%s)
      python bench.py $A > $O/${T}_bench_line.json 2> $O/${T}_bench.err; echo "bench rc=$? wall $(( $(date +%s) - T0 )) s"
      cp bench_full.json $O/${T}_bench_full.json 2>/dev/null
      wc -c $O/${T}_bench_line.json; cat $O/${T}_bench_line.json; tail -c 600 $O/${T}_bench.err | grep -v '^{' ;;
    prof:*)
      DT=${ST#prof:}
      if [ $DT = bf16 ] || [ $DT = mxfp8 ]; then
        bash tools/prof.sh ${T}_prof_$DT --streams 1 --no-parity-mode --dtype $DT > $O/${T}_summary_by_shape_128_${DT}_1lane.txt 2>&1
        cd $GRAFT_REPO_ROOT
        cp $O/${T}_prof_$DT/r_kernel_stats.csv $O/${T}_kernel_stats_ddim20_b64_128_${DT}_1lane.csv
        python tools/prof_seq.py $O/${T}_prof_$DT/r_kernel_trace.csv _ > $O/${T}_all_per_launch_128_$DT.txt 2>&1
        rm -rf $O/${T}_prof_$DT
        head -40 $O/${T}_summary_by_shape_128_${DT}_1lane.txt
      else
        bash tools/gpu_prof_mode.sh $DT ${T}_prof_$DT > /dev/null 2>&1
        cd $GRAFT_REPO_ROOT
        head -40 $O/${T}_prof_$DT/prof_summary.txt
      fi;;
    pmc)
      bash tools/pmc.sh ${T}_pmc > /dev/null 2>&1
      cd $GRAFT_REPO_ROOT
      rm -rf $O/${T}_pmc_FETCH_SIZE $O/${T}_pmc_WRITE_SIZE $O/${T}_pmc_SQ
      cp $O/${T}_pmc_conv_hbm_traffic.json profiles/conv_hbm_traffic.json      # (a later `bench` stage of this call reads it; commit the same file)
      cat $O/${T}_pmc_conv_hbm_traffic.json | head -c 600; echo;;
    splitpmc)
      PRG_SPLIT_W512=0 bash tools/gpu_split_pmc.sh > $O/${T}_split_conv_pmc_ws.txt 2>&1; grep -E "ws_kernel|p64_kernel|Kernel|kernel" $O/${T}_split_conv_pmc_ws.txt | head -30
      PRG_SPLIT_W512=2 bash tools/gpu_split_pmc.sh > $O/${T}_split_conv_pmc_w512.txt 2>&1; grep -E "w512_kernel|p64_kernel" $O/${T}_split_conv_pmc_w512.txt | head -30;;
    split)
      bash tools/gpu_split_bench.sh f16x3 > $O/${T}_split_conv_bench.txt 2>&1; tail -30 $O/${T}_split_conv_bench.txt
      bash tools/gpu_split_pmc.sh > $O/${T}_split_conv_pmc.txt 2>&1; tail -20 $O/${T}_split_conv_pmc.txt;;
    hostsat) python tools/host_saturation.py --out $O/${T}_host_saturation_8x.json 2>&1 | tail -30;;
    power) bash tools/gpu_power_modes.sh > $O/${T}_power_modes.log 2>&1; cp $O/power_modes.json $O/${T}_power_clock_mfma_busy_per_mode.json; tail -25 $O/${T}_power_modes.log;;
    sh:*) bash tools/${ST#sh:} 2>&1 | tail -60;;
    *) echo "unknown stage $ST";;
  esac
done
echo "=== done ($(date +%T))"
