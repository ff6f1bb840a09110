#!/bin/bash
# c64 conv ablations: one rocprof kernel trace per variant library, print the c64 kernels' average durations
cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-base 4 6 22}; do
  if [ $v = base ]; then unset PRG_HIP_LIB; else export PRG_HIP_LIB=$GRAFT_REPO_ROOT/pointreggpt_amd/libprg_exp$v.so; fi
  OUT=$GRAFT_REPO_ROOT/gpurun_out/c64_abl_$v
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d $OUT -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --sampling-steps 12 --no-cpu-baseline --no-roofline --no-e2e-files --no-drift --sampler-only > $OUT.log 2>&1)
  python - "$OUT/r_kernel_trace.csv" $v <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
per = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "c64" in n:
        per["PRO" if "<true>" in n else "plain"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = []
for k, v in sorted(per.items()):
    big = sorted(v)[len(v) // 2:]          # the level-0 launches (upper half by duration)
    out.append(f"{k}: n={len(v)} level0-median={sorted(big)[len(big)//2]:.1f}us")
print(sys.argv[2], " | ".join(out))
PY
done
