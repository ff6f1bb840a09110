#!/bin/bash
# socket power and shader clock (rocm-smi, every 0.5 s) during the timed loop with one lane and with two:
#   bash tools/gpu_power_lanes.sh   -> gpurun_out/r3_power_lanes.json
cd $GRAFT_REPO_ROOT
for n in 1 2; do
  ( for i in $(seq 1 90); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk" | tr '\n' ' '; echo; sleep 0.5; done ) > gpurun_out/r3_power_trace_$n.txt &
  SAMP=$!
  python bench.py --steps 6 --warmup 2 --streams $n --no-cpu-baseline --no-e2e-files --no-drift --no-configs4 --no-roofline --no-parity-mode > gpurun_out/r3_power_bench_$n.json 2>/dev/null
  kill $SAMP 2>/dev/null; wait $SAMP 2>/dev/null
done
python - <<'PY'
import json, re
out = {}
for n in (1, 2):
    pw, ck = [], []
    for line in open(f"gpurun_out/r3_power_trace_{n}.txt"):
        m = re.search(r"Power \(W\):\s*([0-9.]+)", line)
        c = re.search(r"sclk[^()]*\(([0-9]+)Mhz\)", line)
        if m and c:
            pw.append(float(m.group(1))); ck.append(float(c.group(1)))
    busy = [(p, c) for p, c in zip(pw, ck) if p > 900]          # samples inside the sampler loop
    b = json.load(open(f"gpurun_out/r3_power_bench_{n}.json"))
    out[f"lanes_{n}"] = {"pairs_per_s": b["value"], "samples_in_loop": len(busy),
                         "mean_power_W": sum(p for p, _ in busy) / max(1, len(busy)), "max_power_W": max([p for p, _ in busy] or [0]),
                         "mean_sclk_MHz": sum(c for _, c in busy) / max(1, len(busy))}
out["what"] = "rocm-smi socket power / shader clock sampled every 0.5 s during bench.py --steps 6 --warmup 2 (B = 64, 128x128, 1000-step DDNM, bf16); samples above 900 W = inside the sampler loop; power cap 1400 W, nominal clock 2400 MHz"
json.dump(out, open("gpurun_out/r3_power_lanes.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
