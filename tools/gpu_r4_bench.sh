#!/bin/bash
# round 4: the driver's bench command + the two new bench tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bench" > gpurun_out/r4c/bench_tests.log 2>&1; tail -5 gpurun_out/r4c/bench_tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4c/bench_driver.json 2> gpurun_out/r4c/bench_driver.err
tail -3 gpurun_out/r4c/bench_driver.err
python - <<'PY'
import json
j = json.load(open("gpurun_out/r4c/bench_driver.json"))
print("value", j["value"], "roofline", j["roofline"]["frac"], j["roofline"]["achieved"])
pm = j["parity_mode"]
for k in ("fp32", "f16x3"):
    print(k, pm[k]["pairs_per_s"], pm[k]["ms_per_transition"], pm[k].get("roofline", {}).get("achieved"), pm[k].get("roofline", {}).get("frac"))
print("f16x3/fp32", pm["f16x3_vs_fp32"])
for n, r in j["drift_vs_reference"].items():
    if isinstance(r, dict):
        print(n, {d: (r[d]["xyz_linf_m"], r[d]["inpainted_depth_m"]["mean"]) for d in ("bf16", "mxfp8", "f16x3") if d in r})
print("cpu", j["cpu_baseline"]["value"], j["cpu_baseline"]["cores"], j["cpu_baseline"]["host_filled"])
print("configs4", j["configs4"]["value"], "e2e", j["e2e_files"]["value"])
PY
