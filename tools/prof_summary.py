"""Summarise a rocprofv3 --kernel-trace CSV: per kernel and per (kernel, grid) shape totals."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
# U-Net evaluations in the trace: every forward (denoiser or MaskUnet) launches exactly ONE stem kernel, so the count is read off
# the trace itself (round 4's fixed "22" was wrong by 2x once bench.py ran a lane-setup batch first: VERDICT round 4, item 12)
auto = sum(1 for r in rows if "stem_" in r["Kernel_Name"])
nfw = float(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] != "auto" else float(max(1, auto))
per = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    short = n.replace("(anonymous namespace)::", "").replace("void prg::", "").replace("prg::", "")
    short = short.split("(")[0][:60]
    key = (short, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    per[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in per.values())
print(f"total kernel time {tot / 1e3:.2f} ms; per forward ({nfw:g} U-Net evaluations = stem launches in the trace: {auto}): {tot / 1e3 / nfw:.3f} ms")
conv_t = sum(sum(v) for k, v in per.items() if "conv" in k[0] and "stem" not in k[0] and "head" not in k[0] and "nchw" not in k[0].lower())
nl = sum(len(v) for v in per.values())
print(f"conv kernels {conv_t / 1e3 / nfw:.3f} ms/fwd, everything else {(tot - conv_t) / 1e3 / nfw:.3f} ms/fwd; {nl / nfw:.1f} launches per forward")
kt = collections.defaultdict(float)
for k, v in per.items():
    kt[k[0]] += sum(v)
for k, v in sorted(kt.items(), key=lambda kv: -kv[1])[:25]:
    print(f"  {v / 1e3 / nfw:8.3f} ms/fwd {100 * v / tot:5.1f}%  {k}")
print("-- by shape (top 30)")
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:30]:
    print(f"  {sum(v) / 1e3 / nfw:8.3f} ms/fwd  n/fwd={len(v) / nfw:5.1f} avg {sum(v) / len(v):8.1f} us  {k}")
