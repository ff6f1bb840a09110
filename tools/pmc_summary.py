"""Summarise a rocprofv3 --pmc counter_collection CSV per (kernel, grid)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in rows:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void prg::", "").replace("prg::", "").split("(")[0][:52]
    key = (n, int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])) if "Grid_Size" in r else 0)
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(key, r["Counter_Name"])] += 1
names = sorted({r["Counter_Name"] for r in rows})
filt = sys.argv[2] if len(sys.argv) > 2 else ""
print("counters:", names)
for key, d in sorted(agg.items(), key=lambda kv: -kv[1].get(names[0], 0)):
    if filt and filt not in key[0]:
        continue
    n = max(1, cnt[(key, names[0])])
    print(key, "launches", n, {k: f"{v / n:.4g}" for k, v in d.items()})
