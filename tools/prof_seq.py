"""Launch-by-launch view of one U-Net evaluation from a rocprofv3 --kernel-trace CSV: the kernel sequence of a graph replay
is periodic, so launch i of the period is the same layer in every replay; prints the average duration per position.
usage: prof_seq.py <kernel_trace.csv> [substring filter, default conv]   (two CSVs: side-by-side A/B of the same positions)"""
import csv
import sys


def load(path, flt):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    seq = []
    for r in rows:
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void prg::", "").replace("prg::", "").split("(")[0]
        seq.append((n, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    names = [n for n, _ in seq]
    # period of the middle of the sequence (the first and last transitions of a run differ from the steady ones)
    N = len(names)
    P = None
    for mid in (N // 2, N // 4, (3 * N) // 4, N // 3):      # (the middle of a two-batch trace is the gap between the batches)
        for cand in range(50, N // 6):
            if names[mid:mid + cand] == names[mid + cand:mid + 2 * cand] == names[mid - cand:mid]:
                P = cand
                break
        if P:
            break
    if not P:
        raise SystemExit("no period found")
    starts = [s for s in range(mid % P, N - P + 1, P) if names[s:s + P] == names[mid:mid + P]]
    # rotate the window so that it starts at the U-Net's first kernel (the stem conv)
    rot = next((i for i in range(P) if "stem" in names[mid + i]), 0)
    starts = [s + rot for s in starts if s + rot + P <= N and names[s + rot:s + rot + P] == names[mid + rot:mid + rot + P]]
    reps = len(starts)
    out = []
    for i in range(P):
        n = names[starts[0] + i]
        if flt in n:
            d = [seq[s + i][1] for s in starts]
            out.append((i, n[:44], sum(d) / len(d)))
    return P, reps, out


flt = "conv"
paths = [a for a in sys.argv[1:] if a.endswith(".csv")]
for a in sys.argv[1:]:
    if not a.endswith(".csv"):
        flt = a
res = [load(p, flt) for p in paths]
print("period", [r[0] for r in res], "replays", [r[1] for r in res])
if len(res) == 1:
    for i, n, d in res[0][2]:
        print(f"{i:4d} {d:8.1f} us  {n}")
    print("sum", round(sum(d for _, _, d in res[0][2]), 1))
else:
    a, b = res[0][2], res[1][2]
    assert len(a) == len(b), (len(a), len(b))
    for (i, n, d), (j, m, e) in zip(a, b):
        print(f"{i:4d} {d:8.1f} us  {n:44s} | {e:8.1f} us  {m if m != n else ''}")
    print("sum", round(sum(d for _, _, d in a), 1), round(sum(d for _, _, d in b), 1))
