"""Print a barrier trace written by conv_ws.hip under PRG_WS_TRACE (see phase_barrier)."""
import sys

import numpy as np

S = 4096
for path in sys.argv[1:]:
    a = np.fromfile(path, dtype=np.uint64).reshape(8, S)
    n = int(a[:, 0].min())
    arr = a[:, 1:1 + 2 * n].reshape(8, n, 2).astype(np.int64)
    arrive, leave = arr[..., 0], arr[..., 1]
    t0 = arrive.min()
    print(f"== {path}: {n} barriers; total {(leave.max() - t0)} clk")
    last = np.argmax(arrive, axis=0)                 # which wave arrived last at each barrier
    period = np.diff(leave.max(axis=0))
    print("  median barrier period (clk):", int(np.median(period)), " p90:", int(np.percentile(period, 90)))
    print("  last-arriver histogram (wave 0-3 consumers, 4.. producers):", np.bincount(last, minlength=8).tolist())
    wait = leave - arrive                            # time each wave spent inside the barrier
    print("  mean wait per wave (clk):", [int(x) for x in wait.mean(axis=1)])
    per = leave.max(axis=0)
    dper = np.diff(per)                                # dper[k-1] = duration of the interval ending at barrier k
    print("  by phase-in-step p = (k-1) % 9: mean period / most frequent last arriver")
    for p in range(9):
        ks = [k for k in range(10, n) if (k - 1) % 9 == p]
        if ks:
            lastw = np.bincount(last[ks], minlength=8)
            print(f"    p{p}: {int(np.mean([dper[k - 1] for k in ks])):6d} clk   last: wave {int(lastw.argmax())} ({int(lastw.max())}/{len(ks)})")
    k0 = min(40, n - 1)
    print("  arrival offsets vs earliest, barriers", k0, "..", k0 + 17)
    for k in range(k0, min(n, k0 + 18)):
        rel = arrive[:, k] - arrive[:, k].min()
        print(f"   b{k:4d} period {int(leave[:, k].max() - leave[:, k - 1].max()):6d}  " + " ".join(f"{int(x):5d}" for x in rel))
