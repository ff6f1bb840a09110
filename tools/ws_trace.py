"""Print a barrier trace written by conv_ws.hip under PRG_WS_TRACE (see phase_barrier).

For every wave (0-3 consumers, 4-11 producers) and phase-in-step p: mean work time (barrier leave -> next arrive) and
the barrier period.  Usage: python tools/ws_trace.py <trace.bin> ...
"""
import sys

import numpy as np

S = 4096
for path in sys.argv[1:]:
    a = np.fromfile(path, dtype=np.uint64).reshape(-1, S)
    n = int(a[:, 0].min())
    NW = a.shape[0]
    arr = a[:, 1:1 + 2 * n].reshape(NW, n, 2).astype(np.int64)
    arrive, leave = arr[..., 0], arr[..., 1]
    work = arrive[:, 1:] - leave[:, :-1]            # work[:, k-1] = work before barrier k
    period = np.diff(leave.max(axis=0))
    wall = int(a[0, S - 1]) - int(a[0, S - 2])          # 100 MHz ticks between the first and the last barrier of wave 0
    clk = int(leave[0, n - 1] - leave[0, 0])
    mhz = clk / (wall / 100.0) if wall > 0 else float("nan")
    print(f"== {path.split('/')[-1]}: {n} barriers, total {leave.max() - arrive.min()} clk, median period {int(np.median(period))}, shader clock {mhz:.0f} MHz")
    for p in range(9):
        ks = [k for k in range(10, n - 1) if (k - 1) % 9 == p]
        if not ks:
            continue
        w = np.array([work[:, k - 1] for k in ks])
        per = np.mean([period[k - 1] for k in ks])
        m0 = a[4:, S // 2:S // 2 + 2 * n:2].astype(np.int64)      # producers: after the weight wait
        m1 = a[4:, S // 2 + 1:S // 2 + 2 * n:2].astype(np.int64)  # ... all issued, before the LDS wait
        if m0.any():
            ww = np.array([m0[:, k] - leave[4:, k - 1] for k in ks]).mean(0)
            bb = np.array([m1[:, k] - m0[:, k] for k in ks]).mean(0)
            lw = np.array([arrive[4:, k] - m1[:, k] for k in ks]).mean(0)
            print(f"       producers: weight wait {[int(x) for x in ww]}  body {[int(x) for x in bb]}  LDS drain {[int(x) for x in lw]}")
        print(f"  p{p}: period {int(per):5d}  work mean {[int(x) for x in w.mean(0)]}  (median {[int(x) for x in np.median(w, 0)]})")
