"""Print a barrier trace written by conv_ws.hip under PRG_WS_TRACE (see phase_barrier).

For every wave (0-3 consumers, 4-7 producers) and phase-in-step p: mean work time (barrier leave -> next arrive) and
the barrier period.  Usage: python tools/ws_trace.py <trace.bin> ...
"""
import sys

import numpy as np

S = 4096
for path in sys.argv[1:]:
    a = np.fromfile(path, dtype=np.uint64).reshape(8, S)
    n = int(a[:, 0].min())
    arr = a[:, 1:1 + 2 * n].reshape(8, n, 2).astype(np.int64)
    arrive, leave = arr[..., 0], arr[..., 1]
    work = arrive[:, 1:] - leave[:, :-1]            # work[:, k-1] = work before barrier k
    period = np.diff(leave.max(axis=0))
    print(f"== {path.split('/')[-1]}: {n} barriers, total {leave.max() - arrive.min()} clk, median period {int(np.median(period))}")
    for p in range(9):
        ks = [k for k in range(10, n - 1) if (k - 1) % 9 == p]
        if not ks:
            continue
        w = np.array([work[:, k - 1] for k in ks])
        per = np.mean([period[k - 1] for k in ks])
        print(f"  p{p}: period {int(per):5d}  work mean {[int(x) for x in w.mean(0)]}  (median {[int(x) for x in np.median(w, 0)]})")
