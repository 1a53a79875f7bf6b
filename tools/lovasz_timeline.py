#!/usr/bin/env python3
"""Timeline of ONE LovaszLoss forward + backward out of a rocprofv3 --kernel-trace database of tools/prof_lovasz.py (PTB_PROF_BIG_ONLY=1):
every launch of the 10th iteration with its duration and the idle gap in front of it -- where the call's time goes that no single
kernel's average shows.

    python tools/lovasz_timeline.py /tmp/prof_lov/run_results.db
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
starts = [i for i, r in enumerate(rows) if "lovasz_error_hist_kernel" in r[0]]
i0, i1 = starts[9], starts[10]           # iteration 10 of the 20 forward + backward calls
t_prev = rows[i0][1]
busy = 0
print("| # | kernel | gap us | duration us |")
print("|---|---|---|---|")
for n, (name, s, e) in enumerate(rows[i0:i1]):
    short = name.split("(")[0].replace("void ", "").replace("ptb::", "")
    print(f"| {n} | `{short[:90]}` | {(s - t_prev) / 1e3:.1f} | {(e - s) / 1e3:.1f} |")
    busy += e - s
    t_prev = e
span = rows[i1][1] - rows[i0][1]
print(f"\n{i1 - i0} launches; kernels {busy / 1e3:.1f} us + gaps {(span - busy) / 1e3:.1f} us = {span / 1e3:.1f} us from one error kernel to the next")
