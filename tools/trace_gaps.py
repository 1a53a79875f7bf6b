#!/usr/bin/env python3
"""Gaps between consecutive launches of one kernel in a rocprofv3 --kernel-trace database (rocpd sqlite).
    python tools/trace_gaps.py <results.db> <kernel-name-substring>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
sel = [(s, e) for n, s, e in rows if sys.argv[2] in n]
big = [(s, e) for s, e in sel if e - s > 0.5 * sorted(e2 - s2 for s2, e2 in sel)[len(sel) // 2]]
gaps = [(big[i + 1][0] - big[i][1]) / 1e3 for i in range(len(big) - 1)]
gaps_s = sorted(gaps)
durs = [(e - s) / 1e3 for s, e in big]
print(f"{len(big)} launches, duration avg {sum(durs) / len(durs):.1f} us; gap to the next launch: median {gaps_s[len(gaps_s) // 2]:.1f} us, "
      f"p10 {gaps_s[len(gaps_s) // 10]:.1f}, p90 {gaps_s[9 * len(gaps_s) // 10]:.1f}, mean {sum(gaps) / len(gaps):.1f}")
print("last 25 gaps (us):", [round(g, 1) for g in gaps[-25:]])
print("last 25 durations (us):", [round(d, 1) for d in durs[-25:]])
mid = len(gaps) // 2
print("25 gaps from the middle of the run (us):", [round(g, 1) for g in gaps[mid:mid + 25]])
print("durations there (us):", [round(d, 1) for d in durs[mid:mid + 25]])
other = [(n, s, e) for n, s, e in rows if sys.argv[2] not in n]
print("other kernels in the trace:", len(other))
