#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats result database (rocpd sqlite, ROCm 7.2) as a markdown table.
    python tools/rocprof_summary.py gpurun_out/prof/x_results.db [min_calls] > profiles/rNN_kernel_stats.md"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else cols[0]
rows = cur.execute(f"select {name_col}, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) from kernels group by {name_col} order by 3 desc").fetchall()
total = sum(r[2] for r in rows) or 1
print("| kernel | calls | total ms | avg us | min us | max us | % |")
print("|---|---|---|---|---|---|---|")
for name, n, tot, avg, mn, mx in rows[:40]:
    short = name if len(name) < 150 else name[:147] + "..."
    print(f"| `{short}` | {n} | {tot / 1e6:.3f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100.0 * tot / total:.1f} |")
