#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database: per-kernel count / total / avg / min / max duration (like --stats),
and, when counters were collected, per-kernel averages of each PMC counter.

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db [--md]
"""
import sqlite3
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select * from kernels").fetchall()
    ix = {c: i for i, c in enumerate(cols)}
    name_col = "name" if "name" in ix else [c for c in cols if "name" in c][0]
    stats = defaultdict(list)
    for r in rows:
        stats[r[ix[name_col]]].append(r[ix["end"]] - r[ix["start"]])
    total = sum(sum(v) for v in stats.values())
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for name, v in sorted(stats.items(), key=lambda kv: -sum(kv[1])):
        short = name if len(name) < 110 else name[:107] + "..."
        print(f"| `{short}` | {len(v)} | {sum(v) / 1e6:.3f} | {sum(v) / len(v) / 1e3:.2f} | {min(v) / 1e3:.2f} | {max(v) / 1e3:.2f} | {100 * sum(v) / total:.1f} |")
    try:
        pcols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        prow = cur.execute("select * from counters_collection").fetchall()
    except sqlite3.Error:
        prow = []
    if prow:
        px = {c: i for i, c in enumerate(pcols)}
        kn = "kernel_name" if "kernel_name" in px else [c for c in pcols if "name" in c and "counter" not in c][0]
        cn = "counter_name" if "counter_name" in px else [c for c in pcols if "counter" in c and "name" in c][0]
        vn = "value" if "value" in px else [c for c in pcols if "value" in c][0]
        agg = defaultdict(lambda: defaultdict(list))
        for r in prow:
            agg[r[px[kn]]][r[px[cn]]].append(r[px[vn]])
        print("\n| kernel | counter | dispatches | avg per dispatch |")
        print("|---|---|---|---|")
        for k, d in agg.items():
            short = k if len(k) < 90 else k[:87] + "..."
            for c, v in sorted(d.items()):
                print(f"| `{short}` | {c} | {len(v)} | {sum(v) / len(v):.1f} |")


if __name__ == "__main__":
    main()
