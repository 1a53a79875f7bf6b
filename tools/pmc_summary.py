#!/usr/bin/env python3
"""Per-kernel averages of the counters in rocprofv3 --pmc result databases (rocpd sqlite, ROCm 7.2).
    python tools/pmc_summary.py gpurun_out/pmc/*_results.db [kernel-name-filter]"""
import sqlite3
import sys

flt = None
files = []
for a in sys.argv[1:]:
    (files if a.endswith(".db") else [None]).append(a) if a.endswith(".db") else None
    if not a.endswith(".db"):
        flt = a
for f in files:
    db = sqlite3.connect(f)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else [c for c in cols if "name" in c and "kernel" in c][0]
    cname = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
    val = "value" if "value" in cols else [c for c in cols if "value" in c][0]
    disp = "dispatch_id" if "dispatch_id" in cols else None
    q = f"select {name_col}, {cname}, sum({val}), count(distinct {disp}) from counters_collection group by {name_col}, {cname}"
    for kn, cn, tot, nd in cur.execute(q):
        if flt and flt not in kn:
            continue
        print(f"{kn[:70]:70s} {cn:28s} {tot / max(nd, 1):16.1f} per dispatch ({nd} dispatches)")
