#!/usr/bin/env python3
"""Per-pool averages of PMC counters for the band kernel in a rocprofv3 run of tools/placement_map.py (first phase: 30 warm-up
images on pool 0, then 3 rounds x K pools x 9 images x 5 launches).  python tools/pmc_by_pool.py <results.db> K"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
K = int(sys.argv[2])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
name_col = "kernel_name" if "kernel_name" in cols else [c for c in cols if "name" in c and "kernel" in c][0]
rows = cur.execute(f"select dispatch_id, counter_name, sum(value) from counters_collection where {name_col} like '%band_plan%' group by dispatch_id, counter_name order by dispatch_id").fetchall()
by_counter = defaultdict(list)
for did, cn, v in rows:
    by_counter[cn].append((did, v))
for cn, seq in by_counter.items():
    vals = [v for _, v in sorted(seq)]
    vals = vals[30 * 5:]                       # drop the warm-up images
    per_pool = defaultdict(list)
    block = 9 * 5
    for i in range(0, min(len(vals), 3 * K * block), block):
        per_pool[(i // block) % K].extend(vals[i + 5:i + block])      # skip the first image of each measurement
    print(cn, " ".join(f"pool{k}:{sum(v) / max(len(v), 1):.0f}" for k, v in sorted(per_pool.items())))
