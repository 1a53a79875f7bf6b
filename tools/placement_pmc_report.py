#!/usr/bin/env python3
"""Per-pool statistics of the band kernel's counters in a rocprofv3 --pmc run of tools/placement_pmc.py: for every counter the sum
per launch and -- when the database keeps one row per hardware instance (TCC channel / XCC) -- the imbalance max / mean over the
instances, averaged over the launches of a pool.   python tools/placement_pmc_report.py <results.db> K ROUNDS IMAGES [WARM]"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
K, ROUNDS, IMAGES = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
WARM = int(sys.argv[5]) if len(sys.argv) > 5 else 20
cur = db.cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
src = "counters_collection" if "counters_collection" in tables else [t for t in tables if "counter" in t.lower() and "collect" in t.lower()][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({src})")]
print(f"# source {src}: columns {cols}")
name_col = "kernel_name" if "kernel_name" in cols else [c for c in cols if "name" in c and "kernel" in c][0]
rows = cur.execute(f"select dispatch_id, counter_name, value from {src} where {name_col} like '%band_plan%' order by dispatch_id").fetchall()
per = defaultdict(lambda: defaultdict(list))          # counter -> dispatch -> [instance values]
for did, cn, v in rows:
    per[cn][did].append(float(v))
for cn, by_disp in sorted(per.items()):
    dids = sorted(by_disp)
    dids = dids[WARM * 5:]
    block = IMAGES * 5
    tot, imb, ninst = defaultdict(list), defaultdict(list), 0
    for i, did in enumerate(dids[:ROUNDS * K * block]):
        pool = (i // block) % K
        if i % block < 5:
            continue          # the first image on a pool
        vals = by_disp[did]
        ninst = max(ninst, len(vals))
        tot[pool].append(sum(vals))
        if len(vals) > 1 and sum(vals) > 0:
            imb[pool].append(max(vals) / (sum(vals) / len(vals)))
    line = f"{cn:34s} inst={ninst:4d} " + " ".join(f"pool{k}:{sum(v) / max(len(v), 1):.4g}" for k, v in sorted(tot.items()))
    if imb:
        line += " | max/mean over instances " + " ".join(f"pool{k}:{sum(v) / len(v):.3f}" for k, v in sorted(imb.items()))
    print(line)
