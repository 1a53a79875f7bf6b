#!/usr/bin/env python3
"""Per-INSTANCE statistics (16 L2 channels x 8 XCDs = 128 TCC instances) of the band kernel's counters in a rocprofv3 --pmc run of
tools/placement_pmc.py.  The `counters_collection` view of rocpd sums a counter over its instances; the `pmc_events` view keeps one row
per instance (in a fixed order per dispatch), which is what separates "the same requests, spread less evenly over the channels" from
"the same requests, evenly spread, served later".

    python tools/placement_channels.py <results.db> K ROUNDS IMAGES [WARM]
"""
import sqlite3
import sys
from collections import defaultdict

import numpy as np

db = sqlite3.connect(sys.argv[1])
K, ROUNDS, IMAGES = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
WARM = int(sys.argv[5]) if len(sys.argv) > 5 else 20
cur = db.cursor()
rows = cur.execute("select dispatch_id, counter_name, counter_value from pmc_events where name like '%band_plan%' order by dispatch_id, id").fetchall()
per = defaultdict(lambda: defaultdict(list))          # counter -> dispatch -> [instance values in id order]
for did, cn, v in rows:
    per[cn][did].append(float(v))
for cn, by_disp in sorted(per.items()):
    dids = sorted(by_disp)[WARM * 5:]
    block = IMAGES * 5
    prof = defaultdict(list)
    for i, did in enumerate(dids[:ROUNDS * K * block]):
        pool = (i // block) % K
        if i % block < 5:
            continue          # the first image on a pool
        prof[pool].append(by_disp[did])
    ninst = len(next(iter(by_disp.values())))
    print(f"## {cn}: {ninst} instances per launch")
    print("| pool | sum / launch | max / mean | min / mean | std / mean | XCD sums max/mean (8 x 16 fold) | channel sums max/mean (fold) | busiest 4 instances |")
    print("|---|---|---|---|---|---|---|---|")
    base = None
    for pool, vals in sorted(prof.items()):
        a = np.asarray(vals, dtype=np.float64).mean(axis=0)
        m = a.mean() or 1.0
        fold = a.reshape(8, -1) if ninst % 8 == 0 else a.reshape(1, -1)
        xs, cs = fold.sum(axis=1), fold.sum(axis=0)
        top = np.argsort(-a)[:4]
        corr = ""
        if base is None:
            base = a
        elif a.std() > 0 and base.std() > 0:
            corr = f" (r vs pool 0: {np.corrcoef(a, base)[0, 1]:.2f})"
        print(f"| {pool} | {a.sum():.4g} | {a.max() / m:.3f} | {a.min() / m:.3f} | {a.std() / m:.3f} | {xs.max() / xs.mean():.3f} | {cs.max() / cs.mean():.3f} | "
              + " ".join(f"{int(t)}:{a[t] / m:.2f}" for t in top) + corr + " |")
    print()
