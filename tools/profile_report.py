#!/usr/bin/env python3
"""Turn gpurun_out/prof/*/run_results.db (rocprofv3 rocpd sqlite) into the committed evidence under profiles/:

    python tools/profile_report.py r01        -> profiles/r01_kernel_stats.md, profiles/r01_pmc.md, profiles/traffic.json

HBM bytes follow guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are in KiB, collected in
separate passes, and on gfx950 FETCH_SIZE counts 64 B per 128 B request for wide coalesced reads, so it is doubled.
The merge kernel's exactly-known byte count (read 524 288 000 B, write 419 430 400 B) is printed as the calibration."""
import json
import os
import sqlite3
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "gpurun_out", "prof")


def kernels(db):
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    ix = {c: i for i, c in enumerate(cols)}
    out = defaultdict(list)
    order = " order by start" if "start" in ix else ""
    for r in cur.execute("select * from kernels" + order):
        out[r[ix["name"]]].append((r[ix["end"]] - r[ix["start"]], r[ix.get("grid_size_x", ix.get("grid_size", 0))] if ("grid_size_x" in ix or "grid_size" in ix) else 0))
    return out


def counters(db):
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    ix = {c: i for i, c in enumerate(cols)}
    kn = "kernel_name" if "kernel_name" in ix else [c for c in cols if "name" in c and "counter" not in c][0]
    out = defaultdict(lambda: defaultdict(list))
    for r in cur.execute("select * from counters_collection"):
        out[r[ix[kn]]][r[ix["counter_name"]]].append(r[ix["value"]])
    return out


def short(name):
    return name.replace("ptb::", "").replace("void ", "")[:100]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    # on the GPU box only gpurun_out/ travels back: PTB_PROFILE_OUT=gpurun_out/profiles writes the reports there (the raw
    # databases can then be deleted on the box; they exceed the 64 MiB that is copied back)
    outdir = os.environ.get("PTB_PROFILE_OUT") or os.path.join(ROOT, "profiles")
    os.makedirs(outdir, exist_ok=True)
    ks = kernels(os.path.join(PROF, "trace", "run_results.db"))
    total = sum(d for v in ks.values() for d, _ in v)
    lines = [f"# {tag}: rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline", "",
             "| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for name, v in sorted(ks.items(), key=lambda kv: -sum(d for d, _ in kv[1])):
        d = [x for x, _ in v]
        lines.append(f"| `{short(name)}` | {len(d)} | {sum(d) / 1e6:.3f} | {sum(d) / len(d) / 1e3:.2f} | {min(d) / 1e3:.2f} | {max(d) / 1e3:.2f} | {100 * sum(d) / total:.1f} |")
    band = [n for n in ks if "band_plan_kernel" in n and "6166440" in n]
    if band:
        seq = [x for x, _ in ks[band[0]]]          # in launch order
        d = sorted(seq)
        med = d[len(d) // 2]
        full = [x for x in seq if x > 0.5 * med]   # (the variants block also runs 256-row launches: they are listed, not averaged here)
        # bench.py measures the headline on the pool as first allocated; AFTER it, config.placement / config.best_placement try
        # candidate pools of different speed (their launches are at the end of the trace and are reported separately)
        tail_n = 0
        try:
            line = json.loads(open(os.path.join(ROOT, "gpurun_out", "bench_under_rocprof.json")).read().strip().splitlines()[-1])
            tail_n = 5 * int(line["config"]["placement"].get("steps_after_the_headline", 0))
        except Exception:
            pass
        head, tail = (full[:-tail_n], full[-tail_n:]) if 0 < tail_n < len(full) else (full, [])
        lines += ["", f"Dominant kernel `{short(band[0])}`: {len(seq)} launches, of which {len(full)} are 1024-row launch groups of the headline "
                      f"configuration (5 per image).  The first {len(head)} read the model-output pool as first allocated (probe step, ramp, warm-up, "
                      f"timed steps, variants): average **{sum(head) / max(len(head), 1) / 1e3:.2f} us**, as in bench.py's roofline block; the last "
                      f"{len(tail)} belong to the placement search that follows the headline (candidate pools + the best one timed: average "
                      f"{sum(tail) / max(len(tail), 1) / 1e3:.2f} us, min {min(tail or [0]) / 1e3:.2f}, max {max(tail or [0]) / 1e3:.2f})."]
    accum = [n for n in ks if "view_accum_kernel" in n]
    if accum:
        d = sorted(x for x, _ in ks[accum[0]])
        full = [x for x in d if x > 0.5 * d[len(d) // 2]]
        lines += ["", f"Dominant kernel `{short(accum[0])}`: {len(full)} full 8-tile launches, average {sum(full) / len(full) / 1e3:.2f} us "
                      f"(the {len(d) - len(full)} one-tile tail launches of each image are excluded, as in bench.py's roofline block)."]
    open(os.path.join(outdir, f"{tag}_kernel_stats.md"), "w").write("\n".join(lines) + "\n")

    rows = ["| kernel | counter | dispatches | avg per dispatch |", "|---|---|---|---|"]
    agg = {}
    for sub in ("FETCH_SIZE", "WRITE_SIZE", "FETCH_SIZE_cal", "WRITE_SIZE_cal", "TCC_HIT_sum", "SQ_WAVES", "SQ_LDS_BANK_CONFLICT"):
        db = os.path.join(PROF, sub, "run_results.db")
        if not os.path.exists(db):
            continue
        for k, d in counters(db).items():
            for c, v in sorted(d.items()):
                if c in ("FETCH_SIZE", "WRITE_SIZE") and ("view_accum" in k or "band_plan" in k):
                    med = sorted(v)[len(v) // 2]
                    v = [x for x in v if x > 0.5 * med]  # full launches only (8-tile batches / 1024-row launch groups)
                agg[(k, c)] = sum(v) / len(v)
                rows.append(f"| `{short(k)}` | {c} | {len(v)} | {sum(v) / len(v):.1f} |")
    notes = ["", "HBM bytes per launch (FETCH_SIZE x 2 x 1024 + WRITE_SIZE x 1024; gfx950 correction from guides/MI355X_MICROARCH.md):", ""]
    traffic = {}
    for k in sorted({k for k, _ in agg}):
        f, w = agg.get((k, "FETCH_SIZE")), agg.get((k, "WRITE_SIZE"))
        if f is None or w is None:
            continue
        b = f * 2 * 1024 + w * 1024
        notes.append(f"* `{short(k)}`: read {f * 2 * 1024 / 1e6:.1f} MB + write {w * 1024 / 1e6:.1f} MB = **{b / 1e6:.1f} MB**")
        if "view_accum_kernel" in k and "6166440" in k:
            traffic["view_accum_d4_bytes_per_launch"] = int(b)
            traffic["view_accum_d4_read_bytes"] = int(f * 2 * 1024)
            traffic["view_accum_d4_write_bytes"] = int(w * 1024)
        if "band_plan_kernel" in k and "6166440" in k:
            traffic["band_plan_d4_bytes_per_launch"] = int(b)
            traffic["band_plan_d4_read_bytes"] = int(f * 2 * 1024)
            traffic["band_plan_d4_write_bytes"] = int(w * 1024)
        if "merge_div" in k:
            traffic["merge_bytes_per_launch"] = int(b)
            notes.append(f"  (calibration: the merge kernel must read 524 288 000 B and write 419 430 400 B; measured {f * 2 * 1024:.0f} / {w * 1024:.0f})")
    open(os.path.join(outdir, f"{tag}_pmc.md"), "w").write(f"# {tag}: rocprofv3 --pmc <counter> --kernel-trace (one pass per counter group)\n\n" + "\n".join(rows + notes) + "\n")
    traffic["source"] = f"profiles/{tag}_pmc.md"
    tj = os.path.join(outdir, "traffic.json")
    try:
        old = json.load(open(tj))      # keep what another run (the incremental merger, --no-defer) measured
    except Exception:
        old = {}
    if "view_accum_d4_bytes_per_launch" in old and "view_accum_d4_bytes_per_launch" not in traffic:
        for k in ("view_accum_d4_bytes_per_launch", "view_accum_d4_read_bytes", "view_accum_d4_write_bytes"):
            traffic[k] = old[k]
        traffic["view_accum_source"] = old.get("view_accum_source", old.get("source", "").replace("_pmc.md", "_pmc_incremental.md"))
    json.dump(traffic, open(tj, "w"), indent=1)
    print("\n".join(lines[-3:]))
    print("\n".join(notes))


if __name__ == "__main__":
    main()
