#!/usr/bin/env python3
"""The numbers block README.md / INTEGRATION.md / DESIGN.md carry between `<!-- numbers:begin -->` and `<!-- numbers:end -->`, generated
from the round's committed bench line so that the documents cannot drift from it (tests/test_profile_report_cpu.py compares).

    python tools/docs_numbers.py [profiles/r06_bench.json]            # print the block
    python tools/docs_numbers.py --write                              # rewrite the block in the three documents
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = os.path.join(ROOT, "profiles", "r06_bench.json")
DOCS = ("README.md", "INTEGRATION.md", "DESIGN.md")
BEGIN, END = "<!-- numbers:begin -->", "<!-- numbers:end -->"


def block(path=DEFAULT):
    d = json.loads([ln for ln in open(path).read().splitlines() if ln.startswith("{")][-1])
    c, v, r = d["config"], d["config"]["variants"], d["roofline"]
    s = c.get("secondary") or {}
    lit = c["dropin_literal"]

    def pct(x):
        return f"{100.0 * x:.1f} %"

    rows = [
        ("headline: `TileMerger(crops=, defer=True)` + `integrate_batch_deaugment` (`value`)", f"{d['ms_per_step']:.3f} ms", f"{d['value'] / 1e3:.2f} GP/s", pct(c["region_hbm_frac"])),
        ("the reference's literal loop on the library's defaults: new `TileMerger(shape, C, weight)` per image + `integrate_batch(tta.d4_image_deaugment(y), crops)` + `merge()`"
         f" (merger mode: {lit['merger_mode']})", f"{lit['ms_per_step']:.3f} ms", f"{lit['value_MP_s'] / 1e3:.2f} GP/s", pct(lit["region_hbm_frac"])),
        ("literal loop, one merger + `reset()` per image", f"{v['dropin_literal_ms']:.3f} ms", "", pct(v["dropin_literal_hbm_frac"])),
        ("literal loop, self-planning off (`tiles.set_auto_plan(False)`: round 4's default)", f"{v['dropin_literal_no_self_planning_ms']:.3f} ms", "", ""),
        ("literal loop, `set_strict_dropin()` (no lazy handles, no self-planning)", f"{v['dropin_literal_eager_ms']:.3f} ms", "", ""),
        ("`TileMerger(crops=)` without `defer=` / `TileMerger(auto_plan=False)` + `integrate_batch_deaugment`", f"{v['planned_no_defer_ms']:.3f} / {v['unplanned_fused_ms']:.3f} ms", "", ""),
        ("`band_plan_kernel` per launch (roofline block)", f"{r['avg_launch_ms'] * 1e3:.1f} us", f"{r['achieved']:.0f} GB/s", pct(r["frac"])),
    ]
    for key, name in (("cfg4_fwd", "cfg4 fused focal + Dice + Jaccard forward"), ("cfg4_fwd_bwd", "cfg4 forward + backward"),
                      ("cfg5_mean", "cfg5 multiscale + fliplr, mean"), ("cfg5_gmean", "cfg5 multiscale + fliplr, gmean"),
                      ("lovasz_fwd", "LovaszLoss forward, no gradient"), ("lovasz_fwd_bwd", "LovaszLoss forward + backward")):
        e = s.get(key)
        if isinstance(e, dict) and "ms" in e:
            rows.append((name, f"{e['ms']:.4f} ms", "", pct(e["frac"]) if "frac" in e else ""))
    nt = s.get("no_tta_5000")
    if isinstance(nt, dict) and "ms" in nt:
        rows.append((f"no TTA: new `TileMerger(shape, C, weight)` per image + `integrate_batch(pred, crops)` + `merge()` (merger mode: {nt.get('merger_mode')}; "
                     f"self-planning off: {nt.get('incremental_ms')} ms)", f"{nt['ms']:.3f} ms", "", pct(nt["frac"])))
    if v.get("dropin_literal_bf16_ms"):
        rows.append(("literal loop on bfloat16 model outputs (`torch.autocast`; new merger per image) / `TileMerger(crops=, defer=True)` + `integrate_batch_deaugment` "
                     "on the same tensors (6.48 GB)", f"{v['dropin_literal_bf16_ms']:.3f} / {v['deferred_bands_bf16_ms']:.3f} ms", "",
                     pct(v["dropin_literal_bf16_hbm_frac"]) + " / " + pct(v["deferred_bands_bf16_hbm_frac"])))
    rows[1:1] = spread()
    cb = d.get("cpu_baseline")
    if cb:
        rows.append((f"host CPU, the reference's op chain ({cb['cores']} threads)", "", f"{cb['value']:.1f} MP/s", ""))
    out = [BEGIN, f"Measured by `python bench.py` on one MI355X, `{os.path.relpath(path, ROOT)}` (this table is generated: `tools/docs_numbers.py`):", "",
           "| what | time per 5000 x 5000 image / call | rate | of 8 TB/s |", "|---|---|---|---|"]
    out += [f"| {a} | {b} | {c_} | {e} |" for a, b, c_, e in rows]
    out.append(END)
    return "\n".join(out)


def spread():
    """Round 6: one bench line is one box.  Every committed line of the headline command -- the driver's
    `BENCH_rNN.json` records and the builder's `profiles/rNN_bench*.json` -- as min / median / max, the driver's runs named one by one."""
    import glob

    def last_line(path):
        if os.path.basename(path).startswith("BENCH_"):
            return json.load(open(path)).get("parsed")
        lines = [ln for ln in open(path).read().splitlines() if ln.startswith("{")]
        return json.loads(lines[-1]) if lines else None

    head, lit, driver = [], [], []
    for path in sorted(glob.glob(os.path.join(ROOT, "BENCH_r*.json")) + glob.glob(os.path.join(ROOT, "profiles", "r0*_bench*.json"))):
        try:
            d = last_line(path)
        except Exception:  # noqa: BLE001
            d = None
        if not d or "ms_per_step" not in d or d.get("n_gpus", 1) != 1 or "under_rocprof" in path or "incremental" in path or "sharded" in path:
            continue
        frac = (d.get("config") or {}).get("region_hbm_frac") or (d.get("roofline") or {}).get("frac")
        if frac is None:
            continue
        name = os.path.basename(path)
        head.append((d["ms_per_step"], frac, name))
        if name.startswith("BENCH_"):
            driver.append((int(name[7:9]), d["ms_per_step"], frac))
        l = ((d.get("config") or {}).get("dropin_literal") or {})
        if l.get("ms_per_step"):
            lit.append((l["ms_per_step"], l["region_hbm_frac"], name))

    def mmm(rows):
        rows = sorted(rows)
        lo, md, hi = rows[0], rows[len(rows) // 2], rows[-1]
        return (f"{lo[0]:.3f} / {md[0]:.3f} / {hi[0]:.3f} ms", f"{100 * lo[1]:.1f} / {100 * md[1]:.1f} / {100 * hi[1]:.1f} %", len(rows))

    out = []
    if head:
        t, f, n = mmm(head)
        drv = ", ".join(f"{ms:.2f} ms = {100 * fr:.0f} % (driver, round {r})" for r, ms, fr in sorted(driver, reverse=True))
        out.append((f"headline over all {n} committed bench lines of rounds 1-6 (min / median / max; every line is another box): {drv}", t, "", f))
    if lit:
        t, f, n = mmm(lit)
        out.append((f"literal loop (new merger per image) over the {n} committed lines that carry it (rounds 5-6), min / median / max", t, "", f))
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    text = block(args[0] if args else DEFAULT)
    if "--write" not in sys.argv:
        print(text)
        return
    for name in DOCS:
        p = os.path.join(ROOT, name)
        s = open(p).read()
        if BEGIN not in s:
            raise SystemExit(f"{name}: no {BEGIN} marker")
        s = re.sub(re.escape(BEGIN) + r".*?" + re.escape(END), lambda _m: text, s, flags=re.S)
        open(p, "w").write(s)
        print("rewrote", name)


if __name__ == "__main__":
    main()
