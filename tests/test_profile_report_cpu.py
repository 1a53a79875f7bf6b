"""tools/profile_report.py on a synthetic rocpd database: bench.py measures the headline on the pool as first allocated and runs its
placement search AFTERWARDS (config.placement / config.best_placement); the search's launches (candidate pools of different speed, at
the end of the trace) must be reported apart from the headline's, whose average is what bench.py's roofline block is compared with."""
import json
import os
import sqlite3
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def test_report_separates_the_placement_search(tmp_path):
    prof = ROOT / "gpurun_out" / "prof" / "trace"
    made_dirs = not prof.exists()
    prof.mkdir(parents=True, exist_ok=True)
    db = prof / "run_results.db"
    backup = db.read_bytes() if db.exists() else None
    line_path = ROOT / "gpurun_out" / "bench_under_rocprof.json"
    line_backup = line_path.read_text() if line_path.exists() else None
    try:
        con = sqlite3.connect(db)
        con.execute("drop table if exists kernels")
        con.execute("create table kernels (id integer primary key, name text, start integer, end integer, grid_size_x integer)")
        name = "void ptb::band_plan_kernel<8, 6166440, 0, 1>(ptb::ViewArgs, ptb::BandItem const*, ptb::GroupTiles)"
        t = 0
        rows = []
        for i in range(100):         # probe, ramp, warm-up, timed steps, variants on the pool as first allocated: 400 us
            rows.append((name, t, t + 400_000, 10240)); t += 500_000
        for i in range(20):          # a variant with one 256-row band per launch: short launches, not averaged
            rows.append((name, t, t + 100_000, 2560)); t += 200_000
        for i in range(50):          # 10 steps of the search + best-placement timing on slow / fast candidate pools: 460 us
            rows.append((name, t, t + 460_000, 10240)); t += 500_000
        con.executemany("insert into kernels (name, start, end, grid_size_x) values (?, ?, ?, ?)", rows)
        con.commit()
        con.close()
        line_path.write_text(json.dumps({"config": {"placement": {"steps_run_by_the_search": 4, "steps_after_the_headline": 10}}}) + "\n")
        out = tmp_path / "out"
        env = dict(os.environ, PTB_PROFILE_OUT=str(out))
        r = subprocess.run([sys.executable, str(ROOT / "tools" / "profile_report.py"), "t00"], env=env, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-2000:]
        text = (out / "t00_kernel_stats.md").read_text()
        assert "The first 100 read the model-output pool as first allocated" in text and "the last 50 belong to the placement search" in text
        assert "average 460.00 us" in text and "average **400.00 us**" in text
    finally:
        if backup is not None:
            db.write_bytes(backup)
        else:
            db.unlink(missing_ok=True)
        if line_backup is not None:
            line_path.write_text(line_backup)
        else:
            line_path.unlink(missing_ok=True)
        if made_dirs:
            for d in (prof, prof.parent):
                try:
                    d.rmdir()
                except OSError:
                    pass


def test_documents_quote_the_committed_bench_line():
    """README.md, INTEGRATION.md and DESIGN.md carry a numbers block generated from the round's committed bench line
    (tools/docs_numbers.py over profiles/r06_bench.json): the documents cannot drift from the measurement (round 4's review found
    INTEGRATION.md quoting 2.43 ms for the literal loop against 2.59 in the last bench JSON)."""
    sys.path.insert(0, str(ROOT / "tools"))
    try:
        import docs_numbers
    finally:
        sys.path.pop(0)
    want = docs_numbers.block()
    assert "deferred bands" in want and "literal loop" in want
    for name in docs_numbers.DOCS:
        text = (ROOT / name).read_text()
        assert want in text, f"{name}: the numbers block is not the one tools/docs_numbers.py generates from profiles/r06_bench.json (python tools/docs_numbers.py --write)"
    line = json.loads([ln for ln in (ROOT / "profiles" / "r05_bench.json").read_text().splitlines() if ln.startswith("{")][-1])
    lit = line["config"]["dropin_literal"]
    assert lit["merger_mode"] == "deferred bands" and lit["region_hbm_frac"] >= 0.70, "the literal loop of the committed line is below the 70 % it is documented at"
