"""GPU: bench.py keeps its contract -- the ONE JSON line the driver parses -- for N = 1 and through the N > 1 code path (two processes
sharing the box's one GPU over gloo: RCCL refuses two ranks on one device, so what this exercises is everything around the fabric:
sharding, pipelined steps, the drain inside the timed region, the per-rank report gathered on rank 0)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def _line(out):
    lines = [ln for ln in out.strip().splitlines() if ln.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


def test_single_gpu_line():
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--repeats", "1", "--no-secondary",
                          "--no-cpu-baseline", "--no-variants", "--placement-tries", "1", "--ramp-max-ms", "300"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert run.returncode == 0, run.stderr[-3000:]
    d = _line(run.stdout)
    assert all(k in d for k in CONTRACT) and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["unit"] == "MP/s"
    assert d["higher_is_better"] is True and d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "cfg2" in d["config"]["workload"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(d["value"] - 25.0 / d["ms_per_step"] * 1e3) / d["value"] < 0.01
    # `value` IS the first-allocation figure (no placement search before it)
    assert d["config"]["first_allocation"]["ms_per_step"] == d["ms_per_step"] and d["config"]["best_placement"] is None
    assert 5000.0 < d["value"] < 16000.0
    par = d["config"]["parity"]       # the timed path's output against the plain merger, every value of the merged map
    assert par["bit_identical"] is True and par["max_abs_diff_vs_plain_merger"] == 0.0 and par["values_checked"] == 4 * 5120 * 5120


def test_two_ranks_on_one_gpu_take_the_sharded_pipelined_path():
    env = dict(os.environ, PTB_BENCH_SAME_GPU="1", PTB_BENCH_BACKEND="gloo")
    run = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29631", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--repeats", "1"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert run.returncode == 0, run.stderr[-3000:]
    d = _line(run.stdout)
    assert all(k in d for k in CONTRACT) and d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "strong"
    sh = d["config"]["sharded"]
    assert sh["mode"].startswith("pipelined") and len(sh["per_rank"]) == 2 and sorted(e["rank"] for e in sh["per_rank"]) == [0, 1]
    assert sum(e["tiles"] for e in sh["per_rank"]) == 361
    for e in sh["per_rank"]:
        assert e["compute_only_ms"] > 0 and e["out_MB_per_link"] and e["in_MB_per_link"]
    assert sh["pipelined_ms_per_image"] > 0 and sh["latency_mode_ms_per_image"] > 0
    assert "cpu_baseline" not in d            # rank 0 at N = 1 only


def test_gpus_2_launches_itself_and_checks_its_band_against_the_single_device_merge():
    """`python bench.py --gpus 2` as the driver types it -- no torch.distributed.run around it, no WORLD_SIZE: bench.py starts its two
    ranks itself (here both on the box's one GPU, over gloo).  Before timing, every rank compares the rows it owns with the single-device
    merge of the same per-tile seeded model outputs; the line carries the worst difference and how many ranks run the library's RCCL
    exchange (none under gloo, and the line says why)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PTB_BENCH_SAME_GPU="1", PTB_BENCH_BACKEND="gloo")
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--repeats", "1"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert run.returncode == 0, run.stderr[-3000:]
    assert "launching" in run.stderr and "torch.distributed.run" in run.stderr
    d = _line(run.stdout)
    assert all(k in d for k in CONTRACT) and d["n_gpus"] == 2 and d["steps"] == 2
    sh = d["config"]["sharded"]
    assert 0.0 <= sh["parity_max_abs_diff"] <= sh["parity_tolerance"] == 1e-5
    assert sh["parity_values_checked"] == 4 * 5120 * 5120, "the two bands together are the whole merged map"
    assert sh["rccl_ranks"] == 0 and "NOT used" in sh["exchange"]
    assert len(sh["per_rank"]) == 2
