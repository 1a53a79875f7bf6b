#!/usr/bin/env python3
"""Does any launch configuration of the band kernel remove its sensitivity to WHERE the model outputs live?  K pools (12.1 GB each,
all kept) x {workgroup order 0 / 1 / 2 (ptb_set_tunable 10)} x {64- / 32-row work items (tunable 11)} x {1024 / 256 rows per launch}:
ms per image of every combination on every pool, same process.   python tools/placement_ab.py [K]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_toolbelt_amd import _native as N  # noqa: E402
from pytorch_toolbelt_amd.inference.tiles import ImageSlicer, TileMerger  # noqa: E402

dev = torch.device("cuda:0")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
slicer = ImageSlicer((5000, 5000, 3), 512, 256, weight="pyramid")
crops = slicer.crops
batches = [(b0, min(len(crops), b0 + 8)) for b0 in range(0, len(crops), 8)]
total = sum(8 * (b1 - b0) for b0, b1 in batches) * 4 * 512 * 512
pools = []
for k in range(K):
    base = torch.empty(total, device=dev, dtype=torch.float32)
    base.normal_()
    outs, off = [], 0
    for b0, b1 in batches:
        n = 8 * (b1 - b0) * 4 * 512 * 512
        outs.append(base[off:off + n].view(8 * (b1 - b0), 4, 512, 512))
        off += n
    pools.append((base, outs))
lib = N.load()


def measure(merger, outs, steps=8):
    def step():
        for t, (b0, b1) in zip(outs, batches):
            merger.integrate_batch_deaugment(t, crops[b0:b1], group="d4", reduction="mean")
        merger.merge()
        merger.reset()
    step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


warm = TileMerger(slicer.target_shape, 4, slicer.weight, device=dev, crops=crops, defer=True)
for _ in range(30):
    measure(warm, pools[0][1], 1)
print(f"{'item rows':>9s} {'rows/launch':>11s} {'order':>5s} | " + " ".join(f"pool{k:d}" for k in range(K)) + " | spread")
ROWS = [int(v) for v in os.environ.get("PTB_AB_ROWS", "1024,256").split(",")]
ITEMS = [int(v) for v in os.environ.get("PTB_AB_ITEMS", "64,32").split(",")]
ORDERS = [int(v) for v in os.environ.get("PTB_AB_ORDERS", "0,1,2").split(",")]
for item_rows in ITEMS:
    assert lib.ptb_set_tunable(11, item_rows) == 0
    for rows in ROWS:
        merger = TileMerger(slicer.target_shape, 4, slicer.weight, device=dev, crops=crops, defer=True, defer_rows=rows)
        for order in ORDERS:
            assert lib.ptb_set_tunable(10, order) == 0
            ms = [measure(merger, o) for _b, o in pools]
            print(f"{item_rows:9d} {rows:11d} {order:5d} | " + " ".join(f"{v:5.3f}" for v in ms) + f" | {max(ms) / min(ms):.3f}")
        lib.ptb_set_tunable(10, 0)
lib.ptb_set_tunable(11, 64)
