#!/usr/bin/env python3
"""Map of the headline loop's speed over device memory: K pools of model outputs (12.1 GB each) are allocated one after the other
and ALL kept, then the deferred band merge is timed on each pool in turn (two rounds, so drift over time shows as a difference
between the rounds, placement as a difference between the pools).  python tools/placement_map.py [K]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_toolbelt_amd.inference.tiles import ImageSlicer, TileMerger  # noqa: E402

dev = torch.device("cuda:0")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 12
slicer = ImageSlicer((5000, 5000, 3), 512, 256, weight="pyramid")
crops = slicer.crops
batches = [(b0, min(len(crops), b0 + 8)) for b0 in range(0, len(crops), 8)]
total = sum(8 * (b1 - b0) for b0, b1 in batches) * 4 * 512 * 512
pools = []
for k in range(K):
    base = torch.empty(total, device=dev, dtype=torch.float32)          # ONE device allocation per pool
    base.normal_()
    outs, off = [], 0
    for b0, b1 in batches:
        n = 8 * (b1 - b0) * 4 * 512 * 512
        outs.append(base[off:off + n].view(8 * (b1 - b0), 4, 512, 512))
        off += n
    pools.append((base, outs))
merger = TileMerger(slicer.target_shape, 4, slicer.weight, device=dev, crops=crops, defer=True)


def measure(outs, steps=8):
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


for _ in range(30):
    measure(pools[0][1], 1)
rounds = [[measure(o) for _, o in pools] for _ in range(3)]
# the same memory carved into batches of 9 (and 7) tiles: the 8 views of a tile are then 36 (28) MiB apart instead of 32 MiB
def recarve(base, nb):
    bb = [(b0, min(len(crops), b0 + nb)) for b0 in range(0, len(crops), nb)]
    outs, off = [], 0
    for b0, b1 in bb:
        n = 8 * (b1 - b0) * 4 * 512 * 512
        outs.append(base[off:off + n].view(8 * (b1 - b0), 4, 512, 512))
        off += n
    return bb, outs


alt = {}
for nb in (9, 7):
    res = []
    for base, _ in pools:
        bb, outs = recarve(base, nb)
        batches_saved = batches
        batches = bb
        res.append(measure(outs))
        batches = batches_saved
    alt[nb] = res
for k, (base, _) in enumerate(pools):
    print(f"pool {k:2d} (allocated {k * total * 4 / 2**30:6.1f} GiB into the process, va 0x{base.data_ptr():x}): " + "  ".join(f"{r[k]:.3f}" for r in rounds) +
          f" ms per image | batches of 9: {alt[9][k]:.3f}  of 7: {alt[7][k]:.3f}")
