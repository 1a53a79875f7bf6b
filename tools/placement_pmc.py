#!/usr/bin/env python3
"""Workload of the placement / PMC experiment: K pools of model outputs (12.1 GB each) allocated one after
the other and all kept, the deferred band merge run on each in a FIXED order so that the dispatches of a rocprofv3 --pmc pass can
be attributed to pools by position: WARM images on pool 0, then ROUNDS x K pools x IMAGES images (5 band launches per image).
Prints the HIP-event ms per image of every (round, pool).   python tools/placement_pmc.py [K] [ROUNDS] [IMAGES]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_toolbelt_amd.inference.tiles import ImageSlicer, TileMerger  # noqa: E402

WARM = 20
dev = torch.device("cuda:0")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
IMAGES = int(sys.argv[3]) if len(sys.argv) > 3 else 6
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


def image(outs):
    for t, (b0, b1) in zip(outs, batches):
        merger.integrate_batch_deaugment(t, crops[b0:b1], group="d4", reduction="mean")
    merger.merge()
    merger.reset()


for _ in range(WARM):
    image(pools[0][1])
torch.cuda.synchronize()
print(f"K={K} ROUNDS={ROUNDS} IMAGES={IMAGES} WARM={WARM} launches_per_image=5")
for r in range(ROUNDS):
    row = []
    for k, (_base, outs) in enumerate(pools):
        image(outs)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(IMAGES - 1):
            image(outs)
        e1.record()
        torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) / (IMAGES - 1))
    print(f"round {r}: " + " ".join(f"pool{k}:{v:.3f}" for k, v in enumerate(row)) + " ms per image")
print("va " + " ".join(f"pool{k}:0x{b.data_ptr():x}" for k, (b, _o) in enumerate(pools)))
