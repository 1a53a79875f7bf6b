#!/usr/bin/env python3
"""Same device memory, different RELATIVE placement of the batch tensors: one big allocation per pool, batch i carved at
i * (256 MiB + delta) for several deltas; the headline loop timed for each.  python tools/placement_offsets.py [pools]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_toolbelt_amd.inference.tiles import ImageSlicer, TileMerger  # noqa: E402

dev = torch.device("cuda:0")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
slicer = ImageSlicer((5000, 5000, 3), 512, 256, weight="pyramid")
crops = slicer.crops
batches = [(b0, min(len(crops), b0 + 8)) for b0 in range(0, len(crops), 8)]
per = 64 * 4 * 512 * 512                      # elements of a full batch
deltas = [0, 1024, 16 * 1024, 256 * 1024, (1 << 20) + 4096, (2 << 20) + 8192, (5 << 20) + 12288, (17 << 20) + 20480]   # bytes
slack = max(deltas) // 4 * len(batches)
pools = [torch.empty(per * len(batches) + slack, device=dev).normal_() for _ in range(K)]
merger = TileMerger(slicer.target_shape, 4, slicer.weight, device=dev, crops=crops, defer=True)


def loop_ms(outs, steps=8):
    def step():
        for t, (b0, b1) in zip(outs, batches):
            merger.integrate_batch_deaugment(t, crops[b0:b1], group="d4", reduction="mean")
        merger.merge()
        merger.reset()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


for _ in range(20):
    loop_ms([pools[0][i * per:i * per + 8 * (b1 - b0) * 4 * 512 * 512].view(-1, 4, 512, 512) for i, (b0, b1) in enumerate(batches)], 1)
print("delta between consecutive batch tensors (bytes): " + "  ".join(f"{d:>9d}" for d in deltas))
for k, pool in enumerate(pools):
    row = []
    for d in deltas:
        st = per + d // 4
        outs = [pool[i * st:i * st + 8 * (b1 - b0) * 4 * 512 * 512].view(-1, 4, 512, 512) for i, (b0, b1) in enumerate(batches)]
        row.append(loop_ms(outs))
    print(f"pool {k}: ms per image                               " + "  ".join(f"{v:9.3f}" for v in row))
