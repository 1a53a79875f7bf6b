#!/usr/bin/env python3
"""Where the host time of one integrate_batch_deaugment call goes (cProfile over the bench loop, no device syncs)."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_toolbelt_amd.inference.tiles import ImageSlicer, TileMerger  # noqa: E402

dev = torch.device("cuda:0")
slicer = ImageSlicer((5000, 5000, 3), 512, 256, weight="pyramid")
crops = slicer.crops
x = torch.randn((64, 4, 512, 512), device=dev)
planned = len(sys.argv) > 1 and sys.argv[1] in ("planned", "deferred")
deferred = len(sys.argv) > 1 and sys.argv[1] == "deferred"
m = TileMerger(slicer.target_shape, 4, slicer.weight, device=dev, crops=crops if planned else None, defer=deferred)
batches = [crops[b0:b0 + 8] for b0 in range(0, len(crops), 8)]
tensors = [x[:8 * len(c)] for c in batches]


def image():
    m.reset()
    for t, c in zip(tensors, batches):
        m.integrate_batch_deaugment(t, c, group="d4", reduction="mean")
    return m.merge()


for _ in range(3):
    image()
torch.cuda.synchronize()
if "--no-gc-freeze" not in sys.argv:
    import gc

    gc.collect()
    gc.freeze()   # (a generation-2 pass over torch's import-time objects costs 30-45 ms)
t0 = time.perf_counter()
for _ in range(20):
    image()
host = (time.perf_counter() - t0) / 20
torch.cuda.synchronize()
print(f"host issue per image: {host * 1e3:.3f} ms = {host / 46 * 1e6:.1f} us per integrate call ({'deferred' if deferred else 'planned' if planned else 'unplanned'})")
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    image()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
