#!/usr/bin/env python3
"""Placement at batch granularity: M separate 256 MiB buffers ([64, 4, 512, 512] model outputs of one 8-tile batch) are timed one by
one with the plain d4 de-augment kernel; then the headline loop is run on the 46 fastest, the 46 slowest and the first 46.
python tools/placement_fine.py [M]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_toolbelt_amd.inference import tta  # noqa: E402
from pytorch_toolbelt_amd.inference.tiles import ImageSlicer, TileMerger  # noqa: E402

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 184
slicer = ImageSlicer((5000, 5000, 3), 512, 256, weight="pyramid")
crops = slicer.crops
batches = [(b0, min(len(crops), b0 + 8)) for b0 in range(0, len(crops), 8)]
bufs = [torch.empty((64, 4, 512, 512), device=dev).normal_() for _ in range(M)]


def t_one(buf, reps=6):
    tta.d4_image_deaugment(buf)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        tta.d4_image_deaugment(buf)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for b in bufs[:20]:
    t_one(b)
us = np.array([[t_one(b) for b in bufs] for _ in range(2)]).min(0)
print("per-buffer d4 de-augment us: min %.1f  p25 %.1f  median %.1f  p75 %.1f  max %.1f" % (us.min(), np.percentile(us, 25), np.median(us), np.percentile(us, 75), us.max()))
print("in allocation order:", " ".join(f"{v:.0f}" for v in us))
merger = TileMerger(slicer.target_shape, 4, slicer.weight, device=dev, crops=crops, defer=True)


def loop_ms(sel, steps=10):
    outs = [bufs[i][:8 * (b1 - b0)] for i, (b0, b1) in zip(sel, batches)]

    def step():
        for t, (b0, b1) in zip(outs, batches):
            merger.integrate_batch_deaugment(t, crops[b0:b1], group="d4", reduction="mean")
        merger.merge()
        merger.reset()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


order = np.argsort(us)
n = len(batches)
for name, sel in (("first 46 allocated", list(range(n))), ("46 fastest", list(order[:n])), ("46 slowest", list(order[-n:])), ("first 46 allocated", list(range(n))),
                  ("46 fastest", list(order[:n]))):
    print(f"headline loop on the {name}: {loop_ms(sel):.3f} ms per image")
