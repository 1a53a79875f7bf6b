#!/usr/bin/env python3
"""Does WHERE the model outputs live in device memory change the speed of the headline loop inside ONE process?
Per attempt: (optionally hold a pad allocation of a different size) allocate the 46 output batches of BASELINE configs[1], run the
deferred band merge for a few images, print ms per image, free everything back to the driver.  python tools/placement_probe.py [pads MB ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_toolbelt_amd.inference.tiles import ImageSlicer, TileMerger  # noqa: E402

dev = torch.device("cuda:0")
slicer = ImageSlicer((5000, 5000, 3), 512, 256, weight="pyramid")
crops = slicer.crops
batches = [(b0, min(len(crops), b0 + 8)) for b0 in range(0, len(crops), 8)]
pads = [int(v) for v in sys.argv[1:]] or [0, 0, 1024, 4096, 0, 16384, 0]


def measure(outs, merger, steps=12):
    def step():
        for t, (b0, b1) in zip(outs, batches):
            merger.integrate_batch_deaugment(t, crops[b0:b1], group="d4", reduction="mean")
        merger.merge()
        merger.reset()
    for _ in range(6):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


for attempt, pad_mb in enumerate(pads):
    pad = torch.empty(pad_mb << 20, device=dev, dtype=torch.uint8) if pad_mb else None
    total = sum(8 * (b1 - b0) for b0, b1 in batches) * 4 * 512 * 512
    torch.empty(total, device=dev, dtype=torch.float32)        # one reservation, carved below (bench.py default)
    outs = [torch.empty((8 * (b1 - b0), 4, 512, 512), device=dev).normal_() for b0, b1 in batches]
    merger = TileMerger(slicer.target_shape, 4, slicer.weight, device=dev, crops=crops, defer=True)
    ms = measure(outs, merger)
    print(f"attempt {attempt}: pad {pad_mb:6d} MB  first output at 0x{outs[0].data_ptr():x}  {ms:.4f} ms per image", flush=True)
    del outs, merger, pad
    torch.cuda.empty_cache()
