#!/usr/bin/env python3
"""Host cost of ONE integrate_batch call on a deferred / self-planned merger (the loop of BASELINE.md section 3, no TTA, and the d4 literal
loop): wall time per call with the GPU idle-free (tiny tiles keep the kernels negligible), and a cProfile by own time.  Diagnostic (GPU box)."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_toolbelt_amd.inference import tta  # noqa: E402
from pytorch_toolbelt_amd.inference.tiles import ImageSlicer, TileMerger  # noqa: E402

dev = torch.device("cuda:0")
# the headline's crop list (361 tiles in 46 batches) on tiles small enough that the GPU never is the bottleneck: 5000/8 = 625, 64/32
slicer = ImageSlicer((625, 625, 3), 64, 32, weight="pyramid")
crops = slicer.crops
n = len(crops)
batches = [(b0, min(n, b0 + 8)) for b0 in range(0, n, 8)]
plain = [torch.randn((b1 - b0, 4, 64, 64), device=dev) for b0, b1 in batches]
d4 = [torch.randn((8 * (b1 - b0), 4, 64, 64), device=dev) for b0, b1 in batches]
crop_slices = [crops[b0:b1] for b0, b1 in batches]
print("tiles", n, "calls per image", len(batches))


def image_plain():
    m = TileMerger(slicer.target_shape, 4, slicer.weight, device=dev)
    for y, c in zip(plain, crop_slices):
        m.integrate_batch(y, c)
    return m, m.merge()


def image_d4():
    m = TileMerger(slicer.target_shape, 4, slicer.weight, device=dev)
    for y, c in zip(d4, crop_slices):
        m.integrate_batch(tta.d4_image_deaugment(y), c)
    return m, m.merge()


def image_ext(m):
    m.reset()
    for y, c in zip(d4, crop_slices):
        m.integrate_batch_deaugment(y, c, group="d4", reduction="mean")
    return m, m.merge()


for name, fn in (("plain integrate_batch, new merger per image", image_plain), ("literal d4 loop, new merger per image", image_d4)):
    for _ in range(5):
        m, _r = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        m, _r = fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"{name}: {(t1 - t0) / 200 * 1e3:.4f} ms per image = {(t1 - t0) / 200 / len(batches) * 1e6:.2f} us per call (ctor + merge included); mode {m.mode}")
m = TileMerger(slicer.target_shape, 4, slicer.weight, device=dev, crops=crops, defer=True)
for _ in range(5):
    image_ext(m)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    image_ext(m)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"integrate_batch_deaugment on TileMerger(crops=, defer=True) + reset(): {(t1 - t0) / 200 * 1e3:.4f} ms per image = {(t1 - t0) / 200 / len(batches) * 1e6:.2f} us per call; mode {m.mode}")
# phases of the plain loop
tc = tl = tm = 0.0
for _ in range(200):
    a = time.perf_counter()
    m = TileMerger(slicer.target_shape, 4, slicer.weight, device=dev)
    b = time.perf_counter()
    for y, c in zip(plain, crop_slices):
        m.integrate_batch(y, c)
    c_ = time.perf_counter()
    m.merge()
    d = time.perf_counter()
    tc, tl, tm = tc + b - a, tl + c_ - b, tm + d - c_
print(f"plain loop phases (us per image): ctor {tc / 200 * 1e6:.1f}, {len(batches)} calls {tl / 200 * 1e6:.1f} ({tl / 200 / len(batches) * 1e6:.2f} each), merge {tm / 200 * 1e6:.1f}")
if "--profile" in sys.argv:
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(100):
        image_plain()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
