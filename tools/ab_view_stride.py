#!/usr/bin/env python3
"""Two falsifiable experiments on the band plan kernel's placement sensitivity, in ONE process on ONE
pool of model outputs, interleaved:

  (i)  the eight d4 view streams of a batch lie exactly 32 MiB apart (view stride = 8 tiles x 4 channels x 1 MiB).  Here the same model
       outputs are laid out with a PADDED view stride (+64 KiB ... +1 MiB + 64 KiB) -- the C ABI takes the stride as an argument
       (ptb_band_plan_submit) -- so the streams no longer alias modulo any power of two;
  (ii) ptb_set_tunable(22, 1): odd work items issue their view loads starting at view 4 (registers, reduction order and bits unchanged).

If DRAM bank / channel aliasing of the 32-MiB-apart streams is what makes some pools slow, (i) and / or (ii) speed a slow pool up.
    python tools/ab_view_stride.py            # one line per variant: ms per image, fraction of 8 TB/s; min over ROUNDS interleaved rounds"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_toolbelt_amd import _native as N  # noqa: E402
from pytorch_toolbelt_amd.inference.tiles import ImageSlicer, TileMerger  # noqa: E402

dev = torch.device("cuda:0")
lib = N.load()
slicer = ImageSlicer((5000, 5000, 3), 512, 256, weight="pyramid")
crops = slicer.crops
n, C, T, V, B = len(crops), 4, 512, 8, 8
per_tile = C * T * T
PADS_KIB = [int(v) for v in os.environ.get("PTB_AB_PADS_KIB", "0,64,192,1088").split(",")]
ROUNDS = int(os.environ.get("PTB_AB_ROUNDS", "3"))
STEPS = int(os.environ.get("PTB_AB_STEPS", "20"))
ALG = V * n * per_tile * 4 + C * 5120 * 5120 * 4

m = TileMerger(slicer.target_shape, C, slicer.weight, device=dev, crops=crops, defer=True)
bands, plan = m._deferred.bands, m._plan
merged = torch.empty((C, 5120, 5120), device=dev)
varr = N.int_array([0, 5, 6, 3, 1, 4, 7, 2])
stream = N.stream_ptr(dev)
g = torch.Generator(device=dev).manual_seed(0)
starts = list(range(0, n, B))
layouts = {}
for pad in PADS_KIB:      # every layout gets its own pool of buffers, allocated up front (so that all variants live side by side)
    pad_el = pad * 1024 // 4
    bufs = []
    for b0 in starts:
        nb = min(B, n - b0)
        vs = nb * per_tile + pad_el
        buf = torch.empty(V * vs, device=dev)
        buf.normal_(generator=g)
        bufs.append((buf, nb, vs))
    layouts[pad] = bufs


def image(pad):
    lib.ptb_band_plan_reset(bands.handle)
    pos = 0
    for buf, nb, vs in layouts[pad]:
        rc = lib.ptb_band_plan_submit(bands.handle, pos, nb, buf.data_ptr(), per_tile, vs, N.F32, V, varr, N.RED_MEAN, merged.data_ptr(),
                                      plan.norm_full.data_ptr(), m.weight.data_ptr(), stream)
        assert rc >= 0, rc
        pos += nb


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(STEPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / STEPS


# bits: every layout / rotation gives the same map for the same values (the pads hold other random numbers, so compare pad 0 with itself rotated)
image(PADS_KIB[0])
ref = merged.clone()
lib.ptb_set_tunable(22, 1)
image(PADS_KIB[0])
assert torch.equal(ref, merged), "rotated view issue order changed the result"
lib.ptb_set_tunable(22, 0)
best = {}
for r in range(ROUNDS):
    for rot in (0, 1):
        lib.ptb_set_tunable(22, rot)
        for pad in PADS_KIB:
            ms = timed(lambda: image(pad))
            best[(pad, rot)] = min(best.get((pad, rot), 1e9), ms)
lib.ptb_set_tunable(22, 0)
base = best[(PADS_KIB[0], 0)]
print(f"band plan kernel, 5000 x 5000 d4 C=4 fp32, {ROUNDS} interleaved rounds x {STEPS} images, min per variant; first-allocation pools of this process")
for (pad, rot), ms in sorted(best.items()):
    print(f"view stride 32 MiB + {pad:5d} KiB, view issue order {'rotated on odd items' if rot else 'v0..v7             '}: {ms:.4f} ms per image = "
          f"{ALG / (ms * 1e-3) / 8e12 * 100:5.1f} % of 8 TB/s  ({(ms / base - 1) * 100:+5.1f} % vs default)")
