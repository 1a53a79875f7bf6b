#!/usr/bin/env python3
"""Does the placement of the MERGER'S OWN buffers (merged map, normaliser, window, band table) move the headline loop?  (profiles/r06_bench.json:
the headline merger ran at 2.139 ms, a second merger of the same configuration on the same model outputs at 1.923 ms.)  One pool of model
outputs, K mergers whose buffers are allocated with different amounts of padding in front, each timed on the same loop; then the buffers of
the fastest and the slowest are swapped one at a time."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_toolbelt_amd.inference.tiles import ImageSlicer, TileMerger  # noqa: E402

dev = torch.device("cuda:0")
slicer = ImageSlicer((5000, 5000, 3), 512, 256, weight="pyramid")
crops = slicer.crops
n = len(crops)
K = int(os.environ.get("PTB_AB_MERGERS", "6"))
mergers, pads = [], []
first = TileMerger(slicer.target_shape, 4, slicer.weight, device=dev, crops=crops, defer=True)      # (like bench.py: the merger exists before the model outputs)
g = torch.Generator(device=dev).manual_seed(0)
outs = [torch.randn((8 * min(8, n - b0), 4, 512, 512), device=dev, generator=g) for b0 in range(0, n, 8)]
pc = [crops[b0:b0 + 8] for b0 in range(0, n, 8)]
mergers.append(("created before the model outputs", first))
for k in range(K):
    pads.append(torch.empty((k * 37 + 1) * (1 << 20) + k * 4096, dtype=torch.uint8, device=dev))      # shifts where the next allocations land
    mergers.append((f"created after them, +{pads[-1].numel() >> 20} MiB of padding", TileMerger(slicer.target_shape, 4, slicer.weight, device=dev, crops=crops, defer=True)))


def image(m, keep=None):
    m.reset()
    if keep is not None:
        m._merged = keep
    for t, c in zip(outs, pc):
        m.integrate_batch_deaugment(t, c, group="d4", reduction="mean")
    return m.merge()


def timed(m, keep=None, steps=20):
    for _ in range(3):
        image(m, keep)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        image(m, keep)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


res = []
for rnd in range(2):
    for name, m in mergers:
        ms = timed(m)
        out = image(m)
        res.append((ms, name, out.data_ptr(), m._plan.norm_full.data_ptr(), m.weight.data_ptr(), m._deferred.bands.table.data_ptr()))
        print(f"round {rnd} {name:55s} {ms:.4f} ms   merged @{out.data_ptr():#x}  norm @{m._plan.norm_full.data_ptr():#x}  window @{m.weight.data_ptr():#x}  table @{m._deferred.bands.table.data_ptr():#x}")
# fixed output buffers: the same merger writing into maps at different addresses
name, m = mergers[0]
bufs = [torch.empty((4, 5120, 5120), device=dev) for _ in range(4)]
for b in bufs:
    print(f"merger 0 writing into a map @{b.data_ptr():#x}: {timed(m, b):.4f} ms")
# ping-pong: consecutive images write into DIFFERENT maps (what a caller that keeps the previous result alive gets from the allocator)
for depth in (1, 2, 3):
    ring = bufs[:depth]
    state = {"i": 0}

    def image_ring(m=m):
        state["i"] += 1
        return image(m, ring[state["i"] % depth])

    for _ in range(4):
        image_ring()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        image_ring()
    e1.record()
    torch.cuda.synchronize()
    print(f"merger 0, consecutive images into a ring of {depth} map(s): {e0.elapsed_time(e1) / 30:.4f} ms")
