#!/usr/bin/env python3
"""Workload for rocprofv3 passes over the band plan kernel on bf16 model outputs (5000 x 5000, d4, C = 4): a few images through
TileMerger(crops=, defer=True).  PTB_PROF_DTYPE=float32|bfloat16|float16, PTB_PROF_H8=0|1 (ptb_set_tunable 24)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_toolbelt_amd import _native as N  # noqa: E402
from pytorch_toolbelt_amd.inference.tiles import ImageSlicer, TileMerger  # noqa: E402

dev = torch.device("cuda:0")
dt = getattr(torch, os.environ.get("PTB_PROF_DTYPE", "bfloat16"))
N.load().ptb_set_tunable(24, int(os.environ.get("PTB_PROF_H8", "1")))
slicer = ImageSlicer((5000, 5000, 3), 512, 256, weight="pyramid")
crops = slicer.crops
n = len(crops)
outs = [torch.randn((8 * min(8, n - b0), 4, 512, 512), device=dev).to(dt) for b0 in range(0, n, 8)]
pc = [crops[b0:b0 + 8] for b0 in range(0, n, 8)]
m = TileMerger(slicer.target_shape, 4, slicer.weight, device=dev, crops=crops, defer=True)
for _ in range(int(os.environ.get("PTB_PROF_N", "4"))):
    m.reset()
    for t, c in zip(outs, pc):
        m.integrate_batch_deaugment(t, c, group="d4", reduction="mean")
    m.merge()
torch.cuda.synchronize()
