#!/usr/bin/env python3
"""The headline loop on half-precision model outputs (AMP inference: fp16 / bf16 outputs are read natively by the band kernel and
widened in registers): ms per 5000 x 5000 image and the fraction of 8 TB/s for ITS algorithmic bytes (half the reads), with and
without the band plan kernel's prefetch of the next covering tile (ptb_set_tunable key 21).
    python tools/bench_half.py            # PTB_HALF_ROWS=64,32: also the 32-row work items"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_toolbelt_amd.inference.tiles import ImageSlicer, TileMerger  # noqa: E402

dev = torch.device("cuda:0")
slicer = ImageSlicer((5000, 5000, 3), 512, 256, weight="pyramid")
crops = slicer.crops
n = len(crops)
import itertools

from pytorch_toolbelt_amd import _native as N  # noqa: E402

ROWS = [int(v) for v in os.environ.get("PTB_HALF_ROWS", "64").split(",")]
PFS = [int(v) for v in os.environ.get("PTB_HALF_PF", "2,0").split(",")]
DBS = [int(v) for v in os.environ.get("PTB_HALF_DB", "1").split(",")]      # ptb_set_tunable key 25: double-buffered LDS tiles
for rows, pf, db, dt in itertools.product(ROWS, PFS, DBS, (torch.float32, torch.float16, torch.bfloat16)):
    assert N.load().ptb_set_tunable(25, db) == 0
    assert N.load().ptb_set_tunable(11, rows) == 0          # rows per work item of band plans created from now on
    assert N.load().ptb_set_tunable(21, pf) == 0            # prefetch the next covering tile of half / bf16 sources
    outs = [torch.randn((8 * min(8, n - b0), 4, 512, 512), device=dev).to(dt) for b0 in range(0, n, 8)]
    pc = [crops[b0:b0 + 8] for b0 in range(0, n, 8)]
    m = TileMerger(slicer.target_shape, 4, slicer.weight, device=dev, crops=crops, defer=True)

    def image():
        m.reset()
        for t, c in zip(outs, pc):
            m.integrate_batch_deaugment(t, c, group="d4", reduction="mean")
        return m.merge()

    for _ in range(30):
        image()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40):
        image()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 40
    nbytes = 8 * n * 4 * 512 * 512 * outs[0].element_size() + 4 * 5120 * 5120 * 4
    print(f"item rows {rows:2d} prefetch {pf} lds-db {db} {str(dt):15s} {ms:7.3f} ms per image   {nbytes / 1e9:6.2f} GB algorithmic   {nbytes / (ms * 1e-3) / 1e12:5.2f} TB/s = {nbytes / (ms * 1e-3) / 8e12 * 100:5.1f} % of 8 TB/s")
    del outs, m
    torch.cuda.empty_cache()
