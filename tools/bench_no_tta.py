#!/usr/bin/env python3
"""The plain loop without TTA at the headline geometry (integrate_batch(pred, crops) in batches of 8 + merge(), 361 tiles,
C = 4, 5000 x 5000; 1 933 574 144 algorithmic bytes): a new TileMerger(shape, C, weight) per image on the library's defaults, and the
explicit TileMerger(crops=, defer=True) + reset(); GPU time per image (HIP events around 40 images) and the fraction of 8 TB/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_toolbelt_amd.inference.tiles import ImageSlicer, TileMerger  # noqa: E402

dev = torch.device("cuda:0")
slicer = ImageSlicer((5000, 5000, 3), 512, 256, weight="pyramid")
crops = slicer.crops
n = len(crops)
ALG = n * 4 * 512 * 512 * 4 + 4 * 5120 * 5120 * 4
for dt in (torch.float32, torch.bfloat16):
    preds = [torch.randn((min(8, n - b0), 4, 512, 512), device=dev).to(dt) for b0 in range(0, n, 8)]
    pc = [crops[b0:b0 + 8] for b0 in range(0, n, 8)]
    alg = n * 4 * 512 * 512 * preds[0].element_size() + 4 * 5120 * 5120 * 4

    def literal():
        m = TileMerger(slicer.target_shape, 4, slicer.weight, device=dev)
        for t, c in zip(preds, pc):
            m.integrate_batch(t, c)
        return m.merge()

    md = TileMerger(slicer.target_shape, 4, slicer.weight, device=dev, crops=crops, defer=True)

    def explicit():
        md.reset()
        for t, c in zip(preds, pc):
            md.integrate_batch(t, c)
        return md.merge()

    for name, fn in (("new TileMerger(shape, C, weight) per image (defaults)", literal), ("TileMerger(crops=, defer=True) + reset()", explicit)):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 40
        print(f"{str(dt):15s} {name:55s} {ms:.4f} ms per image = {alg / (ms * 1e-3) / 8e12 * 100:5.1f} % of 8 TB/s for its {alg / 1e9:.2f} GB")

    # rows merged per launch (defer_rows=): 5 launches of ~75 us at the default 1024 -- do fewer, longer launches pay here?
    if os.environ.get("PTB_NO_TTA_ROWS"):
        for rows in (512, 1024, 2048, 2560, 5120):
            mr = TileMerger(slicer.target_shape, 4, slicer.weight, device=dev, crops=crops, defer=True, defer_rows=rows)

            def explicit_rows(mr=mr):
                mr.reset()
                for t, c in zip(preds, pc):
                    mr.integrate_batch(t, c)
                return mr.merge()

            for _ in range(10):
                explicit_rows()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40):
                explicit_rows()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 40
            print(f"{str(dt):15s} defer_rows={rows:5d}: {ms:.4f} ms per image = {alg / (ms * 1e-3) / 8e12 * 100:5.1f} %")

    # in-process A/B of ptb_set_tunable settings on the explicit merger: PTB_NO_TTA_AB="11=32 11=64 27=0 27=1"
    if os.environ.get("PTB_NO_TTA_AB"):
        from pytorch_toolbelt_amd import _native as N

        for rnd in range(2):
            for kv in os.environ["PTB_NO_TTA_AB"].split():
                k, v = (int(x) for x in kv.split("="))
                assert N.load().ptb_set_tunable(k, v) == 0, kv
                mr = TileMerger(slicer.target_shape, 4, slicer.weight, device=dev, crops=crops, defer=True)

                def explicit_ab(mr=mr):
                    mr.reset()
                    for t, c in zip(preds, pc):
                        mr.integrate_batch(t, c)
                    return mr.merge()

                for _ in range(10):
                    explicit_ab()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(40):
                    explicit_ab()
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 40
                print(f"{str(dt):15s} tunable {kv:6s}: {ms:.4f} ms per image = {alg / (ms * 1e-3) / 8e12 * 100:5.1f} %")
