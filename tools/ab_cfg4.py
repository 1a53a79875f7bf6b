#!/usr/bin/env python3
"""A/B of a loss-kernel tunable at the cfg4 shape ([32,16,512,512] logits + int64 labels): BinaryFocalLoss / DiceLoss / JaccardLoss /
FocalDiceJaccardLoss forward (no_grad) and forward + backward, values printed so that two settings can be compared.
    python tools/ab_cfg4.py <tunable key> <value A> <value B>"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_toolbelt_amd import _native as N  # noqa: E402
from pytorch_toolbelt_amd import losses as L  # noqa: E402


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    key, values = int(sys.argv[1]), [int(v) for v in sys.argv[2:]]
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn((32, 16, 512, 512), device=dev, generator=g)
    labels = torch.randint(0, 16, (32, 512, 512), device=dev, generator=g)
    crits = {"BinaryFocalLoss": L.BinaryFocalLoss(), "BinaryFocalLoss(normalized)": L.BinaryFocalLoss(normalized=True), "DiceLoss": L.DiceLoss("multiclass"),
             "JaccardLoss": L.JaccardLoss("multiclass"), "FocalDiceJaccardLoss": L.FocalDiceJaccardLoss("multiclass")}
    lib = N.load()
    fwd_bytes = x.numel() * 4 + labels.numel() * 8
    for _ in range(200):
        crits["DiceLoss"](x, labels)
    for rnd in range(2):
        for v in values:
            assert lib.ptb_set_tunable(key, v) == 0
            for name, crit in crits.items():
                with torch.no_grad():
                    tf = timeit(lambda: crit(x, labels))
                    val = float(crit(x, labels))
                xg = x.clone().requires_grad_(True)

                def fb():
                    xg.grad = None
                    crit(xg, labels).backward()

                tb = timeit(fb, 15)
                print(f"round {rnd} tunable {key}={v}  {name:28s} fwd {tf * 1e3:7.1f} us ({fwd_bytes / tf / 1e9 / 8000 * 100:5.1f} %)   fwd+bwd {tb * 1e3:7.1f} us   value {val:.9f}")


main()
