#!/usr/bin/env python3
"""A/B of a loss-kernel tunable on one box:  python tools/ab_losses.py <tunable key> <value A> <value B>
Times BinaryFocalLoss / CrossEntropyFocalLoss / DiceLoss forward and forward+backward at the cfg4 shape, interleaved."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_toolbelt_amd import _native as N  # noqa: E402
from pytorch_toolbelt_amd import losses as L  # noqa: E402


def timeit(fn, reps=10):
    for _ in range(2):
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
    key, va, vb = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    dev = torch.device("cuda:0")
    B, C, H, W = 32, 16, 512, 512
    x = torch.randn((B, C, H, W), device=dev)
    labels = torch.randint(0, C, (B, H, W), device=dev)
    dense = (torch.rand((B, C, H, W), device=dev) < 0.3).float()
    crits = {"BinaryFocalLoss(labels)": (L.BinaryFocalLoss(), labels), "focal_loss_with_logits(dense)": (lambda a, b: L.focal_loss_with_logits(a, b), dense),
             "BinaryFocalLoss(alpha,gamma=1.5)": (L.BinaryFocalLoss(alpha=0.25, gamma=1.5), labels)}
    lib = N.load()
    for rnd in range(2):
        for v in (va, vb):
            assert lib.ptb_set_tunable(key, v) == 0
            for name, (crit, tgt) in crits.items():
                with torch.no_grad():
                    tf = timeit(lambda: crit(x, tgt))
                xg = x.clone().requires_grad_(True)

                def fb():
                    xg.grad = None
                    crit(xg, tgt).backward()

                tb = timeit(fb)
                print(f"round {rnd} tunable {key}={v}  {name:34s} fwd {tf:7.3f} ms   fwd+bwd {tb:7.3f} ms")


if __name__ == "__main__":
    main()
