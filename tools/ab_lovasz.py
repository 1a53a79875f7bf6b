#!/usr/bin/env python3
"""LovaszLoss [4,16,512,512]: forward (no_grad), forward, forward + backward under ptb_set_tunable(KEY, v) for v in VALUES.

    python tools/ab_lovasz.py [KEY VALUES...]      default: 17 0 1  (XCD-contiguous tile order of the radix scatter)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_toolbelt_amd import _native as N  # noqa: E402
from pytorch_toolbelt_amd import losses as L  # noqa: E402

key = int(sys.argv[1]) if len(sys.argv) > 1 else 17
values = [int(v) for v in sys.argv[2:]] or [0, 1]
dev = torch.device("cuda:0")
torch.manual_seed(0)
probs = torch.softmax(torch.randn((4, 16, 512, 512), device=dev) * 3, 1).requires_grad_(True)
lab = torch.randint(0, 16, (4, 512, 512), device=dev)
loss = L.LovaszLoss()


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def fwd_bwd():
    probs.grad = None
    loss(probs, lab).backward()


def no_grad():
    with torch.no_grad():
        loss(probs, lab)


from pytorch_toolbelt_amd.losses import lovasz as LV  # noqa: E402

for rnd in range(3):
    for v in values:
        if key == 0:         # key 0: binned (1) / scattered (0) gradient (a Python switch, not a tunable of the library)
            LV.BINNED_GRADIENT = bool(v)
        elif key == -2:      # key -2: gscale * coef inside the backward kernel, no materialised zero gradient for fg_total (1) / separate launches (0)
            LV.FUSED_TAIL = bool(v)
        elif key == -1:      # key -1: key-only sort (1) / pair sort (0) for the forward without gradient
            LV.KEY_ONLY_FORWARD = bool(v)
        else:
            assert N.load().ptb_set_tunable(key, v) == 0
        a, b, c = timeit(no_grad), timeit(lambda: loss(probs, lab)), timeit(fwd_bwd)
        fwd_bwd()
        print(f"tunable {key} = {v}: no_grad {a:7.1f} us | forward {b:7.1f} us | forward+backward {c:7.1f} us | loss {loss(probs, lab).item():.9f} |grad| {probs.grad.abs().sum().item():.6e}")
