#!/usr/bin/env python3
"""Interleaved A/B of the loss kernels under a tunable: python tools/ab_losses.py <key> <v1> <v2> ..."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_toolbelt_amd import _native as N, losses as L
key, vals = int(sys.argv[1]), [int(v) for v in sys.argv[2:]]
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn((32, 16, 512, 512), device=dev, generator=g)
labels = torch.randint(0, 16, (32, 512, 512), device=dev, generator=g)
crits = {"focal": L.BinaryFocalLoss(), "dice": L.DiceLoss("multiclass"), "cefocal": L.CrossEntropyFocalLoss()}
os.environ["PTB_SKIP_LABEL_CHECK"] = "1"
def t(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
for rnd in range(2):
    for v in vals:
        assert N.load().ptb_set_tunable(key, v) == 0
        row = []
        for name, c in crits.items():
            with torch.no_grad():
                f = t(lambda: c(x, labels))
            xg = x.clone().requires_grad_(True)
            def fb():
                xg.grad = None; c(xg, labels).backward()
            row.append(f"{name} fwd {f:.3f} f+b {t(fb):.3f}")
        print(f"key {key} = {v}: " + " | ".join(row))
