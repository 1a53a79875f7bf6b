#!/usr/bin/env python3
"""LovaszLoss [4,16,512,512] + BinaryLovaszLoss [4,512,512], 20 forward+backward calls each (for rocprofv3 --kernel-trace --stats)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_toolbelt_amd import losses as L  # noqa: E402

dev = torch.device("cuda:0")
probs = torch.softmax(torch.randn((4, 16, 512, 512), device=dev), 1).requires_grad_(True)
lab = torch.randint(0, 16, (4, 512, 512), device=dev)
x = torch.randn((4, 512, 512), device=dev, requires_grad=True)
t = (torch.rand((4, 512, 512), device=dev) < 0.3).float()
BIG_ONLY = os.environ.get("PTB_PROF_BIG_ONLY") == "1"      # only the multi-class case (per-kernel averages of ONE problem size)
for _ in range(20):
    L.LovaszLoss()(probs, lab).backward()
    if not BIG_ONLY:
        L.BinaryLovaszLoss(per_image=True)(x, t).backward()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    L.LovaszLoss()(probs, lab)
e1.record()
torch.cuda.synchronize()
print(f"LovaszLoss forward [4,16,512,512]: {e0.elapsed_time(e1) / 20:.3f} ms")
with torch.no_grad():
    for _ in range(3):
        L.LovaszLoss()(probs, lab)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        L.LovaszLoss()(probs, lab)
    e1.record()
    torch.cuda.synchronize()
print(f"LovaszLoss forward [4,16,512,512] under no_grad (no per-pixel gradient scatter): {e0.elapsed_time(e1) / 20:.3f} ms")
