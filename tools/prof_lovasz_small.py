#!/usr/bin/env python3
"""LovaszLoss [4,16,512,512]: 6 forward + backward calls (a short run for rocprofv3 --pmc passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_toolbelt_amd import losses as L  # noqa: E402

dev = torch.device("cuda:0")
probs = torch.softmax(torch.randn((4, 16, 512, 512), device=dev) * 3, 1).requires_grad_(True)
lab = torch.randint(0, 16, (4, 512, 512), device=dev)
for _ in range(6):
    probs.grad = None
    L.LovaszLoss()(probs, lab).backward()
torch.cuda.synchronize()
