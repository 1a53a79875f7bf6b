#!/usr/bin/env python3
"""Run the cfg4 region losses a few times (for rocprofv3 --kernel-trace --stats): Dice, fused focal + Dice + Jaccard, forward only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_toolbelt_amd import losses as L  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn((32, 16, 512, 512), device=dev, generator=g)
labels = torch.randint(0, 16, (32, 512, 512), device=dev, generator=g)
which = sys.argv[1] if len(sys.argv) > 1 else "fused"
crit = {"dice": L.DiceLoss("multiclass"), "fused": L.FocalDiceJaccardLoss("multiclass")}[which]
with torch.no_grad():
    for _ in range(40):
        crit(x, labels)
torch.cuda.synchronize()
