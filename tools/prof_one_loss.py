"""Workload for a rocprofv3 PMC pass over ONE loss at the cfg4 shape: python tools/prof_one_loss.py dice|focal|fused|cefocal [bwd]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_toolbelt_amd import losses as L  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn((32, 16, 512, 512), device=dev, generator=g)
labels = torch.randint(0, 16, (32, 512, 512), device=dev, generator=g)
crit = {"dice": L.DiceLoss("multiclass"), "focal": L.BinaryFocalLoss(), "fused": L.FocalDiceJaccardLoss("multiclass"),
        "cefocal": L.CrossEntropyFocalLoss()}[sys.argv[1]]
bwd = len(sys.argv) > 2 and sys.argv[2] == "bwd"
if os.environ.get("PTB_TUNABLE4"):
    from pytorch_toolbelt_amd import _native as N
    assert N.load().ptb_set_tunable(4, int(os.environ["PTB_TUNABLE4"])) == 0
for _ in range(4):
    if bwd:
        xg = x.clone().requires_grad_(True)
        crit(xg, labels).backward()
    else:
        with torch.no_grad():
            crit(x, labels)
torch.cuda.synchronize()
