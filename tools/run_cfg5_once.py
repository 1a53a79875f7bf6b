#!/usr/bin/env python3
"""One-pass cfg5 merge, a few launches (for rocprofv3 --pmc runs)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_toolbelt_amd.inference import tta  # noqa: E402

dev = torch.device("cuda:0")
N_, C, V = 4096, 4, 2
offs = [-N_ // 4, 0, N_ // 4]
ys = [torch.rand((V, C, N_ + o, N_ + o), device=dev) * 0.9 + 0.05 for o in offs]
inner, outer = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ("mean", "mean")
for _ in range(5):
    tta.ms_flips_image_deaugment(ys, offs, group="fliplr", inner_reduction=inner, reduction=outer, align_corners=False)
torch.cuda.synchronize()
