"""Workload for rocprofv3 --kernel-trace --stats: every loss kernel at the BASELINE cfg4 shape ([32,16,512,512] logits),
forward and forward+backward, plus the ensemble reduce and the device-side loop edges (summary -> profiles/)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_toolbelt_amd import losses as L  # noqa: E402
from pytorch_toolbelt_amd.inference import ensembling as E  # noqa: E402
from pytorch_toolbelt_amd.inference.tiles import ImageSlicer, TileMerger  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn((32, 16, 512, 512), device=dev, generator=g)
labels = torch.randint(0, 16, (32, 512, 512), device=dev, generator=g)
dense = (torch.rand((32, 16, 512, 512), device=dev, generator=g) < 0.3).float()
xb = torch.randn((32, 1, 512, 512), device=dev, generator=g)
tb = (torch.rand((32, 1, 512, 512), device=dev, generator=g) < 0.3).float()
label_losses = [L.BinaryFocalLoss(), L.DiceLoss("multiclass"), L.JaccardLoss("multiclass"), L.CrossEntropyFocalLoss(),
                L.FocalDiceJaccardLoss("multiclass"), L.SoftCrossEntropyLoss(smooth_factor=0.1)]
dense_losses = [L.SoftBCEWithLogitsLoss(smooth_factor=0.1), L.BalancedBCEWithLogitsLoss(), L.QualityFocalLoss(), L.WingLoss(), L.LogCoshLoss()]
slicer = ImageSlicer((5000, 5000, 3), 512, 256, weight="pyramid")
img = torch.randint(0, 256, (5000, 5000, 3), dtype=torch.uint8, device=dev)
merger = TileMerger(slicer.target_shape, 4, slicer.weight, device=dev)
merger.image.normal_()
merger.norm_mask.fill_(1.5)
models = [torch.randn((8, 16, 512, 512), device=dev) for _ in range(4)]
for _ in range(3):
    for crit, tgt in [(c, labels) for c in label_losses] + [(c, dense) for c in dense_losses] + [(L.BinaryBiTemperedLogisticLoss(0.8, 1.2), tb)]:
        inp = xb if tgt is tb else x
        with torch.no_grad():
            crit(inp, tgt)
        xg = inp.clone().requires_grad_(True)
        crit(xg, tgt).backward()
    probs = torch.softmax(x[:4], 1)
    with torch.no_grad():
        L.LovaszLoss()(probs, labels[:4])
        L.BinaryLovaszLoss()(x[:4, 0].contiguous(), (labels[:4] == 1).float())
        for act in (0, 1, 2):
            E._ensemble_native(models, 2, act, 0.5, 1)      # gmean
        slicer.split_device(img, slice(0, 8), augment="d4", scale=[1 / 255.0] * 3, bias=[0.0] * 3)
        merger.merge_crop(slicer, dtype=torch.uint8)
        merger.merge_crop(slicer, argmax=True, dtype=torch.uint8)
torch.cuda.synchronize()
