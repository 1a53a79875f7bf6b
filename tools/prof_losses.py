import os, sys, torch
sys.path.insert(0, "/root/repo")
from pytorch_toolbelt_amd import losses as L
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn((32, 16, 512, 512), device=dev, generator=g)
labels = torch.randint(0, 16, (32, 512, 512), device=dev, generator=g)
for _ in range(3):
    with torch.no_grad():
        L.BinaryFocalLoss()(x, labels); L.DiceLoss("multiclass")(x, labels); L.CrossEntropyFocalLoss()(x, labels)
    xg = x.clone().requires_grad_(True)
    (L.BinaryFocalLoss()(xg, labels) + L.DiceLoss("multiclass")(xg, labels) + L.CrossEntropyFocalLoss()(xg, labels)).backward()
    probs = torch.softmax(x[:4], 1)
    with torch.no_grad():
        L.LovaszLoss()(probs, labels[:4])
        L.BinaryLovaszLoss()(x[:4, 0].contiguous(), (labels[:4] == 1).float())
torch.cuda.synchronize()
