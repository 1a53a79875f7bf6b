import sys

import torch
sys.path.insert(0, "/root/repo")
from pytorch_toolbelt_amd import losses as L, _native as N
dev = torch.device("cuda:0")
x = torch.randn((32, 16, 512, 512), device=dev)
lab = torch.randint(0, 16, (32, 512, 512), device=dev)
crit = L.FocalDiceJaccardLoss("multiclass")
def t(n=30):
    for _ in range(5): crit(x, lab)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): crit(x, lab)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for rep in range(2):
    for v in (0, 1):
        N.load().ptb_set_tunable(5, v)
        print("pix2" if v else "pix4", round(t(), 4), "ms", float(crit(x, lab)))

ce = L.CrossEntropyFocalLoss()
xg = x.clone().requires_grad_(True)
def tb(n=20):
    for _ in range(3):
        xg.grad = None; ce(xg, lab).backward()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        xg.grad = None; ce(xg, lab).backward()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for rep in range(2):
    for v in (0, 1):
        N.load().ptb_set_tunable(5, v)
        ms = tb()
        print("CE-focal fwd+bwd", "pix2" if v else "pix4", round(ms, 4), "ms", float(xg.grad.abs().sum()))
