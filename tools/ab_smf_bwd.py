"""A/B of the softmax focal (CrossEntropyFocalLoss) backward at BASELINE cfg4: tunable 7 = 0 (two transcendental passes), 2, 4
(per-class terms kept in registers, 2 / 4 pixels per lane).  Prints forward+backward ms and a checksum of the gradient."""
import sys
import torch
sys.path.insert(0, "/root/repo")
from pytorch_toolbelt_amd import losses as L, _native as N

dev = torch.device("cuda:0")
x = torch.randn((32, 16, 512, 512), device=dev)
lab = torch.randint(0, 16, (32, 512, 512), device=dev)
xg = x.clone().requires_grad_(True)
for name, ce in (("gamma2", L.CrossEntropyFocalLoss()), ("gamma1.5", L.CrossEntropyFocalLoss(gamma=1.5))):
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
    ref = None
    for rep in range(2):
        for v in (0, 2, 4):
            assert N.load().ptb_set_tunable(7, v) == 0
            ms = tb()
            g = xg.grad.clone()
            if v == 0: ref = g
            print(name, "stash", v, round(ms, 4), "ms fwd+bwd; max|grad - two-pass| =", float((g - ref).abs().max()), "max|grad| =", float(ref.abs().max()))
N.load().ptb_set_tunable(7, 4)
