import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_toolbelt_amd import losses as L, _native as N
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn((32, 16, 512, 512), device=dev, generator=g)
labels = torch.randint(0, 16, (32, 512, 512), device=dev, generator=g)
crit = L.FocalDiceJaccardLoss("multiclass")
xg = x.clone().requires_grad_(True)
def fb():
    xg.grad = None
    crit(xg, labels).backward()
def t(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
lib = N.load()
ref = None
for rep in range(3):
    for nt in (0, 1):
        lib.ptb_set_tunable(16, nt)
        ms = t(fb)
        gsum = float(xg.grad.abs().sum())
        print(f"nt stores {nt}: fwd+bwd {ms*1e3:.1f} us  |grad| {gsum:.6e}")
