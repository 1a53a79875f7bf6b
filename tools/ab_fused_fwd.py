"""A/B of the fused focal + Dice + Jaccard forward at BASELINE cfg4 ([32,16,512,512]): ptb_set_tunable key 8 (prefetch of the next
pixel group) x key 4 (workgroups per launch).  Prints ms per criterion call (kernel + epilogue + finalize) and the loss value."""
import sys
import torch
sys.path.insert(0, "/root/repo")
from pytorch_toolbelt_amd import losses as L, _native as N

dev = torch.device("cuda:0")
x = torch.randn((32, 16, 512, 512), device=dev)
lab = torch.randint(0, 16, (32, 512, 512), device=dev)
crit = L.FocalDiceJaccardLoss("multiclass")


def t(n=40):
    with torch.no_grad():
        for _ in range(5):
            crit(x, lab)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            crit(x, lab)
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


lib = N.load()
for rep in range(2):
    for pf in (0, 1):
        for cap in (0, 512, 768, 1024, 1536, 3072):
            assert lib.ptb_set_tunable(8, pf) == 0 and lib.ptb_set_tunable(4, cap) == 0
            print(f"prefetch {pf} grid cap {cap:5d}: {t():.4f} ms  loss {float(crit(x, lab)):.7f}")
lib.ptb_set_tunable(8, 0); lib.ptb_set_tunable(4, 0)
