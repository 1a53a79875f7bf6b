import os, sys, torch
sys.path.insert(0, "/root/repo")
from pytorch_toolbelt_amd import losses as L, _native as N
dev = torch.device("cuda:0")
pads = [torch.empty(int(os.environ.get("PAD_GB", "0")) << 30, dtype=torch.uint8, device=dev)] if os.environ.get("PAD_GB") else []
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn((32, 16, 512, 512), device=dev, generator=g)
labels = torch.randint(0, 16, (32, 512, 512), device=dev, generator=g)
crit = L.FocalDiceJaccardLoss("multiclass")
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
lib = N.load()
caps = [int(c) for c in os.environ.get("CAPS", "0").split(",")]
for rep in range(2):
    for cap in caps:
        lib.ptb_set_tunable(13, cap if cap else 512)
        for pk in (2, 1, 0):
            lib.ptb_set_tunable(12, pk)
            with torch.no_grad():
                v = float(crit(x, labels)); ms = t(lambda: crit(x, labels))
            print(f"grid cap {cap:5d} pk={pk} loss={v:.9f} fwd {ms*1e3:.1f} us")
