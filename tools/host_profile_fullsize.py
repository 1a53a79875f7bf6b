import sys, time, torch
sys.path.insert(0, "/root/repo")
from pytorch_toolbelt_amd.inference import tta
from pytorch_toolbelt_amd.inference.tiles import ImageSlicer, TileMerger
dev = torch.device("cuda:0")
slicer = ImageSlicer((5000, 5000, 3), 512, 256, weight="pyramid")
crops = slicer.crops; n = len(crops)
outs = [torch.randn((8 * min(8, n - b0), 4, 512, 512), device=dev) for b0 in range(0, n, 8)]
pc = [crops[b0:b0 + 8] for b0 in range(0, n, 8)]
def lit(fresh):
    global m
    if fresh: m = TileMerger(slicer.target_shape, 4, slicer.weight, device=dev)
    else: m.reset()
    t0 = time.perf_counter()
    for t, c in zip(outs, pc):
        m.integrate_batch(tta.d4_image_deaugment(t), c)
    t1 = time.perf_counter()
    r = m.merge()
    return t1 - t0
def ext():
    m2.reset()
    t0 = time.perf_counter()
    for t, c in zip(outs, pc):
        m2.integrate_batch_deaugment(t, c, group="d4", reduction="mean")
    t1 = time.perf_counter()
    m2.merge()
    return t1 - t0
m = TileMerger(slicer.target_shape, 4, slicer.weight, device=dev)
m2 = TileMerger(slicer.target_shape, 4, slicer.weight, device=dev, crops=crops, defer=True)
for name, fn in (("literal, new merger", lambda: lit(True)), ("literal, reset", lambda: lit(False)), ("explicit deferred", ext)):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); host = 0.0
    for _ in range(20): host += fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:22s} GPU {e0.elapsed_time(e1)/20:.3f} ms per image; host loop {host/20*1e3:.3f} ms per image = {host/20/46*1e6:.1f} us per call; mode {m.mode}")
def plain(**kw):
    mm = TileMerger(slicer.target_shape, 4, slicer.weight, device=dev, **kw)
    for t, c in zip(preds, pc): mm.integrate_batch(t, c)
    return mm.merge()
preds = [torch.randn((min(8, n - b0), 4, 512, 512), device=dev) for b0 in range(0, n, 8)]
for kw in ({"auto_plan": False}, {}):
    for _ in range(4): plain(**kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): plain(**kw)
    e1.record(); torch.cuda.synchronize()
    print(f"plain loop {kw}: {e0.elapsed_time(e1)/20:.3f} ms per image")
