#!/usr/bin/env python3
"""What this box gives a plain copy (537 MB read + 537 MB written, torch's own kernel and hipMemcpyAsync device-to-device): the ceiling a
loss backward -- logits + labels in, gradient out -- can be measured against.    python tools/copy_ceiling.py"""
import torch

dev = torch.device("cuda:0")
x = torch.randn((32, 16, 512, 512), device=dev)
y = torch.empty_like(x)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


nbytes = 2 * x.numel() * 4
for name, fn in (("y.copy_(x) (torch elementwise kernel)", lambda: y.copy_(x)), ("torch.add(x, 1.0, out=y)", lambda: torch.add(x, 1.0, out=y)),
                 ("x.sum() (read only, 537 MB)", lambda: x.sum())):
    t = timeit(fn)
    b = nbytes if "read only" not in name else nbytes // 2
    print(f"{name:45s} {t * 1e6:8.1f} us  {b / t / 1e9:7.1f} GB/s")
