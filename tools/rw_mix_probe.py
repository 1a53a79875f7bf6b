#!/usr/bin/env python3
"""What a stream with a given share of WRITES reaches on this box (torch's own elementwise kernels over 420 MB fp32 operands: copy = 1 read :
1 write, add = 2 : 1, addcmul = 3 : 1) next to the library's read-only probe -- the ceiling the loop without TTA (3.6 reads : 1 write) and the
cfg4 backward (1 : 1) should be priced against, since HBM pays for every read/write turnaround."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")
n = 4 * 5120 * 5120
a, b, c, d = (torch.randn(n, device=dev) for _ in range(4))
out = torch.empty(n, device=dev)


def timed(fn, reps=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, fn, ops in (("copy_      (1 read : 1 write)", lambda: out.copy_(a), 2), ("add        (2 reads : 1 write)", lambda: torch.add(a, b, out=out), 3),
                      ("addcmul    (3 reads : 1 write)", lambda: torch.addcmul(a, b, c, out=out), 4), ("sum        (1 read : 0 writes)", lambda: a.sum(), 1)):
    ms = timed(fn)
    gb = ops * n * 4 / 1e9
    print(f"{name}: {ms * 1e3:7.1f} us for {gb:.2f} GB = {gb / ms * 1e3:.0f} GB/s = {gb / ms * 1e3 / 80:.1f} % of 8 TB/s")
