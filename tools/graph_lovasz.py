#!/usr/bin/env python3
"""Does the launch chain of LovaszLoss (17 launches without a gradient) wait for launches?  The forward under torch.no_grad() eagerly
and replayed from a captured HIP graph, same process."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_toolbelt_amd import losses as L  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
probs = torch.softmax(torch.randn((4, 16, 512, 512), device=dev) * 3, 1)
lab = torch.randint(0, 16, (4, 512, 512), device=dev)
loss = L.LovaszLoss()


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


with torch.no_grad():
    eager = timeit(lambda: loss(probs, lab))
    ref = float(loss(probs, lab))
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            loss(probs, lab)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = loss(probs, lab)
    g.replay()
    torch.cuda.synchronize()
    replay = timeit(g.replay)
    print(f"eager {eager:.1f} us | graph replay {replay:.1f} us | loss eager {ref:.9f} graph {float(out):.9f}")
