import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from torch import nn
sys.path.insert(0, "/root/repo/tools")
from e2e_bench import DummyUNet
dev = torch.device("cuda:0")
for width in (32, 16, 8):
    torch.manual_seed(0)
    model = DummyUNet(width=width).eval().to(dev)
    x = torch.rand((64, 3, 512, 512), device=dev)
    for name, ctx in (("fp32", torch.autocast("cuda", enabled=False)), ("bf16", torch.autocast("cuda", dtype=torch.bfloat16))):
        with torch.no_grad(), ctx:
            t0 = time.perf_counter()
            y = model(x); torch.cuda.synchronize()
            first = time.perf_counter() - t0
            t0 = time.perf_counter()
            for _ in range(3):
                y = model(x)
            torch.cuda.synchronize()
            per = (time.perf_counter() - t0) / 3
        print(f"width {width} {name}: first call {first:.2f} s, then {per * 1e3:.1f} ms per 64-view batch -> {per * 46:.2f} s per 5000x5000 image; out dtype {y.dtype}; peak mem {torch.cuda.max_memory_allocated() / 1e9:.1f} GB", flush=True)
