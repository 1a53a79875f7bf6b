#!/usr/bin/env python3
"""Workload for the instruction-level comparison of the cfg4 fused forward with a plain read stream over the SAME bytes (VERDICT round 5,
item 6): N x FocalDiceJaccardLoss forward on [32,16,512,512] logits + int64 labels (seg_focal_pk_kernel), then N x ptb_read_probe_multi
over the same two buffers (read_probe_multi_kernel).  Run under rocprofv3 --pmc (tools/pmc_cmd.sh); prints HIP-event times itself."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_toolbelt_amd import _native as N  # noqa: E402
from pytorch_toolbelt_amd import losses as L  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn((32, 16, 512, 512), device=dev, generator=g)
labels = torch.randint(0, 16, (32, 512, 512), device=dev, generator=g)
crit = L.FocalDiceJaccardLoss("multiclass")
lib = N.load()
sink = torch.zeros(1 << 16, device=dev)
ptrs = (ctypes.c_void_p * 2)(x.data_ptr(), labels.data_ptr())
sizes = (ctypes.c_int64 * 2)(x.numel() * 4, labels.numel() * 8)
NREP = int(os.environ.get("PTB_PROF_N", "6"))
WGS = int(os.environ.get("PTB_PROBE_WGS", "2048"))


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(NREP):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / NREP * 1e3


def fwd():
    with torch.no_grad():
        crit(x, labels)


def probe():
    got = lib.ptb_read_probe_multi(ptrs, sizes, 2, sink.data_ptr(), WGS, N.stream_ptr(dev))
    assert got > 0


nbytes = x.numel() * 4 + labels.numel() * 8
a, b = timed(fwd), timed(probe)
print(f"fused forward {a:.1f} us = {nbytes / a / 1e6:.2f} TB/s; read probe over the same {nbytes / 1e6:.0f} MB ({WGS} workgroups) {b:.1f} us = {nbytes / b / 1e6:.2f} TB/s")
