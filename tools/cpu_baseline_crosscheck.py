#!/usr/bin/env python3
"""Build container only: time the UNMODIFIED reference (/root/reference behind oracle/ref_shim) on the headline loop next to
oracle/torch_chain.py (what bench.py's cpu_baseline leg times on the GPU box, where the reference does not exist).
    python tools/cpu_baseline_crosscheck.py            -> one line per implementation, same host, same threads"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "ref_shim"))
sys.path.insert(0, os.environ.get("PTB_REFERENCE", "/root/reference"))
import torch  # noqa: E402

from oracle import torch_chain as TC  # noqa: E402
from pytorch_toolbelt.inference import tiles as rt  # noqa: E402
from pytorch_toolbelt.inference import tta as rtta  # noqa: E402

cores, logical, model = TC.host_description()
torch.set_num_threads(cores)
s = rt.ImageSlicer((5000, 5000, 3), tile_size=(512, 512), tile_step=(256, 256), weight="pyramid")
sample = torch.randn((64, 4, 512, 512), generator=torch.Generator().manual_seed(0))


def run(deaug, make):
    m = make()
    deaug(sample)
    t0 = time.perf_counter()
    for b0 in range(0, 361, 8):
        nb = min(8, 361 - b0)
        x = sample if nb == 8 else sample.view(8, 8, 4, 512, 512)[:, :nb].reshape(8 * nb, 4, 512, 512)
        m.integrate_batch(deaug(x), s.crops[b0:b0 + nb])
    m.merge()
    return time.perf_counter() - t0


for _ in range(2):
    t_ref = run(lambda y: rtta.d4_image_deaugment(y, reduction="mean"), lambda: rt.TileMerger(s.target_shape, 4, s.weight))
    t_port = run(lambda y: TC.image_deaugment(y, "d4", "mean"), lambda: TC.Merger(s.target_shape, 4, s.weight))
    print(f"{model}, {cores} threads: unmodified reference {t_ref:.2f} s/image = {25.0 / t_ref:.2f} MP/s; oracle/torch_chain.py {t_port:.2f} s/image = {25.0 / t_port:.2f} MP/s")
