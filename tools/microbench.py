#!/usr/bin/env python3
"""Kernel-level A/B timings on one MI355X (HIP events, rotating buffers larger than the 256 MiB Infinity Cache).

    python tools/microbench.py [--reps 20]

Prints GB/s of algorithmic bytes for: device copy, the view kernels with identity / row-preserving / D4 view sets,
the fused accumulate, and merge -- for each chunk-rows variant.  Used to locate where the fused d4 kernel loses
bandwidth (transposes? RMW? launch structure?)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_toolbelt_amd import _native as N  # noqa: E402
from pytorch_toolbelt_amd.inference import _views as V  # noqa: E402
from pytorch_toolbelt_amd.inference.tiles import ImageSlicer, TileMerger  # noqa: E402
from pytorch_toolbelt_amd.inference.tta import DEAUGMENT_VIEWS  # noqa: E402


def timeit(fn, reps, nbuf):
    for i in range(3):
        fn(i % nbuf)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i % nbuf)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = N.load()
    B, C, T = 8, 4, 512
    nbuf = 6
    bufs = [torch.randn((8 * B, C, T, T), device=dev) for _ in range(nbuf)]
    nbytes = bufs[0].numel() * 4
    out = torch.empty_like(bufs[0])
    rows = []

    t = timeit(lambda i: out.copy_(bufs[i]), args.reps, nbuf)
    rows.append(("torch copy 268MB (r+w bytes)", 2 * nbytes / t / 1e9, t))
    t = timeit(lambda i: torch.sum(bufs[i].view(8, -1), dim=0), args.reps, nbuf)
    rows.append(("torch sum over 8 views (read bytes)", nbytes / t / 1e9, t))

    t = timeit(lambda i: bufs[i].sum(), args.reps, nbuf)
    rows.append(("torch full sum (read-only ceiling probe)", nbytes / t / 1e9, t))
    # power-of-two view stride probe: 7 / 9 tiles per batch move the 8 view streams off the 32 MiB spacing
    for nb in (7, 9, 16):
        xs = [torch.randn((8 * nb, C, T, T), device=dev) for _ in range(4)]
        t = timeit(lambda i: V._raw_deaug_reduce(xs[i % 4], list(DEAUGMENT_VIEWS["d4"]), N.RED_MEAN), args.reps, 4)
        rows.append((f"CH=64 deaug_reduce d4, {nb} tiles/batch (view stride {nb * 4} MiB)", xs[0].numel() * 4 / t / 1e9, t))
        del xs

    slicer = ImageSlicer((5000, 5000, 3), T, 256, weight="pyramid")
    merger = TileMerger(slicer.target_shape, C, slicer.weight, device=dev)
    crops_row = slicer.crops[:8]
    crops_sep = slicer.crops[[0, 2, 4, 6, 8, 10, 12, 14]]  # non-overlapping tiles: one tile per cell

    view_sets = {
        "identity x8": [N.IDENT] * 8,
        "row-preserving x8 (id,lr,ud,r180)x2": [N.IDENT, N.FLIPLR, N.FLIPUD, N.ROT180] * 2,
        "d4": list(DEAUGMENT_VIEWS["d4"]),
        "transposing x4 + id x4": [N.TRANSPOSE, N.ROT90_CW, N.ROT90_CCW, N.ANTITRANSPOSE, N.IDENT, N.IDENT, N.IDENT, N.IDENT],
    }
    for ch, nt in ((64, 0), (64, 1), (32, 0), (32, 1), (16, 0), (16, 1)):
        lib.ptb_set_tunable(0, ch)
        lib.ptb_set_tunable(2, nt)
        ch = f"{ch} nt={nt}"
        for name, views in view_sets.items():
            t = timeit(lambda i: V._raw_deaug_reduce(bufs[i], views, N.RED_MEAN), args.reps, nbuf)
            rows.append((f"CH={ch} deaug_reduce {name} -> [8,4,512,512]", nbytes / t / 1e9, t))
        t = timeit(lambda i: merger.integrate_batch_deaugment(bufs[i], crops_row, group="d4"), args.reps, nbuf)
        rows.append((f"CH={ch} fused d4 accumulate, row of 8 (50% overlap)", nbytes / t / 1e9, t))
        t = timeit(lambda i: merger.integrate_batch_deaugment(bufs[i], crops_sep, group="d4"), args.reps, nbuf)
        rows.append((f"CH={ch} fused d4 accumulate, 8 disjoint tiles", nbytes / t / 1e9, t))
        t = timeit(lambda i: merger.integrate_batch(bufs[i][:8], crops_row), args.reps, nbuf)
        rows.append((f"CH={ch} integrate_batch (1 view) row of 8", nbytes / 8 / t / 1e9, t))
    lib.ptb_set_tunable(0, 32)
    for nt in (0, 1):
        lib.ptb_set_tunable(2, nt)
        t = timeit(lambda i: merger.merge(), args.reps, nbuf)
        rows.append((f"merge 4x5120x5120 (r image+norm, w out) nt={nt}", (2 * merger.image.numel() + merger.norm_mask.numel()) * 4 / t / 1e9, t))
    # d4 augment [8,3,512,512] -> [64,3,512,512]: read 25 MB, write 201 MB
    from pytorch_toolbelt_amd.inference import tta
    xa = [torch.randn((8, 3, T, T), device=dev) for _ in range(nbuf)]
    t = timeit(lambda i: tta.d4_image_augment(xa[i]), args.reps, nbuf)
    rows.append(("d4_image_augment [8,3,512,512] -> [64,...] (r+w bytes)", 9 * xa[0].numel() * 4 / t / 1e9, t))
    t = timeit(lambda i: tta.d4_image_deaugment(bufs[i]), args.reps, nbuf)
    rows.append(("d4_image_deaugment [64,4,512,512] -> [8,...] via public API (read bytes)", nbytes / t / 1e9, t))
    t = timeit(lambda i: (merger.image.zero_(), merger.norm_mask.zero_()), args.reps, nbuf)
    rows.append(("zero accumulators (w bytes)", (merger.image.numel() + merger.norm_mask.numel()) * 4 / t / 1e9, t))
    for name, gbs, t in rows:
        print(f"{name:70s} {gbs:9.1f} GB/s   {t * 1e6:9.1f} us")




def cfg5():
    """BASELINE configs[4]: multiscale (3072/4096/5120) de-augment + gmean to 4096x4096, C=4 (after per-scale fliplr)."""
    from pytorch_toolbelt_amd.inference import tta

    dev = torch.device("cuda:0")
    maps = [torch.rand((1, 4, 4096 + o, 4096 + o), device=dev) * 0.9 + 0.05 for o in (-1024, 0, 1024)]
    flips = [torch.rand((2, 4, 4096 + o, 4096 + o), device=dev) * 0.9 + 0.05 for o in (-1024, 0, 1024)]
    alg = sum(m.numel() for m in maps) * 4 + 4 * 4096 * 4096 * 4
    for tiled in (0, 1):
        N.load().ptb_set_tunable(3, tiled)
        t = timeit(lambda i: tta.ms_image_deaugment(maps, [-1024, 0, 1024], reduction="gmean", align_corners=False), 10, 1)
        print(f"{'cfg5 fused ms_image_deaugment gmean (3 scales -> 4096^2, C=4) tiled=%d' % tiled:70s} {alg / t / 1e9:9.1f} GB/s   {t * 1e6:9.1f} us")
    alg2 = sum(m.numel() for m in flips) * 4 + 4 * 4096 * 4096 * 4
    t = timeit(lambda i: tta.ms_image_deaugment([tta.fliplr_image_deaugment(f) for f in flips], [-1024, 0, 1024], reduction="gmean"), 10, 1)
    print(f"{'cfg5 fliplr de-augment per scale + fused ms gmean (SURVEY 8d bytes)':70s} {alg2 / t / 1e9:9.1f} GB/s   {t * 1e6:9.1f} us")


if len(sys.argv) > 1 and sys.argv[1] == "cfg5":
    cfg5()
    sys.exit(0)

if __name__ == "__main__":
    main()
