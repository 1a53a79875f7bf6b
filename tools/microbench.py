#!/usr/bin/env python3
"""Kernel-level A/B timings on one MI355X (HIP events, rotating buffers larger than the 256 MiB Infinity Cache).

    python tools/microbench.py [--reps 20]

Prints GB/s of algorithmic bytes for: device copy, the view kernels with identity / row-preserving / D4 view sets,
the fused accumulate, and merge -- for each chunk-rows variant.  Used to locate where the fused d4 kernel loses
bandwidth (transposes? RMW? launch structure?)."""
import argparse
import os
import sys

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


def nonlinear():
    """Non-linear TTA reductions through the view kernels (probability inputs): d4 de-augment and the fused merge."""
    dev = torch.device("cuda:0")
    B, C, T = 8, 4, 512
    bufs = [torch.rand((8 * B, C, T, T), device=dev) * 0.9 + 0.05 for _ in range(6)]
    nbytes = bufs[0].numel() * 4
    slicer = ImageSlicer((5000, 5000, 3), T, 256, weight="pyramid")
    merger = TileMerger(slicer.target_shape, C, slicer.weight, device=dev)
    crops_row = slicer.crops[:8]
    for red in ("mean", "gmean", "hmean", "logodd", "log1p", "harmonic1p"):
        code = V.REDUCTION_CODES[red]
        t = timeit(lambda i: V._raw_deaug_reduce(bufs[i], list(DEAUGMENT_VIEWS["d4"]), code), 20, 6)
        print(f"{'d4 de-augment reduction=%s (read bytes)' % red:70s} {nbytes / t / 1e9:9.1f} GB/s   {t * 1e6:9.1f} us")
        t = timeit(lambda i: merger.integrate_batch_deaugment(bufs[i], crops_row, group="d4", reduction=red), 20, 6)
        print(f"{'fused d4 accumulate reduction=%s (read bytes)' % red:70s} {nbytes / t / 1e9:9.1f} GB/s   {t * 1e6:9.1f} us")


def half():
    """Half-precision model outputs read natively (no fp32 copy): fused d4 accumulate and d4 de-augment, 8 tiles."""
    dev = torch.device("cuda:0")
    B, C, T = 8, 4, 512
    slicer = ImageSlicer((5000, 5000, 3), T, 256, weight="pyramid")
    merger = TileMerger(slicer.target_shape, C, slicer.weight, device=dev)
    crops_row = slicer.crops[:8]
    for dt in (torch.float32, torch.float16, torch.bfloat16):
        bufs = [torch.randn((8 * B, C, T, T), device=dev).to(dt) for _ in range(6)]
        nbytes = bufs[0].numel() * bufs[0].element_size()
        t = timeit(lambda i: merger.integrate_batch_deaugment(bufs[i], crops_row, group="d4"), 20, 6)
        print(f"{'fused d4 accumulate, 8 tiles, input %s (read bytes)' % str(dt).replace('torch.', ''):70s} {nbytes / t / 1e9:9.1f} GB/s   {t * 1e6:9.1f} us")
        t = timeit(lambda i: V._raw_deaug_reduce(bufs[i], list(DEAUGMENT_VIEWS["d4"]), N.RED_MEAN), 20, 6)
        print(f"{'d4 de-augment mean, input %s (read bytes)' % str(dt).replace('torch.', ''):70s} {nbytes / t / 1e9:9.1f} GB/s   {t * 1e6:9.1f} us")
        if dt != torch.float32:
            t = timeit(lambda i: merger.integrate_batch_deaugment(bufs[i].float(), crops_row, group="d4"), 20, 6)
            print(f"{'  the same through a .float() copy first (reference route)':70s} {'':9s}        {t * 1e6:9.1f} us")


def edges():
    """Device-side loop edges at the cfg2 geometry: split_device of 8 tiles (d4, affine) and merge_crop."""
    dev = torch.device("cuda:0")
    slicer = ImageSlicer((5000, 5000, 3), 512, 256, weight="pyramid")
    img = torch.randint(0, 256, (5000, 5000, 3), dtype=torch.uint8, device=dev)
    sc, bi = [1 / 255.0] * 3, [0.0] * 3
    starts = [0, 40, 96, 176, 240, 300]
    for aug, V_ in ((None, 1), ("d4", 8)):
        for affine in (False, True):
            t = timeit(lambda i: slicer.split_device(img, slice(starts[i], starts[i] + 8), augment=aug, scale=sc if affine else None,
                                                     bias=bi if affine else None), 20, len(starts))
            by = 8 * 3 * 512 * 512 * (1 + 4 * V_)
            print(f"{'split_device 8 tiles augment=%s affine=%d (r u8 + w fp32 bytes)' % (aug, affine):70s} {by / t / 1e9:9.1f} GB/s   {t * 1e6:9.1f} us")
    merger = TileMerger(slicer.target_shape, 4, slicer.weight, device=dev)
    merger.image.normal_()
    merger.norm_mask.fill_(1.5)
    rd = 5 * 5000 * 5000 * 4
    for name, kw, wr in (("f32 hwc", dict(), 4 * 4), ("f32 chw", dict(layout="chw"), 4 * 4), ("u8 hwc", dict(dtype=torch.uint8), 4),
                         ("argmax u8", dict(argmax=True, dtype=torch.uint8), 1), ("argmax i64", dict(argmax=True, dtype=torch.int64), 8)):
        t = timeit(lambda i: merger.merge_crop(slicer, **kw), 20, 1)
        print(f"{'merge_crop 5000x5000 C=4 -> ' + name + ' (r+w bytes)':70s} {(rd + 5000 * 5000 * wr) / t / 1e9:9.1f} GB/s   {t * 1e6:9.1f} us")
    t = timeit(lambda i: merger.merge(), 20, 1)
    print(f"{'merge (padded fp32 CHW, for comparison)':70s} {(9 * 5120 * 5120 * 4) / t / 1e9:9.1f} GB/s   {t * 1e6:9.1f} us")


if len(sys.argv) > 1 and sys.argv[1] == "cfg5":
    cfg5()
    sys.exit(0)
def ensemble():
    """Ensembler over T = 4 model outputs [8, 16, 512, 512]: fused activation + reduce vs torch ops."""
    from pytorch_toolbelt_amd.inference import ensembling as E

    dev = torch.device("cuda:0")
    T = 4
    sets = [[torch.randn((8, 16, 512, 512), device=dev) for _ in range(T)] for _ in range(3)]
    by = (T + 1) * sets[0][0].numel() * 4
    for act, name in ((0, "none"), (1, "sigmoid"), (2, "softmax")):
        for code, red in ((N.RED_MEAN, "mean"), (N.RED_GMEAN, "gmean")):
            if act == 0 and red == "gmean":
                continue
            t = timeit(lambda i: E._ensemble_native(sets[i], code, act, 0.5, 1), 10, 3)
            print(f"{'ensemble T=4 [8,16,512,512] act=%s reduction=%s (T reads + 1 write)' % (name, red):70s} {by / t / 1e9:9.1f} GB/s   {t * 1e6:9.1f} us")
    t = timeit(lambda i: torch.stack([x.mul(0.5).softmax(dim=1) for x in sets[i]]).log().mean(0).exp(), 5, 3)
    print(f"{'same (softmax, gmean) as the reference op chain in eager torch':70s} {by / t / 1e9:9.1f} GB/s   {t * 1e6:9.1f} us")


if len(sys.argv) > 1 and sys.argv[1] == "half":
    half()
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "nonlinear":
    nonlinear()
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "ensemble":
    ensemble()
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "edges":
    edges()
    sys.exit(0)

if __name__ == "__main__":
    main()
