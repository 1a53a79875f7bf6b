#!/usr/bin/env python3
"""BASELINE configs[3]: [32,16,512,512] fp32 logits + int64 labels -- BinaryFocal / Dice / Jaccard / CE-focal / Lovasz.

Times forward and forward+backward of the HIP losses (HIP events) and reports GB/s of algorithmic bytes
(forward: logits 536 870 912 B + labels 67 108 864 B = 603 979 776 B; backward adds one more read + the gradient write).
For context the same math as eager torch ops on the same GPU (what the reference's code does) is timed too."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_toolbelt_amd import losses as L  # noqa: E402


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def eager_focal(x, labels):
    t = torch.nn.functional.one_hot(labels, x.size(1)).moveaxis(-1, 1).float()
    p = torch.sigmoid(x)
    ce = torch.nn.functional.binary_cross_entropy_with_logits(x, t, reduction="none")
    pt = p * t + (1 - p) * (1 - t)
    return ((1 - pt).pow(2) * ce).mean()


def eager_dice(x, labels):
    p = x.log_softmax(1).exp().flatten(2)
    t = torch.nn.functional.one_hot(labels.flatten(1), x.size(1)).permute(0, 2, 1).float()
    inter, card = (p * t).sum((0, 2)), (p + t).sum((0, 2))
    loss = 1 - 2 * inter / card.clamp_min(1e-7)
    return (loss * (t.sum((0, 2)) > 0)).mean()


def main():
    dev = torch.device("cuda:0")
    B, C, H, W = 32, 16, 512, 512
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn((B, C, H, W), device=dev, generator=g)
    labels = torch.randint(0, C, (B, H, W), device=dev, generator=g)
    fwd_bytes = x.numel() * 4 + labels.numel() * 8
    bwd_bytes = 2 * fwd_bytes - labels.numel() * 8 * 0 + x.numel() * 4
    crits = {
        "BinaryFocalLoss": L.BinaryFocalLoss(),
        "DiceLoss(multiclass)": L.DiceLoss("multiclass"),
        "JaccardLoss(multiclass)": L.JaccardLoss("multiclass"),
        "CrossEntropyFocalLoss": L.CrossEntropyFocalLoss(),
        "FocalDiceJaccardLoss (fused)": L.FocalDiceJaccardLoss("multiclass"),
    }
    print(f"{'loss':34s} {'fwd ms':>8s} {'fwd GB/s':>9s} {'fwd+bwd ms':>11s} {'GB/s':>8s}")
    for name, crit in crits.items():
        with torch.no_grad():
            tf = timeit(lambda: crit(x, labels))
        xg = x.clone().requires_grad_(True)

        def fb():
            xg.grad = None
            crit(xg, labels).backward()

        tb = timeit(fb)
        print(f"{name:34s} {tf:8.3f} {fwd_bytes / tf / 1e6:9.1f} {tb:11.3f} {bwd_bytes / tb / 1e6:8.1f}")
    # SURVEY 8f-3 losses: dense float targets of the logits' shape (2 reads forward; +1 read, +1 write backward),
    # SoftCrossEntropy on int64 labels
    dense = (torch.rand((B, C, H, W), device=dev, generator=g) < 0.3).float()
    dense_fwd = 2 * x.numel() * 4
    extra = {
        "SoftBCEWithLogitsLoss(smooth 0.1)": (L.SoftBCEWithLogitsLoss(smooth_factor=0.1), dense, dense_fwd),
        "BalancedBCEWithLogitsLoss": (L.BalancedBCEWithLogitsLoss(), dense, dense_fwd),
        "QualityFocalLoss": (L.QualityFocalLoss(), dense, dense_fwd),
        "WingLoss": (L.WingLoss(), dense, dense_fwd),
        "LogCoshLoss": (L.LogCoshLoss(), dense, dense_fwd),
        "SoftCrossEntropyLoss(smooth 0.1)": (L.SoftCrossEntropyLoss(smooth_factor=0.1), labels, fwd_bytes),
    }
    for name, (crit, tgt, fb_) in extra.items():
        with torch.no_grad():
            tf = timeit(lambda: crit(x, tgt))
        xg = x.clone().requires_grad_(True)

        def fb2():
            xg.grad = None
            crit(xg, tgt).backward()

        tb = timeit(fb2)
        print(f"{name:34s} {tf:8.3f} {fb_ / tf / 1e6:9.1f} {tb:11.3f} {(2 * fb_ + x.numel() * 4) / tb / 1e6:8.1f}")
    with torch.no_grad():
        tf = timeit(lambda: torch.nn.functional.binary_cross_entropy_with_logits(x, dense), reps=3)
        print(f"{'torch F.bce_with_logits (eager)':34s} {tf:8.3f} {dense_fwd / tf / 1e6:9.1f}")
        tf = timeit(lambda: torch.nn.functional.cross_entropy(x, labels, label_smoothing=0.1), reps=3)
        print(f"{'torch F.cross_entropy ls=0.1 (eager)':34s} {tf:8.3f} {fwd_bytes / tf / 1e6:9.1f}")
    for name, fn in (("eager torch focal (reference math)", eager_focal), ("eager torch dice (reference math)", eager_dice)):
        with torch.no_grad():
            tf = timeit(lambda: fn(x, labels), reps=3)
        print(f"{name:34s} {tf:8.3f} {fwd_bytes / tf / 1e6:9.1f}")
    # CPU reference point: the numpy oracle (one core) on a 1/16 sample of the batch, extrapolated
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import losses_oracle as LO
    xs, ls = x[:2].cpu().numpy(), labels[:2].cpu().numpy()
    for name, fn in (("numpy oracle BinaryFocal (1 core)", lambda: LO.binary_focal_loss(xs, ls)),
                     ("numpy oracle Dice multiclass (1 core)", lambda: LO.dice_loss(xs, ls, "multiclass"))):
        t0 = time.perf_counter()
        fn()
        dt = (time.perf_counter() - t0) * 16 * 1e3
        print(f"{name:34s} {dt:8.1f}   (2 of 32 images timed, x16)")
    probs = torch.softmax(x[:4], 1)
    with torch.no_grad():
        t = timeit(lambda: L.LovaszLoss()(probs, labels[:4]), reps=5)
    print(f"{'LovaszLoss [4,16,512,512]':34s} {t:8.3f}")
    with torch.no_grad():
        t = timeit(lambda: L.BinaryLovaszLoss()(x[:4, 0].contiguous(), (labels[:4] == 1).float()), reps=5)
    print(f"{'BinaryLovaszLoss [4,512,512]':34s} {t:8.3f}")


if __name__ == "__main__":
    main()
