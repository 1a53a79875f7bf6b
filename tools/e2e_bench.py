#!/usr/bin/env python3
"""Secondary, end-to-end timing of the reference's canonical loop (README.md:196-227 of the reference) on one MI355X:

    uint8 image 5000x5000x3 -> ImageSlicer.split (host) -> batches of 8 tiles -> H2D -> float -> d4 augment (HIP)
    -> dummy 4-class UNet (plain torch / MIOpen convs, random weights) -> fused d4 de-augment + integrate (HIP)
    -> merge (HIP) -> D2H -> crop_to_orignal_size

Reports MP/s for the whole loop and a per-stage breakdown (SURVEY.md 8d "secondary" region; not a roofline figure:
the model and PCIe dominate).  `--tiny` uses a 1024x1024 image.

`--device-edges` runs the same loop with the device-side edges: the uint8 image is uploaded once (75 MB),
`ImageSlicer.split_device(..., augment="d4", scale, bias)` builds each model batch in one HIP launch, and
`TileMerger.merge_crop(tiler, dtype=uint8, argmax=True)` downloads 25 MB of class indices instead of 419 MB of fp32."""
import argparse
import os
import sys
import time

import numpy as np
import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_toolbelt_amd.inference import tta  # noqa: E402
from pytorch_toolbelt_amd.inference.tiles import CudaTileMerger, ImageSlicer  # noqa: E402
from pytorch_toolbelt_amd.utils.torch_utils import image_to_tensor, to_numpy  # noqa: E402


def block(cin, cout):
    return nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1), nn.BatchNorm2d(cout), nn.ReLU(inplace=True),
                         nn.Conv2d(cout, cout, 3, padding=1), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


class DummyUNet(nn.Module):
    """4-level conv3x3-BN-ReLU encoder/decoder, 3 -> 4 channels (the shape of the reference's UNet blocks)."""

    def __init__(self, width=8, classes=4):
        super().__init__()
        w = [width, width * 2, width * 4, width * 8]
        self.enc = nn.ModuleList([block(3, w[0]), block(w[0], w[1]), block(w[1], w[2]), block(w[2], w[3])])
        self.dec = nn.ModuleList([block(w[3] + w[2], w[2]), block(w[2] + w[1], w[1]), block(w[1] + w[0], w[0])])
        self.head = nn.Conv2d(w[0], classes, 1)

    def forward(self, x):
        feats = []
        for i, e in enumerate(self.enc):
            x = e(x if i == 0 else nn.functional.max_pool2d(x, 2))
            feats.append(x)
        for d, skip in zip(self.dec, feats[-2::-1]):
            x = d(torch.cat([nn.functional.interpolate(x, scale_factor=2, mode="nearest"), skip], 1))
        return self.head(x)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiny", action="store_true")
    ap.add_argument("--defer", action="store_true", help="with --device-edges: TileMerger(crops=, defer=True)")
    ap.add_argument("--device-edges", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    side = 1024 if args.tiny else 5000
    image = np.random.default_rng(0).integers(0, 256, (side, side, 3), dtype=np.uint8)
    torch.manual_seed(0)
    model = DummyUNet().eval().to(dev)
    stages = dict(split=0.0, h2d=0.0, augment=0.0, model=0.0, integrate=0.0, merge_d2h=0.0)

    def run():
        t = time.perf_counter()
        tiler = ImageSlicer(image.shape, tile_size=(512, 512), tile_step=(256, 256), weight="pyramid")
        tiles = [image_to_tensor(tile) for tile in tiler.split(image)]
        stages["split"] += time.perf_counter() - t
        merger = CudaTileMerger(tiler.target_shape, 4, tiler.weight)
        with torch.no_grad():
            for b0 in range(0, len(tiles), 8):
                t = time.perf_counter()
                batch = torch.stack(tiles[b0:b0 + 8]).to(dev, non_blocking=True).float().div_(255.0)
                torch.cuda.synchronize(); stages["h2d"] += time.perf_counter() - t; t = time.perf_counter()
                aug = tta.d4_image_augment(batch)
                torch.cuda.synchronize(); stages["augment"] += time.perf_counter() - t; t = time.perf_counter()
                pred = model(aug)
                torch.cuda.synchronize(); stages["model"] += time.perf_counter() - t; t = time.perf_counter()
                merger.integrate_batch_deaugment(pred, tiler.crops[b0:b0 + 8], group="d4")
                torch.cuda.synchronize(); stages["integrate"] += time.perf_counter() - t
        t = time.perf_counter()
        merged = tiler.crop_to_orignal_size(np.moveaxis(to_numpy(merger.merge()), 0, -1))
        stages["merge_d2h"] += time.perf_counter() - t
        return merged

    def run_device(defer=False):
        t = time.perf_counter()
        tiler = ImageSlicer(image.shape, tile_size=(512, 512), tile_step=(256, 256), weight="pyramid")
        stages["split"] += time.perf_counter() - t; t = time.perf_counter()
        dimg = torch.from_numpy(image).to(dev, non_blocking=True)
        torch.cuda.synchronize(); stages["h2d"] += time.perf_counter() - t
        merger = CudaTileMerger(tiler.target_shape, 4, tiler.weight, crops=tiler.crops if defer else None, defer=defer)
        inv255 = [1.0 / 255.0] * 3
        with torch.no_grad():
            for b0 in range(0, len(tiler.crops), 8):
                t = time.perf_counter()
                aug = tiler.split_device(dimg, slice(b0, b0 + 8), augment="d4", scale=inv255, bias=[0.0] * 3)
                torch.cuda.synchronize(); stages["augment"] += time.perf_counter() - t; t = time.perf_counter()
                pred = model(aug)
                torch.cuda.synchronize(); stages["model"] += time.perf_counter() - t; t = time.perf_counter()
                merger.integrate_batch_deaugment(pred, tiler.crops[b0:b0 + 8], group="d4")
                torch.cuda.synchronize(); stages["integrate"] += time.perf_counter() - t
        t = time.perf_counter()
        labels = merger.merge_crop(tiler, argmax=True, dtype=torch.uint8).cpu().numpy()
        stages["merge_d2h"] += time.perf_counter() - t
        return labels

    if args.device_edges:
        ref = run_device()
        for k in stages:
            stages[k] = 0.0
        t0 = time.perf_counter()
        out = run_device(defer=args.defer)
        total = time.perf_counter() - t0
        assert out.shape == (side, side) and out.max() < 4
        # (the deferred merger holds references to the model's freshly allocated output tensors: same labels, bit for bit)
        assert np.array_equal(out, ref), "deferred and incremental mergers disagree"
        print(f"end-to-end (device edges) {side}x{side}: {total * 1e3:.1f} ms -> {side * side / 1e6 / total:.1f} MP/s")
        for k, v in stages.items():
            print(f"  {k:10s} {v * 1e3:9.1f} ms  ({100 * v / total:4.1f} %)")
        return

    run()  # warm-up (MIOpen find, allocator)
    for k in stages:
        stages[k] = 0.0
    t0 = time.perf_counter()
    out = run()
    total = time.perf_counter() - t0
    assert out.shape == (side, side, 4) and np.isfinite(out).all()
    print(f"end-to-end {side}x{side}: {total * 1e3:.1f} ms -> {side * side / 1e6 / total:.1f} MP/s")
    for k, v in stages.items():
        print(f"  {k:10s} {v * 1e3:9.1f} ms  ({100 * v / total:4.1f} %)")


if __name__ == "__main__":
    main()
