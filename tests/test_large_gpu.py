"""GPU: tensors with more than 2^31 elements (the MI355X has 288 GB: nothing here may index with 32 bits).
Each case is checked against plain torch evaluated in pieces."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BIG = 2 ** 31 + 8192          # elements


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    if torch.cuda.get_device_properties(0).total_memory < 100e9:
        pytest.skip("needs ~60 GB of HBM")
    return torch.device("cuda:0")


def test_pointwise_loss_and_ensemble_beyond_2g_elements(dev):
    from pytorch_toolbelt_amd import losses as L
    from pytorch_toolbelt_amd.inference.ensembling import Ensembler

    rows = BIG // 4096
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.empty((rows, 4096), device=dev).normal_(generator=g)
    t = (torch.empty((rows, 4096), device=dev).uniform_(generator=g) < 0.3).float()
    got = L.SoftBCEWithLogitsLoss(ignore_index=None, reduction="sum")(x, t)
    want = sum(float(torch.nn.functional.binary_cross_entropy_with_logits(x[i:i + 65536].double(), t[i:i + 65536].double(), reduction="sum"))
               for i in range(0, rows, 65536))
    assert abs(float(got) - want) <= 1e-6 * abs(want)
    # the tail of the tensor is really visited: poison the last row only
    x2 = x.clone()
    x2[-1] += 50.0
    assert float(L.SoftBCEWithLogitsLoss(ignore_index=None, reduction="sum")(x2, t)) > float(got) + 1000

    class Fixed(torch.nn.Module):
        def __init__(self, v):
            super().__init__()
            self.v = v

        def forward(self, _):
            return self.v

    out = Ensembler([Fixed(x), Fixed(t)], reduction="mean")(None)
    for sl in (slice(0, 8), slice(rows // 2, rows // 2 + 8), slice(rows - 8, rows)):
        assert torch.equal(out[sl], (x[sl] + t[sl]) / 2)


def test_views_and_merger_beyond_2g_elements(dev):
    from pytorch_toolbelt_amd.inference import tta
    from pytorch_toolbelt_amd.inference.tiles import TileMerger

    # d2 de-augment of [4 * 2, 1, 16384, 16400]: 2.15 G input elements
    H, W = 16384, 16400
    g = torch.Generator(device=dev).manual_seed(1)
    y = torch.empty((8, 1, H, W), device=dev).normal_(generator=g)
    out = tta.d2_image_deaugment(y, reduction="sum")
    b = 1
    want = (y[b] + y[2 + b].flip(-1) + y[4 + b].flip(-2) + y[6 + b].flip(-1, -2))
    # (the reference's d2 order: identity, fliplr, flipud, both -- tta.py:344-365)
    assert torch.allclose(out[b], want, atol=1e-5)
    del y, out, want
    torch.cuda.empty_cache()
    # accumulator with more than 2^31 elements per plane pair: tiles in the far corner
    S = 46400
    w = np.ones((64, 64), dtype=np.float32)
    m = TileMerger((S, S), 1, w, device=dev)
    tiles = torch.ones((3, 1, 64, 64), device=dev)
    coords = np.array([[S - 64, S - 64, 64, 64], [S - 96, S - 64, 64, 64], [0, 0, 64, 64]])
    m.integrate_batch(tiles, coords)
    merged = m.merge()
    assert merged.shape == (1, S, S)
    assert float(merged[0, S - 1, S - 1]) == 1.0 and float(merged[0, S - 1, S - 80]) == 1.0 and float(merged[0, 0, 0]) == 1.0
    assert torch.isnan(merged[0, S // 2, S // 2])
    assert float(m.image[0, S - 1, S - 40]) == 2.0 and float(m.norm_mask[0, S - 1, S - 40]) == 2.0


def test_headline_geometry_all_merger_modes_agree(dev):
    """BASELINE configs[1] geometry at full size (5000 x 5000, 512 / 256, 361 tiles, d4, batches of 8; C = 2 to bound memory):
    the deferred band merger, the planned merger and the plain merger give the same bits, and the result has the
    size-independent properties of a merge (partition of unity: constant model outputs come back as that constant)."""
    from pytorch_toolbelt_amd.inference.tiles import ImageSlicer, TileMerger

    slicer = ImageSlicer((5000, 5000, 3), 512, 256, weight="pyramid")
    crops, C = slicer.crops, 2
    assert len(crops) == 361 and slicer.target_shape == (5120, 5120)
    mergers = dict(deferred=TileMerger(slicer.target_shape, C, slicer.weight, device=dev, crops=crops, defer=True),
                   planned=TileMerger(slicer.target_shape, C, slicer.weight, device=dev, crops=crops),
                   plain=TileMerger(slicer.target_shape, C, slicer.weight, device=dev))
    g = torch.Generator(device=dev).manual_seed(5)
    for b0 in range(0, 361, 8):
        nb = min(8, 361 - b0)
        y = torch.empty((8 * nb, C, 512, 512), device=dev).normal_(generator=g)
        for m in mergers.values():
            m.integrate_batch_deaugment(y, crops[b0:b0 + nb], group="d4", reduction="mean")
    d = mergers["deferred"]
    assert d._bands_done == len(d._bands.bands) and not d._held
    out = {k: m.merge() for k, m in mergers.items()}
    assert torch.isfinite(out["plain"]).all()
    assert torch.equal(out["deferred"], out["plain"]) and torch.equal(out["planned"], out["plain"])
    # partition of unity through the deferred path: every tile (all 8 views) constant 0.75 -> the merged map is 0.75
    d.reset()
    for b0 in range(0, 361, 8):
        nb = min(8, 361 - b0)
        # (a tensor of its own per batch: the deferred merger refuses a batch that lives in the memory of one it still holds)
        d.integrate_batch_deaugment(torch.full((8 * nb, C, 512, 512), 0.75, device=dev), crops[b0:b0 + nb], group="d4", reduction="mean")
    flat = d.merge()
    assert float((flat - 0.75).abs().max()) <= 1e-6
