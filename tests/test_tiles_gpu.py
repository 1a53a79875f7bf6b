"""GPU parity: TileMerger (HIP) vs the reference's golden vectors and the numpy oracle.

Tolerances: integrate_batch / merge are bit-exact (same fp32 op order as the reference, no FMA contraction);
the fused TTA path is held to 1e-5 absolute on O(1)-scale inputs (BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import tiles_oracle as TO
from oracle import tta_oracle as AO

pytestmark = pytest.mark.gpu

GT = load_golden("tiles.npz")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture()
def native():
    from pytorch_toolbelt_amd import _native as N

    lib = N.load()
    yield N
    lib.ptb_set_tunable(0, 32)
    lib.ptb_set_tunable(1, 0)


def _merger(shape, C, weight, dev, **kw):
    from pytorch_toolbelt_amd.inference.tiles import TileMerger

    return TileMerger(shape, C, weight, device=dev, **kw)


def _window(kw):
    return TO.pyramid_window(*kw["tile_size"])[0] if kw["weight"] == "pyramid" else TO.mean_window(*kw["tile_size"])


@pytest.mark.parametrize("case", GT.by_fn("tile_merger"), ids=lambda c: c["name"])
def test_golden_tile_merger_bit_exact(case, dev, native):
    from pytorch_toolbelt_amd.inference.tiles import ImageSlicer

    kw, n = case["kwargs"], case["name"]
    s = ImageSlicer(kw["image_shape"], kw["tile_size"], kw["tile_step"], weight=kw["weight"])
    m = _merger(s.target_shape, kw["channels"], s.weight, dev)
    pred = torch.from_numpy(GT[f"{n}_pred"]).to(dev)
    calls = native.calls
    for b0 in range(0, len(pred), kw["batch"]):
        m.integrate_batch(pred[b0:b0 + kw["batch"]], s.crops[b0:b0 + kw["batch"]])
    assert native.calls > calls
    assert np.array_equal(m.image.cpu().numpy(), GT[f"{n}_image"])
    assert np.array_equal(m.norm_mask.cpu().numpy(), GT[f"{n}_norm"])
    assert np.array_equal(m.merge().cpu().numpy(), GT[f"{n}_merged"])


def _oracle_merge(geom, C, weight, pred, batch):
    st = TO.merger_new(geom["target_shape"], C, weight)
    for b0 in range(0, len(pred), batch):
        TO.merger_integrate(st, pred[b0:b0 + batch], geom["crops"][b0:b0 + batch])
    return st


@pytest.mark.parametrize("chunk_rows,scalar", [(64, 0), (32, 0), (16, 0), (64, 1)])
@pytest.mark.parametrize(
    "shape,tile,step,C,batch",
    [
        ((300, 420), (128, 128), (64, 64), 3, 8),     # 50% overlap, vector path
        ((256, 256), (64, 64), (16, 16), 2, 7),       # 16-fold cover -> launch groups are split
        ((200, 330), (96, 72), (96, 72), 1, 40),      # no overlap, many tiles per batch
        ((130, 170), (52, 36), (20, 12), 2, 9),       # multiples of 4 but ragged chunks
        ((77, 91), (25, 31), (11, 17), 2, 6),         # odd sizes -> scalar kernels
    ],
)
def test_integrate_batch_matches_oracle_bit_exact(shape, tile, step, C, batch, chunk_rows, scalar, dev, native):
    lib = native.load()
    lib.ptb_set_tunable(0, chunk_rows)
    lib.ptb_set_tunable(1, scalar)
    geom = TO.slicer_geometry(shape, tile, step)
    w = TO.pyramid_window(*tile)[0]
    rng = np.random.default_rng(1)
    pred = rng.standard_normal((len(geom["crops"]), C, *tile)).astype(np.float32)
    st = _oracle_merge(geom, C, w, pred, batch)
    m = _merger(geom["target_shape"], C, w, dev)
    tp = torch.from_numpy(pred).to(dev)
    for b0 in range(0, len(pred), batch):
        m.integrate_batch(tp[b0:b0 + batch], geom["crops"][b0:b0 + batch])
    assert np.array_equal(m.image.cpu().numpy(), st["image"])
    assert np.array_equal(m.norm_mask.cpu().numpy(), st["norm_mask"])
    assert np.array_equal(m.merge().cpu().numpy(), TO.merger_merge(st), equal_nan=True)


def test_coords_formats_errors_and_inplace(dev):
    geom = TO.slicer_geometry((128, 128), 64, 32)
    w = TO.pyramid_window(64, 64)[0]
    rng = np.random.default_rng(2)
    pred = rng.standard_normal((len(geom["crops"]), 2, 64, 64)).astype(np.float32)
    want = _oracle_merge(geom, 2, w, pred, 3)
    tp = torch.from_numpy(pred).to(dev)
    crops = geom["crops"]
    for fmt in (lambda c: c, lambda c: [tuple(int(v) for v in r) for r in c], lambda c: torch.from_numpy(c), lambda c: [torch.tensor(r) for r in c]):
        m = _merger(geom["target_shape"], 2, w, dev)
        for b0 in range(0, len(pred), 3):
            m.integrate_batch(tp[b0:b0 + 3], fmt(crops[b0:b0 + 3]))
        assert np.array_equal(m.image.cpu().numpy(), want["image"])
    # accumulate_single == batch of one; CPU / fp16 inputs are moved + cast like the reference
    m = _merger(geom["target_shape"], 2, w, dev)
    for t, c in zip(pred, crops):
        m.accumulate_single(torch.from_numpy(t), c)
    st1 = _oracle_merge(geom, 2, w, pred, 1)
    assert np.array_equal(m.image.cpu().numpy(), st1["image"])
    # merge_ is in place and aliases image
    merged = m.merge().cpu().numpy()
    out = m.merge_()
    assert out.data_ptr() == m.image.data_ptr() and np.array_equal(out.cpu().numpy(), merged)
    assert m.device == m.image.device
    with pytest.raises(ValueError):
        m.integrate_batch(tp[:2], crops[:3])
    with pytest.raises(RuntimeError):
        m.integrate_batch(tp[:1], np.array([[100, 100, 64, 64]]))  # leaves the accumulator
    with pytest.raises(RuntimeError):
        m.integrate_batch(tp[:1, :, :32], crops[:1])
    # half-precision predictions are cast to the fp32 accumulator (reference tiles.py:334-335)
    m2 = _merger(geom["target_shape"], 2, w, dev)
    m2.integrate_batch(tp[:4].half(), crops[:4])
    st = TO.merger_new(geom["target_shape"], 2, w)
    TO.merger_integrate(st, pred[:4].astype(np.float16).astype(np.float32), crops[:4])
    assert np.array_equal(m2.image.cpu().numpy(), st["image"])


def test_uncovered_pixels_are_nan(dev):
    m = _merger((64, 64), 1, np.ones((32, 32), np.float32), dev)
    m.integrate_batch(torch.ones((1, 1, 32, 32), device=dev), [(0, 0, 32, 32)])
    out = m.merge().cpu().numpy()
    assert out[0, 0, 0] == 1.0 and np.isnan(out[0, 40, 40])


def test_readme_loop_with_dataloader(dev):
    """The reference's canonical loop (README.md:208-226 == tests/test_tiles.py:58-85) on a NON-zero image."""
    from torch.utils.data import DataLoader

    from pytorch_toolbelt_amd.inference.tiles import CudaTileMerger, ImageSlicer
    from pytorch_toolbelt_amd.utils.torch_utils import image_to_tensor, to_numpy

    image = np.random.default_rng(0).integers(0, 256, (500, 620, 3), dtype=np.uint8)
    tiler = ImageSlicer(image.shape, tile_size=(128, 128), tile_step=(64, 64), weight="pyramid")
    tiles = [image_to_tensor(t) for t in tiler.split(image)]
    merger = CudaTileMerger(tiler.target_shape, 1, tiler.weight)
    for tiles_batch, coords_batch in DataLoader(list(zip(tiles, tiler.crops)), batch_size=8, pin_memory=True):
        pred = tiles_batch.float().to(dev).max(dim=1, keepdim=True)[0]
        merger.integrate_batch(pred, coords_batch)
    merged = np.moveaxis(to_numpy(merger.merge()), 0, -1)
    merged = tiler.crop_to_orignal_size(merged)
    np.testing.assert_array_equal(np.rint(merged).astype(np.uint8), image.max(axis=2, keepdims=True))


GROUPS = {"fliplr": 2, "flipud": 2, "flips": 3, "d2": 4, "d4": 8}


@pytest.mark.parametrize("chunk_rows,scalar", [(64, 0), (32, 0), (16, 0), (64, 1)])
@pytest.mark.parametrize("group", list(GROUPS))
def test_fused_deaugment_accumulate(group, chunk_rows, scalar, dev, native):
    lib = native.load()
    lib.ptb_set_tunable(0, chunk_rows)
    lib.ptb_set_tunable(1, scalar)
    V = GROUPS[group]
    shape, tile, step, C, batch = (300, 300), (128, 128), (64, 64), 3, 5
    geom = TO.slicer_geometry(shape, tile, step)
    w = TO.pyramid_window(*tile)[0]
    n = len(geom["crops"])
    rng = np.random.default_rng(4)
    outs = rng.standard_normal((V, n, C, *tile)).astype(np.float32)  # [view][tile]
    st = TO.merger_new(geom["target_shape"], C, w)
    m = _merger(geom["target_shape"], C, w, dev)
    for b0 in range(0, n, batch):
        b1 = min(n, b0 + batch)
        chunk_major = np.concatenate([outs[k, b0:b1] for k in range(V)])
        TO.merger_integrate(st, AO.image_deaugment(chunk_major, group, "mean"), geom["crops"][b0:b1])
        m.integrate_batch_deaugment(torch.from_numpy(chunk_major).to(dev), geom["crops"][b0:b1], group=group, reduction="mean")
    assert np.array_equal(m.norm_mask.cpu().numpy(), st["norm_mask"])
    np.testing.assert_allclose(m.image.cpu().numpy(), st["image"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(m.merge().cpu().numpy(), TO.merger_merge(st), rtol=0, atol=1e-5)


@pytest.mark.parametrize("reduction", ["sum", "gmean", "hmean", "harmonic1p", "logodd", "log1p"])
def test_fused_reductions(reduction, dev):
    geom = TO.slicer_geometry((192, 192), (64, 64), (32, 32))
    w = TO.pyramid_window(64, 64)[0]
    n = len(geom["crops"])
    rng = np.random.default_rng(5)
    outs = (rng.random((8 * n, 2, 64, 64)) * 0.98 + 0.01).astype(np.float32)
    st = TO.merger_new(geom["target_shape"], 2, w)
    TO.merger_integrate(st, AO.image_deaugment(outs, "d4", reduction), geom["crops"])
    m = _merger(geom["target_shape"], 2, w, dev)
    m.integrate_batch_deaugment(torch.from_numpy(outs).to(dev), geom["crops"], group="d4", reduction=reduction)
    np.testing.assert_allclose(m.merge().cpu().numpy(), TO.merger_merge(st), rtol=1e-5, atol=1e-5)


def test_fused_equals_unfused_pipeline(dev):
    """integrate_batch(tta.d4_image_deaugment(y)) and the fused call give the same accumulator (same op order)."""
    from pytorch_toolbelt_amd.inference import tta

    geom = TO.slicer_geometry((256, 256), (128, 128), (64, 64))
    w = TO.pyramid_window(128, 128)[0]
    n = len(geom["crops"])
    y = torch.randn((8 * n, 2, 128, 128), device=dev)
    a = _merger(geom["target_shape"], 2, w, dev)
    b = _merger(geom["target_shape"], 2, w, dev)
    a.integrate_batch(tta.d4_image_deaugment(y), geom["crops"])
    b.integrate_batch_deaugment(y, geom["crops"], group="d4")
    assert torch.equal(a.image, b.image) and torch.equal(a.norm_mask, b.norm_mask)


def test_full_size_cfg2_properties(dev):
    """BASELINE cfg2 geometry (5000x5000, 512/256 pyramid, d4, C=4) through size-independent properties:
    a constant prediction merges to that constant everywhere (the window is a partition of unity after
    normalisation), and the merge is linear in the predictions."""
    from pytorch_toolbelt_amd.inference.tiles import ImageSlicer

    s = ImageSlicer((5000, 5000, 3), 512, 256, weight="pyramid")
    assert len(s.crops) == 361 and s.target_shape == (5120, 5120)
    C, bs = 4, 8
    g = torch.Generator(device=dev).manual_seed(0)
    base = torch.randn((8 * bs, C, 512, 512), device=dev, generator=g)

    def run(batch_fn):
        m = _merger(s.target_shape, C, s.weight, dev)
        for b0 in range(0, 361, bs):
            b1 = min(361, b0 + bs)
            m.integrate_batch_deaugment(batch_fn(b0, b1), s.crops[b0:b1], group="d4")
        return m

    const = run(lambda b0, b1: torch.full((8 * (b1 - b0), C, 512, 512), 0.75, device=dev))
    out = const.merge()
    assert float((out - 0.75).abs().max()) <= 1e-6
    assert float(const.norm_mask.min()) > 0.0

    def tiles(b0, b1, scale):
        nb = b1 - b0
        return (base.view(8, bs, C, 512, 512)[:, :nb] * scale + 0.01 * b0).reshape(8 * nb, C, 512, 512)

    m1 = run(lambda b0, b1: tiles(b0, b1, 1.0)).merge()
    m2 = run(lambda b0, b1: tiles(b0, b1, 2.0)).merge()
    m3 = run(lambda b0, b1: tiles(b0, b1, 3.0)).merge()
    # linearity: f(3x) - f(2x) == f(2x) - f(x)
    assert float(((m3 - m2) - (m2 - m1)).abs().max()) <= 2e-5
    crop = s.crop_to_orignal_size(np.moveaxis(m1.cpu().numpy(), 0, -1))
    assert crop.shape == (5000, 5000, C) and np.isfinite(crop).all()


class _SoloDist:
    """Single-rank stand-in for torch.distributed: exercises the sharded merger's HIP paths (band accumulators, local
    normaliser, strided band merge) on one GPU; the exchange itself is covered over gloo in tests/test_sharded_cpu.py."""

    def __init__(self, rank=0, world=1):
        self.rank, self.world = rank, world

    def get_rank(self, group=None):
        return self.rank

    def get_world_size(self, group=None):
        return self.world


@pytest.mark.parametrize("defer", [True, False], ids=["deferred-bands", "incremental"])
@pytest.mark.parametrize("partition", ["tiles", "rows"])
@pytest.mark.parametrize("world", [1, 2, 3, 5])
def test_sharded_merger_bands_on_gpu(world, partition, defer, dev):
    """Every rank of a `world`-way sharding is played in turn on one GPU; halo rectangles are handed over by hand.
    The assembled result must equal the single-device TileMerger (same HIP kernels, different accumulation order)."""
    from pytorch_toolbelt_amd.parallel import ShardedTileMerger

    geom = TO.slicer_geometry((700, 520), (128, 128), (64, 64))
    w = TO.pyramid_window(128, 128)[0]
    crops, C = geom["crops"], 3
    n = len(crops)
    y = torch.randn((8, n, C, 128, 128), device=dev)
    single = _merger(geom["target_shape"], C, w, dev)
    for b0 in range(0, n, 8):
        idx = list(range(b0, min(n, b0 + 8)))
        single.integrate_batch_deaugment(y[:, idx].reshape(-1, C, 128, 128), crops[idx], group="d4")
    want = single.merge()

    ranks = []
    for r in range(world):
        m = ShardedTileMerger(geom["target_shape"], C, w, crops, device=dev, dist=_SoloDist(r, world), partition=partition, defer=defer)
        m._start_exchange = lambda: None  # no process group here: rectangles are moved below
        mine = m.tiles
        m.reset()
        for b0 in range(0, len(mine), 8):
            idx = mine[b0:b0 + 8]
            m.integrate_batch_deaugment(y[:, idx].reshape(-1, C, 128, 128), crops[idx], group="d4")
        ranks.append(m)
    full = torch.empty_like(want)
    for r, m in enumerate(ranks):
        for buf, (src, r0, r1, c0, c1) in zip(m._recv_buf, m.recvs):  # what rank `src` would have sent
            buf.copy_(ranks[src]._rect(r0, r1, c0, c1))
        m._exchanged = True
        band = m.merge()
        o0, o1 = m.owned_rows
        full[:, o0:o1] = band
    torch.testing.assert_close(full, want, rtol=0, atol=2e-6)


def test_first_touch_reset_and_reuse(dev):
    """`reset()` re-arms the first-touch bitmap: a reused merger must give the same bits as a new one, stale data from
    the previous image must never leak (also where the second image covers less), and mid-way reads see zeros."""
    geom = TO.slicer_geometry((512, 768), (256, 256), (128, 128))
    w = TO.pyramid_window(256, 256)[0]
    crops = geom["crops"]
    n = len(crops)
    rng = np.random.default_rng(8)
    a = rng.standard_normal((n, 2, 256, 256)).astype(np.float32)
    b = rng.standard_normal((n, 2, 256, 256)).astype(np.float32)
    m = _merger(geom["target_shape"], 2, w, dev)
    m.integrate_batch(torch.from_numpy(a).to(dev), crops)
    first = m.merge().cpu().numpy()
    assert np.array_equal(first, TO.merger_merge(_oracle_merge(geom, 2, w, a, n)))
    m.reset()
    half = n // 2
    m.integrate_batch(torch.from_numpy(b[:half]).to(dev), crops[:half])
    st = TO.merger_new(geom["target_shape"], 2, w)
    TO.merger_integrate(st, b[:half], crops[:half])
    # mid-way read: untouched blocks read as zeros (not the previous image), touched ones hold the partial sums
    assert np.array_equal(m.image.cpu().numpy(), st["image"]) and np.array_equal(m.norm_mask.cpu().numpy(), st["norm_mask"])
    m.integrate_batch(torch.from_numpy(b[half:]).to(dev), crops[half:])
    TO.merger_integrate(st, b[half:], crops[half:])
    assert np.array_equal(m.merge().cpu().numpy(), TO.merger_merge(st))
    # the attributes stay assignable like plain tensors
    m.image = torch.ones_like(m.image)
    m.norm_mask = torch.full_like(m.norm_mask, 2.0)
    assert float(m.merge().max()) == 0.5


def test_lazy_norm_mask_cache_and_eager_switch(dev, native):
    """``norm_mask`` is data independent: the accumulate kernels skip it, ``merge`` materialises it from the crop log and
    reuses the buffer when the next image brings the same crops.  Every observable value stays the reference's."""
    geom = TO.slicer_geometry((512, 768), (256, 256), (128, 128))
    w = TO.pyramid_window(256, 256)[0]
    crops, n = geom["crops"], len(geom["crops"])
    rng = np.random.default_rng(11)
    imgs = [rng.standard_normal((n, 2, 256, 256)).astype(np.float32) for _ in range(3)]
    m = _merger(geom["target_shape"], 2, w, dev, auto_plan=False)      # (this test counts the launches of the ordinary path)

    def run(pred, order=None, batch=5):
        idx = np.arange(n) if order is None else order
        st = TO.merger_new(geom["target_shape"], 2, w)
        for b0 in range(0, n, batch):
            sel = idx[b0:b0 + batch]
            m.integrate_batch(torch.from_numpy(pred[sel]).to(dev), crops[sel])
            TO.merger_integrate(st, pred[sel], crops[sel])
        return st

    st = run(imgs[0])
    c0 = native.calls
    assert np.array_equal(m.merge().cpu().numpy(), TO.merger_merge(st))
    first_merge_calls = native.calls - c0           # norm accumulate + merge
    m.reset()
    st = run(imgs[1])
    c0 = native.calls
    assert np.array_equal(m.merge().cpu().numpy(), TO.merger_merge(st))
    assert native.calls - c0 == first_merge_calls - 1, "the normaliser of an identical crop log must be reused"
    # a different tile order is a different log (fp32 sums depend on the order): rebuilt, still bit exact
    m.reset()
    order = rng.permutation(n)
    st = run(imgs[2], order)
    c0 = native.calls
    assert np.array_equal(m.merge().cpu().numpy(), TO.merger_merge(st))
    assert native.calls - c0 == first_merge_calls
    assert np.array_equal(m.norm_mask.cpu().numpy(), st["norm_mask"])
    # partial merge, more tiles, merge again; then a read of norm_mask switches to the kernel-maintained mode
    m.reset()
    st = TO.merger_new(geom["target_shape"], 2, w)
    for lo, hi in ((0, 4), (4, 9)):
        m.integrate_batch(torch.from_numpy(imgs[0][lo:hi]).to(dev), crops[lo:hi])
        TO.merger_integrate(st, imgs[0][lo:hi], crops[lo:hi])
        assert np.array_equal(m.merge().cpu().numpy(), TO.merger_merge(st), equal_nan=True)
    held = m.norm_mask
    assert np.array_equal(held.cpu().numpy(), st["norm_mask"])
    m.integrate_batch(torch.from_numpy(imgs[0][9:]).to(dev), crops[9:])
    TO.merger_integrate(st, imgs[0][9:], crops[9:])
    assert np.array_equal(held.cpu().numpy(), st["norm_mask"]), "a handed-out norm_mask must keep tracking the integrate calls"
    assert np.array_equal(m.merge().cpu().numpy(), TO.merger_merge(st))
    # after such a cycle nothing stale is reused
    m.reset()
    st = run(imgs[1])
    assert np.array_equal(m.merge().cpu().numpy(), TO.merger_merge(st))
    # a new window invalidates the cached normaliser
    m.reset()
    w2 = (w * 0.5 + 0.1).astype(w.dtype)
    m.weight = torch.from_numpy(w2[None]).to(dev, torch.float32)
    st = TO.merger_new(geom["target_shape"], 2, w2)
    m.integrate_batch(torch.from_numpy(imgs[1]).to(dev), crops)
    TO.merger_integrate(st, imgs[1], crops)
    assert np.array_equal(m.merge().cpu().numpy(), TO.merger_merge(st))


def _planned_run(m, pred, crops, order, batch, dev, fused=None):
    st = TO.merger_new((m.image_height, m.image_width), m.channels, m.weight[0].cpu().numpy())
    for b0 in range(0, len(order), batch):
        sel = order[b0:b0 + batch]
        if fused is None:
            m.integrate_batch(torch.from_numpy(pred[sel]).to(dev), crops[sel])
            TO.merger_integrate(st, pred[sel], crops[sel])
        else:
            V = pred.shape[1]
            x = np.ascontiguousarray(np.moveaxis(pred[sel], 1, 0)).reshape(V * len(sel), *pred.shape[2:])   # chunk-major views
            m.integrate_batch_deaugment(torch.from_numpy(x).to(dev), crops[sel], group=fused, reduction="mean")
            TO.merger_integrate(st, AO.image_deaugment(x, fused, "mean"), crops[sel])
    return st


@pytest.mark.parametrize("shape,tile,step,C,batch,all_final", [
    ((1000, 1400), (256, 256), (128, 128), 3, 8, True),   # margins -> 1024 x 1536 target, everything block aligned
    ((512, 768), (256, 256), (128, 128), 2, 5, True),
    # step < size / 2 on x: up to 6 tiles per pixel, so a batch of 7 needs several launch groups -> that batch (and the
    # rest of the image) falls back to the ordinary path; the result must not change
    ((700, 640), (128, 192), (64, 64), 1, 7, False),
    ((700, 640), (128, 192), (64, 64), 1, 2, True),
    ((640, 640), (320, 320), (320, 320), 2, 3, True),     # no overlap: every cell is final at once
])
def test_planned_merger_bit_exact(shape, tile, step, C, batch, all_final, dev, native):
    """TileMerger(crops=...): blocks are divided by the precomputed normaliser in the launch that brings their last tile;
    merge() must equal the ordinary integrate + merge bit for bit, image after image."""
    from pytorch_toolbelt_amd.inference.tiles import TileMerger

    geom = TO.slicer_geometry(shape, tile, step)
    w = TO.pyramid_window(*tile)[0]
    crops, n = geom["crops"], len(geom["crops"])
    rng = np.random.default_rng(n)
    m = TileMerger(geom["target_shape"], C, w, device=dev, crops=crops)
    assert m._plan is not None
    for rep in range(3):
        pred = rng.standard_normal((n, C, *tile)).astype(np.float32)
        st = _planned_run(m, pred, crops, np.arange(n), batch, dev)
        c0 = native.calls
        got = m.merge()
        if all_final:
            assert m._plan.done.all() and not m._plan.remaining.any(), "every block must have been finalised in-launch"
            assert native.calls == c0, "nothing is left for merge() to do"
        assert np.array_equal(got.cpu().numpy(), TO.merger_merge(st))
        keep = got
        m.reset()
    assert np.array_equal(keep.cpu().numpy(), TO.merger_merge(st)), "a returned result must survive reset()"
    # merge_crop reads the planned result
    st = _planned_run(m, pred, crops, np.arange(n), batch, dev)
    want = TO.merger_merge(st)[:, 3:3 + 200, 5:5 + 300]
    assert np.array_equal(m.merge_crop((3, 5, 200, 300), layout="chw").cpu().numpy(), want)


def test_planned_merger_fused_d4_and_deviations(dev, native):
    from pytorch_toolbelt_amd.inference.tiles import TileMerger

    geom = TO.slicer_geometry((768, 1024), (256, 256), (128, 128))
    w = TO.pyramid_window(256, 256)[0]
    crops, n = geom["crops"], len(geom["crops"])
    rng = np.random.default_rng(5)
    views = rng.standard_normal((n, 8, 2, 256, 256)).astype(np.float32)
    m = TileMerger(geom["target_shape"], 2, w, device=dev, crops=crops)
    st = _planned_run(m, views, crops, np.arange(n), 8, dev, fused="d4")
    assert m._plan.done.all()
    assert np.allclose(m.merge().cpu().numpy(), TO.merger_merge(st), atol=1e-5)
    # same data through an unplanned merger: identical bits (the fused reduction itself is order-identical)
    u = TileMerger(geom["target_shape"], 2, w, device=dev)
    _planned_run(u, views, crops, np.arange(n), 8, dev, fused="d4")
    assert torch.equal(m.merge(), u.merge())
    # a different batch size than planned is still the planned sequence
    m.reset()
    pred = rng.standard_normal((n, 2, 256, 256)).astype(np.float32)
    st = _planned_run(m, pred, crops, np.arange(n), 3, dev)
    assert np.array_equal(m.merge().cpu().numpy(), TO.merger_merge(st))
    # deviation: the second half arrives in reverse order -> finalisation stops, the rest is merged the ordinary way
    m.reset()
    order = np.concatenate([np.arange(n // 2), np.arange(n // 2, n)[::-1]])
    st = _planned_run(m, pred, crops, order, 4, dev)
    assert not m._plan.active and not m._plan.done.all()
    c0 = native.calls
    assert np.array_equal(m.merge().cpu().numpy(), TO.merger_merge(st))
    assert native.calls > c0
    # partial image: merge mid-way (NaN where nothing arrived), continue, merge again
    m.reset()
    st = TO.merger_new(geom["target_shape"], 2, w)
    for lo, hi in ((0, 8), (8, n)):
        m.integrate_batch(torch.from_numpy(pred[lo:hi]).to(dev), crops[lo:hi])
        TO.merger_integrate(st, pred[lo:hi], crops[lo:hi])
        assert np.array_equal(m.merge().cpu().numpy(), TO.merger_merge(st), equal_nan=True)
    # restrictions once blocks are final
    with pytest.raises(RuntimeError, match="finalised"):
        m.image
    with pytest.raises(RuntimeError, match="finalised"):
        m.merge_()
    with pytest.raises(RuntimeError, match="already finalised"):
        m.integrate_batch(torch.from_numpy(pred[:1]).to(dev), crops[:1])
    # ...but before anything is final the plan can be dropped silently
    m.reset()
    m.integrate_batch(torch.from_numpy(pred[:1]).to(dev), crops[:1])       # a corner tile alone finalises its outer quarter
    m.reset()
    assert m.image.abs().sum() == 0 and not m._plan.active
    # geometry off the block grid: no plan, ordinary behaviour
    g2 = TO.slicer_geometry((500, 500), (51, 51), (26, 26))
    m2 = TileMerger(g2["target_shape"], 1, TO.pyramid_window(51, 51)[0], device=dev, crops=g2["crops"])
    assert m2._plan is None


@pytest.mark.parametrize("seed", range(24))
def test_random_geometries_all_merger_modes(seed, dev, native):
    """Seeded differential test: random slicer geometries (block aligned and not), batch sizes, channel counts and tile
    orders through the three TileMerger modes -- lazy normaliser (default), kernel-maintained normaliser (norm_mask read
    up front) and planned (crops= given) -- all bit-identical to the oracle's sequential loop, image after image."""
    from pytorch_toolbelt_amd.inference.tiles import TileMerger

    rng = np.random.default_rng(1000 + seed)
    aligned = seed % 3 != 0
    if aligned:
        tile = (int(rng.choice([64, 128, 192])), int(rng.choice([64, 128, 192, 256])))
        step = (int(rng.choice([s for s in (32, 64, 96, 128, 192) if s <= tile[0]])), int(rng.choice([s for s in (64, 128, 192, 256) if s <= tile[1]])))
    else:
        tile = (int(rng.integers(9, 70)), int(rng.integers(9, 70)))
        step = (int(rng.integers(max(1, tile[0] // 3), tile[0] + 1)), int(rng.integers(max(1, tile[1] // 3), tile[1] + 1)))
    shape = (int(rng.integers(tile[0], 4 * tile[0] + 40)), int(rng.integers(tile[1], 4 * tile[1] + 40)))
    C = int(rng.integers(1, 5))
    batch = int(rng.integers(1, 12))
    geom = TO.slicer_geometry(shape, tile, step)
    crops, n = geom["crops"], len(geom["crops"])
    w = TO.pyramid_window(*tile)[0] if seed % 2 else TO.mean_window(*tile)
    mergers = {
        "lazy": TileMerger(geom["target_shape"], C, w, device=dev),
        "eager": TileMerger(geom["target_shape"], C, w, device=dev),
        "planned": TileMerger(geom["target_shape"], C, w, device=dev, crops=crops),
    }
    for image in range(2):
        pred = rng.standard_normal((n, C, *tile)).astype(np.float32)
        order = np.arange(n) if (image == 0 or seed % 4) else rng.permutation(n)
        st = TO.merger_new(geom["target_shape"], C, w)
        for b0 in range(0, n, batch):
            TO.merger_integrate(st, pred[order[b0:b0 + batch]], crops[order[b0:b0 + batch]])
        want = TO.merger_merge(st)
        for name, m in mergers.items():
            m.reset()
            if name == "eager":
                m.norm_mask  # noqa: B018  (handing the tensor out switches to the kernel-maintained normaliser)
            for b0 in range(0, n, batch):
                sel = order[b0:b0 + batch]
                m.integrate_batch(torch.from_numpy(pred[sel]).to(dev), crops[sel])
            got = m.merge().cpu().numpy()
            assert np.array_equal(got, want, equal_nan=True), (name, image, shape, tile, step, C, batch)
            if name != "planned" or not m._plan or not m._plan.done.any():
                assert np.array_equal(m.norm_mask.cpu().numpy(), st["norm_mask"]), (name, image)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_half_precision_model_outputs_are_read_natively(dtype, dev, native):
    """fp16 / bf16 predictions (autocast inference): the kernels widen them in registers, so every result equals what the
    reference's `batch.type_as(self.image)` copy would give -- bit for bit -- in all merger modes, fused or not, and for
    shapes that need the scalar kernels (where the cast fallback is taken)."""
    from pytorch_toolbelt_amd.inference import tta
    from pytorch_toolbelt_amd.inference.tiles import TileMerger

    rng = np.random.default_rng(3)
    for shape, tile, step, C in (((512, 768), (256, 256), (128, 128), 3), ((200, 230), (50, 50), (25, 30), 2)):
        geom = TO.slicer_geometry(shape, tile, step)
        crops, n = geom["crops"], len(geom["crops"])
        w = TO.pyramid_window(*tile)[0]
        pred = torch.from_numpy(rng.standard_normal((n, 8, C, *tile)).astype(np.float32)).to(dev).to(dtype)   # [n, views, C, h, w]
        for planned in (False, True):
            a = TileMerger(geom["target_shape"], C, w, device=dev, crops=crops if planned else None)
            b = TileMerger(geom["target_shape"], C, w, device=dev, crops=crops if planned else None)
            for b0 in range(0, n, 6):
                sel = slice(b0, min(n, b0 + 6))
                a.integrate_batch(pred[sel, 0], crops[sel])
                b.integrate_batch(pred[sel, 0].float(), crops[sel])
            assert torch.equal(a.merge(), b.merge())
            a.reset(); b.reset()
            for b0 in range(0, n, 6):
                sel = slice(b0, min(n, b0 + 6))
                x = pred[sel].transpose(0, 1).reshape(-1, C, *tile).contiguous()      # chunk-major views
                a.integrate_batch_deaugment(x, crops[sel], group="d4", reduction="mean")
                b.integrate_batch_deaugment(x.float(), crops[sel], group="d4", reduction="mean")
            assert torch.equal(a.merge(), b.merge())
        x = pred[:4].transpose(0, 1).reshape(-1, C, *tile).contiguous()
        for red in ("mean", "gmean", "sum"):
            xin = x.abs() + 0.1 if red == "gmean" else x
            got = tta.d4_image_deaugment(xin, reduction=red)
            assert got.dtype == dtype
            assert torch.equal(got, tta.d4_image_deaugment(xin.float(), reduction=red).to(dtype))


# ------------------------------------------------------------------ deferred band merging (TileMerger(crops=, defer=True))
def _deferred_case(dev, shape, tile, step, C, group, reduction, dtype, bs, images=2, seed=0):
    from pytorch_toolbelt_amd.inference.tiles import ImageSlicer, TileMerger
    from pytorch_toolbelt_amd.inference.tta import DEAUGMENT_VIEWS

    slicer = ImageSlicer(shape + (3,), tile, step, weight="pyramid")
    crops = slicer.crops
    V = len(DEAUGMENT_VIEWS[group])
    th = tile if isinstance(tile, int) else tile[0]
    tw = tile if isinstance(tile, int) else tile[1]
    deferred = TileMerger(slicer.target_shape, C, slicer.weight, device=dev, crops=crops, defer=True)
    plain = TileMerger(slicer.target_shape, C, slicer.weight, device=dev)
    g = torch.Generator(device="cpu").manual_seed(seed)
    for image_no in range(images):
        deferred.reset()
        plain.reset()
        for b0 in range(0, len(crops), bs):
            nb = min(bs, len(crops) - b0)
            y = (torch.rand((V * nb, C, th, tw), generator=g) * 0.9 + 0.05).to(dev).to(dtype)
            deferred.integrate_batch_deaugment(y, crops[b0:b0 + nb], group=group, reduction=reduction)
            plain.integrate_batch_deaugment(y.clone(), crops[b0:b0 + nb], group=group, reduction=reduction)
            del y
        got, want = deferred.merge(), plain.merge()
        assert torch.equal(got, want), f"image {image_no}: max diff {float((got - want).abs().max())}"
    return deferred


@pytest.mark.parametrize("shape,tile,step,C,group,reduction,dtype,bs", [
    ((1000, 700), 256, 128, 3, "d4", "mean", torch.float32, 8),
    ((1000, 700), 256, 128, 2, "d4", "gmean", torch.float32, 5),
    ((640, 900), 128, 64, 4, "d2", "mean", torch.float32, 7),
    ((600, 520), (128, 192), (64, 128), 1, "fliplr", "sum", torch.float32, 3),    # non-square tiles: no transposing views
    ((512, 512), 256, 256, 2, "d4", "mean", torch.float32, 4),                    # no overlap at all
    ((700, 700), 256, 192, 2, "flips", "mean", torch.float32, 6),                 # step > tile / 2: bands of 64 and 192 rows
    ((900, 600), 256, 128, 3, "d4", "mean", torch.float16, 8),
    ((900, 600), 256, 128, 3, "d4", "mean", torch.bfloat16, 2),
])
def test_deferred_band_merge_is_bit_identical(shape, tile, step, C, group, reduction, dtype, bs, dev):
    """Every band of the image is merged in ONE launch from all its tiles (no accumulator in HBM); the result must equal the
    incremental merger's bit for bit, image after image, for every view group, reduction, source dtype and batch size."""
    from pytorch_toolbelt_amd import _native as N

    before = N.fresh_fallbacks
    m = _deferred_case(dev, shape, tile, step, C, group, reduction, dtype, bs)
    assert m._bands is not None and m._bands_done == len(m._bands.bands) and m._defer_active   # the deferred path did run
    assert not m._held                                                                          # and let go of every batch
    assert N.fresh_fallbacks == before
    assert m.fast_submits > 0, "the cached host path of the deferred merger was not taken"


def test_deferred_merger_fallbacks_and_restrictions(dev):
    from pytorch_toolbelt_amd.inference.tiles import ImageSlicer, TileMerger

    slicer = ImageSlicer((700, 600, 3), 256, 128, weight="pyramid")
    crops = slicer.crops
    n, C = len(crops), 2
    y = torch.randn((n, C, 256, 256), device=dev)
    ref = TileMerger(slicer.target_shape, C, slicer.weight, device=dev)
    ref.integrate_batch(y, crops)
    want = ref.merge()
    # 1. reading .image before the first band is merged: the held batches are replayed through the incremental path
    m = TileMerger(slicer.target_shape, C, slicer.weight, device=dev, crops=crops, defer=True, defer_rows=128)
    m.integrate_batch(y[:2], crops[:2])
    assert m._defer_active and len(m._held) == 1 and m._bands_done == 0
    img = m.image
    assert not m._defer_active and float(img.abs().sum()) > 0
    m.integrate_batch(y[2:], crops[2:])
    assert torch.equal(m.merge(), want)
    # 2. the same merger, next image: deferred again; a batch that is not the planned one before any band -> fallback
    m.reset()
    assert m._defer_active
    m.integrate_batch(y[:1], crops[:1])
    m.integrate_batch(y[2:3], crops[2:3])          # skips tile 1
    assert not m._defer_active
    m.integrate_batch(y[1:2], crops[1:2])
    m.integrate_batch(y[3:], crops[3:])
    torch.testing.assert_close(m.merge(), want, rtol=1e-6, atol=1e-6)   # (different order of additions)
    # 3. after bands were merged the incremental state does not exist: .image and a partial merge() raise
    m.reset()
    row = int(np.sum(crops[:, 1] == crops[0, 1]))
    m.integrate_batch(y[:row], crops[:row])        # the whole first tile row: band 0 is merged
    assert m._bands_done >= 1
    with pytest.raises(RuntimeError, match="defer=True"):
        _ = m.image
    with pytest.raises(RuntimeError, match="defer=True"):
        m.merge()
    m.integrate_batch(y[row:], crops[row:])
    assert torch.equal(m.merge(), want)
    assert torch.equal(m.merge_crop(slicer, layout="chw"), want[:, slicer.margin_top:slicer.margin_top + 700, slicer.margin_left:slicer.margin_left + 600])
    # 4. geometry the band kernel does not take (origins off the 4-pixel grid): defer is silently off, results unchanged
    odd = ImageSlicer((300, 300, 3), 130, 65, weight="mean")
    mo = TileMerger(odd.target_shape, 1, odd.weight, device=dev, crops=odd.crops, defer=True)
    assert mo._bands is None and not mo._defer_active


def test_deferred_band_merge_fuzz(dev):
    """Random block-aligned geometries, view groups, reductions, channel counts and batch sizes: deferred == incremental, bit
    for bit (where the geometry is not deferrable the merger must fall back silently and still agree)."""
    rng = np.random.default_rng(77)
    ran = 0
    for case in range(14):
        th = int(rng.choice([64, 128, 192, 256]))
        square = bool(rng.integers(0, 2))
        tw = th if square else int(rng.choice([64, 128, 256]))
        sy = int(rng.choice([s for s in (32, 64, 96, 128, 192, 256) if s <= th]))
        sx = int(rng.choice([s for s in (64, 128, 192, 256) if s <= tw]))
        shape = (int(rng.integers(th, 4 * th)), int(rng.integers(tw, 4 * tw)))
        group = str(rng.choice(["d4", "d2", "flips", "fliplr", "flipud"] if th == tw else ["d2", "flips", "fliplr", "flipud"]))
        reduction = str(rng.choice(["mean", "sum", "gmean"]))
        C = int(rng.integers(1, 5))
        bs = int(rng.integers(1, 10))
        try:
            m = _deferred_case(dev, shape, (th, tw), (sy, sx), C, group, reduction, torch.float32, bs, images=1, seed=case)
        except AssertionError as e:
            raise AssertionError(f"case {case}: shape={shape} tile={(th, tw)} step={(sy, sx)} C={C} {group}/{reduction} bs={bs}: {e}") from e
        ran += int(m._bands is not None and m._bands_done > 0)
    assert ran >= 8      # most of the cases really took the deferred path


def test_merge_band_c_abi_contract(dev):
    """ptb_merge_band through ctypes: a band of plain (identity view) tiles equals the oracle's merge rows; tiles that do not
    cover the band, too many tiles and misaligned origins are refused with the documented codes, nothing is launched."""
    from pytorch_toolbelt_amd import _native as N

    lib = N.load()
    C, th, tw, H, W = 2, 64, 64, 128, 256
    w = TO.pyramid_window(tw, th)[0].astype(np.float32)
    weight = torch.from_numpy(w).to(dev)
    xs = np.array([0, 32, 64, 128, 192], dtype=np.int64)
    ys = np.array([0, 0, 0, 0, 0], dtype=np.int64)
    tiles = torch.randn((5, C, th, tw), device=dev)
    merged = torch.full((C, H, W), float("nan"), device=dev)
    norm_full = torch.zeros((1, H, W), device=dev)
    for x in xs:
        norm_full[0, 0:th, x:x + tw] += weight
    src = np.array([tiles[i].data_ptr() for i in range(5)], dtype=np.uint64)
    vs = np.full(5, C * th * tw, dtype=np.int64)
    views = N.int_array([N.IDENT])

    def call(n=5, y0=0, y1=64, xs_=xs, ys_=ys, src_=src):
        with N.on_device(dev):
            return lib.ptb_merge_band(merged.data_ptr(), norm_full.data_ptr(), weight.data_ptr(), src_.ctypes.data, vs.ctypes.data, N.F32, 1,
                                      views, N.RED_SUM, xs_.ctypes.data, ys_.ctypes.data, n, C, th, tw, H, W, y0, y1, N.stream_ptr(dev))

    assert call() == 0
    st = TO.merger_new((H, W), C, w)
    TO.merger_integrate(st, tiles.cpu().numpy(), np.stack([xs, ys, np.full(5, tw), np.full(5, th)], axis=1))
    want = TO.merger_merge(st)
    got = merged.cpu().numpy()
    cover = np.zeros(W, dtype=bool)
    for x in xs:
        cover[x:x + tw] = True
    assert np.array_equal(got[:, :64, cover], want[:, :64, cover])
    assert np.isnan(got[:, 64:]).all() and np.isnan(got[:, :64, ~cover]).all()      # nothing outside the band / the tiles is touched
    assert call(y0=0, y1=96) == -1                                                  # PTB_EINVAL: the tiles end at row 64
    assert call(y0=32, y1=64) == 0                                                  # a sub-band is fine
    bad_y = ys.copy(); bad_y[2] = 32
    assert call(ys_=bad_y) == -1                                                    # tile 2 does not cover rows 0..32
    odd = xs.copy(); odd[1] = 34
    assert call(xs_=odd) == -2                                                      # PTB_EUNSUPPORTED: off the 4-pixel grid
    many_x = np.zeros(49, dtype=np.int64); many_y = np.zeros(49, dtype=np.int64)
    many_src = np.full(49, src[0], dtype=np.uint64)
    global_vs = np.full(49, C * th * tw, dtype=np.int64)
    with N.on_device(dev):
        rc = lib.ptb_merge_band(merged.data_ptr(), norm_full.data_ptr(), weight.data_ptr(), many_src.ctypes.data, global_vs.ctypes.data, N.F32, 1,
                                views, N.RED_SUM, many_x.ctypes.data, many_y.ctypes.data, 49, C, th, tw, H, W, 0, 64, N.stream_ptr(dev))
    assert rc == -2                                                                 # more than 48 tiles in one band
    five = np.zeros(5, dtype=np.int64)
    assert call(xs_=five) == -2                                                     # five tiles over one pixel (MAX_COVER = 4)
    out_of_map = xs.copy(); out_of_map[4] = 224
    assert call(xs_=out_of_map) == -4                                               # PTB_EBOUNDS


def test_reference_default_device_and_loud_fallbacks(dev):
    """The device the caller names decides: TileMerger(shape, C, weight) with the reference's default device="cpu" is the host
    merger (accumulators on the CPU, CUDA batches are moved there like the reference does, tiles.py:330-335), device="cuda" the
    HIP one, and the two agree; a defer / crops request the fast kernels cannot honour is announced instead of silently degraded,
    and `merger.mode` tells which path is live."""
    import warnings

    from pytorch_toolbelt_amd.inference import tiles as T

    T._warned.clear()
    s = T.ImageSlicer((400, 360, 3), 128, 64, weight="pyramid")
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        m = T.TileMerger(s.target_shape, 2, s.weight)
        assert m.device.type == "cpu" and m.mode == "host" and isinstance(m, T.TileMerger) and m.image.dtype == torch.float32
        hip = T.TileMerger(s.target_shape, 2, s.weight, device=dev)
        assert hip.device.type == "cuda" and hip.mode == "incremental"
        yb = torch.rand((len(s.crops), 2, 128, 128), device=dev)
        m.integrate_batch(yb, s.crops)          # CUDA predictions into the host merger: moved, like the reference
        hip.integrate_batch(yb, s.crops)
        assert m.merge().device.type == "cpu" and torch.equal(m.merge(), hip.merge().cpu())
        with pytest.raises(RuntimeError, match="no CPU"):
            T.CudaTileMerger(s.target_shape, 2, s.weight, device="cpu")
        d = T.TileMerger(s.target_shape, 2, s.weight, device=dev, crops=s.crops, defer=True)
    assert d.mode == "deferred bands"
    # off the 4-pixel grid: neither planned nor deferred -> one warning each, ordinary path, same numbers
    s2 = T.ImageSlicer((100, 100, 3), 51, 26, weight="pyramid")
    with pytest.warns(RuntimeWarning, match="block grid"):
        p2 = T.TileMerger(s2.target_shape, 1, s2.weight, device=dev, crops=s2.crops, defer=True)
    assert p2.mode == "incremental"
    tiles = torch.rand((len(s2.crops), 1, 51, 51), device=dev)
    p2.integrate_batch(tiles, s2.crops)
    plain = T.TileMerger(s2.target_shape, 1, s2.weight, device=dev)
    plain.integrate_batch(tiles, s2.crops)
    assert torch.equal(p2.merge(), plain.merge())
    # a deviation from the planned sequence leaves deferred mode with a warning (results stay those of the plain merger)
    T._warned.clear()
    y = torch.rand((len(s.crops), 2, 128, 128), device=dev)
    order = np.arange(len(s.crops))[::-1].copy()
    with pytest.warns(RuntimeWarning, match="deviates from the planned"):
        d.integrate_batch(y[order], s.crops[order])
    assert d.mode == "incremental"
    ref = T.TileMerger(s.target_shape, 2, s.weight, device=dev)
    ref.integrate_batch(y[order], s.crops[order])
    assert torch.equal(d.merge(), ref.merge())


@pytest.mark.parametrize("rows", [64, 128, 320, 1 << 20])
def test_deferred_launch_grouping_is_bit_identical(rows, dev):
    """The rows merged per launch (defer_rows) only change how many launches an image takes, never a bit of the result; pixels no
    tile covers come out as NaN (0 / 0) exactly like the reference's merge()."""
    from pytorch_toolbelt_amd.inference.tiles import ImageSlicer, TileMerger

    slicer = ImageSlicer((700, 520, 3), 128, 64, weight="pyramid")
    crops, C = slicer.crops, 3
    y = torch.randn((8 * len(crops), C, 128, 128), device=dev)
    ref = TileMerger(slicer.target_shape, C, slicer.weight, device=dev)
    m = TileMerger(slicer.target_shape, C, slicer.weight, device=dev, crops=crops, defer=True, defer_rows=rows)
    n = len(crops)
    for b0 in range(0, n, 5):
        b1 = min(n, b0 + 5)
        batch = torch.cat([y[k * n + b0:k * n + b1] for k in range(8)])
        ref.integrate_batch_deaugment(batch, crops[b0:b1], group="d4", reduction="mean")
        m.integrate_batch_deaugment(batch, crops[b0:b1], group="d4", reduction="mean")
    assert m.mode == "deferred bands" and m._bands_done == len(m._bands.bands) and not m._held
    assert torch.equal(m.merge(), ref.merge())
    # a crop list with holes: rows 0..63 and a column strip are never covered
    H, W = 320, 512
    holes = np.array([[64, 64, 128, 128], [256, 64, 128, 128], [64, 192, 128, 128]])
    w = slicer.weight
    t = torch.randn((3, 2, 128, 128), device=dev)
    a = TileMerger((H, W), 2, w, device=dev)
    a.integrate_batch(t, holes)
    want = a.merge()
    for gridded in (np.array([[64, 64, 128, 128], [256, 64, 128, 128], [64, 192, 128, 128]]),):
        b = TileMerger((H, W), 2, w, device=dev, crops=gridded, defer=True, defer_rows=rows)
        if b._bands is None:     # x = 64 is on the 64-column block grid, y = 64 on the 32-row grid: planned
            pytest.fail("geometry should be deferrable")
        b.integrate_batch(t, gridded)
        got = b.merge()
        assert torch.equal(torch.isnan(got), torch.isnan(want)) and bool(torch.isnan(got).any())
        assert torch.equal(torch.nan_to_num(got), torch.nan_to_num(want))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_reference_accumulators_keep_the_reference_bits(dtype, dev):
    """`tiles.set_reference_accumulators(True)` (part of `set_strict_dropin()`): a CUDA merger with float16 / bfloat16 accumulators sums in
    that dtype with the reference's own op sequence (tiles.py:306-308, 334-346) -- its bits, rounding after every `+=` -- instead of the HIP
    merger's float32 sums, under both class names; float32 mergers and the default setting are untouched."""
    import pytorch_toolbelt_amd
    from pytorch_toolbelt_amd.inference import tiles as T
    from pytorch_toolbelt_amd.inference import tta

    slicer = T.ImageSlicer((200, 180, 3), 64, 32, weight="pyramid")
    crops, C = slicer.crops, 2
    g = torch.Generator().manual_seed(4)
    pred = torch.rand((len(crops), C, 64, 64), generator=g).to(dev).to(dtype)
    views = torch.rand((8 * 3, C, 64, 64), generator=g).to(dev).to(dtype)
    # the reference's arithmetic, spelt out with torch ops on the device
    w = torch.from_numpy(np.expand_dims(slicer.weight, 0)).to(dev).to(dtype)
    image = torch.zeros((C, *slicer.target_shape), device=dev, dtype=dtype)
    norm = torch.zeros((1, *slicer.target_shape), device=dev, dtype=dtype)
    for tile, (x, y, tw, th) in zip(pred, crops):
        image[:, y:y + th, x:x + tw] += tile * w
        norm[:, y:y + th, x:x + tw] += w
    want = image / norm
    assert not T.set_reference_accumulators(False)
    plain = T.TileMerger(slicer.target_shape, C, slicer.weight, device=dev, dtype=dtype)
    assert not isinstance(plain, T.HostBackedTileMerger)
    plain.integrate_batch(pred, crops)
    prev = pytorch_toolbelt_amd.set_strict_dropin(True)
    try:
        for cls in (T.TileMerger, T.CudaTileMerger):
            m = cls(slicer.target_shape, C, slicer.weight, device=dev, dtype=dtype)
            assert isinstance(m, T.HostBackedTileMerger) and m.mode == "host" and m.image.dtype == dtype and m.image.is_cuda
            for b0 in range(0, len(crops), 5):
                m.integrate_batch(pred[b0:b0 + 5], crops[b0:b0 + 5])
            got = m.merge()
            assert got.dtype == dtype and torch.equal(got, want) and torch.equal(m.norm_mask, norm)
            m.reset()
            m.integrate_batch_deaugment(views, crops[:3], group="d4", reduction="mean")      # == integrate_batch(d4_image_deaugment(...))
            r = cls(slicer.target_shape, C, slicer.weight, device=dev, dtype=dtype)
            r.integrate_batch(tta.d4_image_deaugment(views), crops[:3])
            assert torch.equal(m.image, r.image)
        assert not isinstance(T.TileMerger(slicer.target_shape, C, slicer.weight, device=dev), T.HostBackedTileMerger)      # float32: the HIP merger
        from pytorch_toolbelt_amd.inference.tiles_3d import HostBackedVolumeMerger, VolumeMerger

        vm = VolumeMerger((8, 8, 8), 1, np.ones((4, 4, 4), dtype=np.float32), device=dev, dtype=dtype)
        assert isinstance(vm, HostBackedVolumeMerger) and vm.volume.dtype == dtype and vm.volume.is_cuda

        class MyMerger(T.TileMerger):          # any other subclass of the HIP merger would sum in float32: refused, like float64
            pass

        with pytest.raises(TypeError, match="accumulators are kept by the torch-op merger"):
            MyMerger(slicer.target_shape, C, slicer.weight, device=dev, dtype=dtype)
    finally:
        pytorch_toolbelt_amd.set_strict_dropin(False)
        tta.set_lazy_deaugment(prev[0]); T.set_auto_plan(prev[1])
    assert not T._REFERENCE_ACCUMULATORS
    # the default (float32 sums, rounded once) is at least as close to the exact blend as the reference's half-precision sums
    exact = (image.double() * 0)
    n64 = torch.zeros_like(norm, dtype=torch.float64)
    w64 = torch.from_numpy(np.expand_dims(slicer.weight, 0)).to(dev)
    for tile, (x, y, tw, th) in zip(pred, crops):
        exact[:, y:y + th, x:x + tw] += tile.double() * w64
        n64[:, y:y + th, x:x + tw] += w64
    exact = exact / n64
    assert float((plain.merge().double() - exact).abs().max()) <= float((want.double() - exact).abs().max()) + 1e-12


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float64])
def test_tile_merger_dtype_argument(dtype, dev):
    """TileMerger(..., dtype=...) (reference tiles.py:295: accumulators of any floating dtype): tile batches of that dtype are
    taken as they are, merge() / image / norm_mask come back in it; the sums themselves are float32."""
    from pytorch_toolbelt_amd.inference.tiles import ImageSlicer, TileMerger
    from pytorch_toolbelt_amd.inference.tiles_3d import VolumeMerger

    slicer = ImageSlicer((200, 180, 3), 64, 32, weight="pyramid")
    crops = slicer.crops
    g = torch.Generator().manual_seed(3)
    pred = torch.rand((len(crops), 2, 64, 64), generator=g)
    m = TileMerger(slicer.target_shape, 2, slicer.weight, device=dev, dtype=dtype)
    ref = TileMerger(slicer.target_shape, 2, slicer.weight, device=dev)
    m.integrate_batch(pred.to(dev).to(dtype), crops)
    ref.integrate_batch(pred.to(dtype).float().to(dev), crops)
    out = m.merge()
    assert out.dtype == dtype and m.image.dtype == dtype and m.norm_mask.dtype == dtype
    tol = {torch.float16: 1e-3, torch.bfloat16: 8e-3, torch.float64: 1e-6}[dtype]
    torch.testing.assert_close(out.float(), ref.merge(), rtol=tol, atol=tol)
    with pytest.raises(TypeError):
        TileMerger(slicer.target_shape, 2, slicer.weight, device=dev, dtype=torch.int32)
    vm = VolumeMerger((8, 8, 8), 1, np.ones((4, 4, 4), dtype=np.float32), device=dev, dtype=dtype)
    assert vm.merge().dtype == dtype
    if dtype == torch.float64:
        # float64 accumulators are float64 SUMS under every name the HIP merger goes by (CudaTileMerger / subclasses fell
        # through to float32 sums cast afterwards)
        from pytorch_toolbelt_amd.inference.tiles import CudaTileMerger, HostBackedTileMerger

        cm = CudaTileMerger(slicer.target_shape, 2, slicer.weight, dtype=torch.float64)
        assert isinstance(cm, HostBackedTileMerger) and cm.mode == "host" and cm.image.dtype == torch.float64 and cm.image.is_cuda
        cm.integrate_batch(pred.to(dev).double(), crops)
        assert torch.equal(cm.merge(), m.merge())
        assert type(CudaTileMerger(slicer.target_shape, 2, slicer.weight)) is CudaTileMerger

        class MyMerger(TileMerger):
            pass

        class MyVolumeMerger(VolumeMerger):
            pass

        with pytest.raises(TypeError, match="float64 accumulators"):
            MyMerger(slicer.target_shape, 2, slicer.weight, device=dev, dtype=torch.float64)
        with pytest.raises(TypeError, match="float64 accumulators"):
            MyVolumeMerger((8, 8, 8), 1, np.ones((4, 4, 4), dtype=np.float32), device=dev, dtype=torch.float64)


@pytest.mark.parametrize("group,reduction,dtype", [("d4", "mean", torch.float32), ("d4", "gmean", torch.float32), ("fliplr", "sum", torch.float32),
                                                   ("d4", "mean", torch.float16)])
def test_band_plan_item_rows_32_and_64_are_bit_identical(group, reduction, dtype, dev):
    """Band plans are made of 64 x 64 work items by default (1024-thread workgroups) or 64 x 32 (ptb_set_tunable key 11): the same
    sums in the same order, so the merged maps must agree bit for bit -- with each other and with the incremental merger."""
    from pytorch_toolbelt_amd import _native as N
    from pytorch_toolbelt_amd.inference.tiles import ImageSlicer, TileMerger
    from pytorch_toolbelt_amd.inference.tta import DEAUGMENT_VIEWS

    slicer = ImageSlicer((900, 700, 3), 256, 128, weight="pyramid")
    crops, C, V = slicer.crops, 3, len(DEAUGMENT_VIEWS[group])
    mergers = {}
    try:
        for rows in (32, 64):
            assert N.load().ptb_set_tunable(11, rows) == 0
            mergers[rows] = TileMerger(slicer.target_shape, C, slicer.weight, device=dev, crops=crops, defer=True)
    finally:
        assert N.load().ptb_set_tunable(11, 64) == 0
    mergers["plain"] = TileMerger(slicer.target_shape, C, slicer.weight, device=dev)
    g = torch.Generator().manual_seed(5)
    for b0 in range(0, len(crops), 7):
        nb = min(7, len(crops) - b0)
        y = (torch.rand((V * nb, C, 256, 256), generator=g) * 0.9 + 0.05).to(dev).to(dtype)
        for m in mergers.values():
            m.integrate_batch_deaugment(y, crops[b0:b0 + nb], group=group, reduction=reduction)
    outs = {k: m.merge() for k, m in mergers.items()}
    for rows in (32, 64):
        assert mergers[rows]._bands is not None and mergers[rows]._bands_done == len(mergers[rows]._bands.bands)
        assert torch.equal(outs[rows], outs["plain"]), rows


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("group,reduction", [("d4", "mean"), ("d4", "gmean"), ("d2", "mean"), ("fliplr", "sum"), (None, None)])
def test_band_plan_prefetch_is_bit_identical(group, reduction, dtype, dev):
    """The band plan kernel requests covering tile e + 1 before it finishes tile e (ptb_set_tunable key 21: 0 never, 1 two-byte sources,
    2 every source type -- the default): the same loads, the same sums in the same order, for 64- and 32-row items, on geometries with
    ragged last rows / columns of items, and equal to the incremental merger."""
    from pytorch_toolbelt_amd import _native as N
    from pytorch_toolbelt_amd.inference.tiles import ImageSlicer, TileMerger
    from pytorch_toolbelt_amd.inference.tta import DEAUGMENT_VIEWS

    lib = N.load()
    for shape, tile, step, C in (((900, 700), 256, 128, 3), ((520, 392), 128, 96, 2)):
        slicer = ImageSlicer(shape + (3,), tile, step, weight="pyramid")
        crops, V = slicer.crops, len(DEAUGMENT_VIEWS[group]) if group else 1
        g = torch.Generator().manual_seed(11)
        ys = [(torch.rand((V * min(5, len(crops) - b0), C, tile, tile), generator=g) * 0.9 + 0.05).to(dev).to(dtype)
              for b0 in range(0, len(crops), 5)]

        def image(m):
            for i, y in enumerate(ys):
                cr = crops[5 * i:5 * i + 5]
                if group:
                    m.integrate_batch_deaugment(y, cr, group=group, reduction=reduction)
                else:
                    m.integrate_batch(y, cr)
            return m.merge().clone()

        want = image(TileMerger(slicer.target_shape, C, slicer.weight, device=dev, auto_plan=False))
        try:
            for rows in (64, 32):
                assert lib.ptb_set_tunable(11, rows) == 0
                m = TileMerger(slicer.target_shape, C, slicer.weight, device=dev, crops=crops, defer=True)
                for pf in (0, 1, 2):
                    assert lib.ptb_set_tunable(21, pf) == 0
                    m.reset()
                    assert torch.equal(image(m), want), (shape, rows, pf)
        finally:
            assert lib.ptb_set_tunable(11, 64) == 0 and lib.ptb_set_tunable(21, 2) == 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_channel_loop_of_the_identity_view_is_bit_identical(dtype, dev):
    """The loop without TTA on a deferred merger (ptb_set_tunable key 27): one workgroup per (item, channel) [0] and one workgroup per item
    walking the channels [1, default] write the same bits as the incremental merger -- ragged right / bottom items, rectangular tiles,
    C = 1 and C = 5, two images per merger."""
    from pytorch_toolbelt_amd import _native as N
    from pytorch_toolbelt_amd.inference.tiles import ImageSlicer, TileMerger

    lib = N.load()
    for shape, tile, step, C in (((1100, 900), 512, 256, 5), ((1100, 1300), 512, 256, 1), ((900, 700), 256, 128, 3), ((520, 392), 128, 96, 2),
                                 ((700, 1000), (256, 512), (128, 256), 2)):
        slicer = ImageSlicer(shape + (3,), tile, step, weight="pyramid")
        crops = slicer.crops
        th, tw = slicer.tile_size
        g = torch.Generator().manual_seed(5)
        ys = [(torch.rand((min(6, len(crops) - b0), C, th, tw), generator=g) * 2 - 1).to(dev).to(dtype) for b0 in range(0, len(crops), 6)]

        def image(m):
            for i, y in enumerate(ys):
                m.integrate_batch(y, crops[6 * i:6 * i + 6])
            return m.merge().clone()

        want = image(TileMerger(slicer.target_shape, C, slicer.weight, device=dev, auto_plan=False))
        try:
            for mode in (0, 1):
                assert lib.ptb_set_tunable(27, mode) == 0
                m = TileMerger(slicer.target_shape, C, slicer.weight, device=dev, crops=crops, defer=True)
                for order in (0, 1, 2, 0):      # (workgroup orders of the band kernel, key 10: the XCD order pads its grid to a multiple of 8)
                    assert lib.ptb_set_tunable(10, order) == 0
                    m.reset()
                    got = image(m)
                    assert m._bands is not None and m._bands_done == len(m._bands.bands), (shape, mode, order)
                    assert torch.equal(got, want), (shape, mode, order)
        finally:
            assert lib.ptb_set_tunable(27, 1) == 0 and lib.ptb_set_tunable(10, 0) == 0
