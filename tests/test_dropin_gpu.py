"""GPU: the reference's LITERAL loop takes the fused / planned kernels (README.md:201-226; inference/tta.py:442-467 feeding
inference/tiles.py:321-346), and the deferred merger enforces its contract.

* ``tta.*_image_deaugment`` returns a lazy handle (inference/_lazy.py) that ``TileMerger.integrate_batch`` fuses into its launch
  and that behaves like the evaluated tensor everywhere else;
* ``TileMerger(shape, C, weight)`` without ``crops=`` plans itself from the second image of a geometry on -- into deferred bands
  where the geometry, the byte budget and what it saw of the model's outputs allow, else into planned blocks -- and never raises for
  something the caller did not opt into;
* ``TileMerger(defer=True)`` refuses a batch that lives in a held batch's memory and notices in-place edits of held batches.

Everything is compared bit for bit with the eager, unplanned path of the same library (itself pinned to the reference's goldens in
tests/test_tiles_gpu.py) and against the numpy oracle within 1e-5 (BASELINE.json north_star)."""
import warnings

import numpy as np
import pytest
import torch

from oracle import tiles_oracle as TO
from oracle import tta_oracle as AO

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture()
def lazy():
    from pytorch_toolbelt_amd.inference import _lazy

    prev = _lazy.set_enabled(True)
    yield _lazy
    _lazy.set_enabled(prev)


@pytest.fixture()
def autoplan():
    from pytorch_toolbelt_amd.inference import tiles

    prev = tiles.set_auto_plan(True)
    tiles._auto.clear()
    tiles._warned.clear()
    yield tiles
    tiles._auto.clear()
    tiles.set_auto_plan(prev)


@pytest.fixture(params=["deferred bands", "planned"])
def flavour(request, monkeypatch):
    """What a self-planned merger turns into: deferred bands (default), or -- with a byte budget no image fits -- planned blocks."""
    if request.param == "planned":
        monkeypatch.setenv("PTB_DEFER_BYTES", "1")
    return request.param


GROUPS = {"fliplr": 2, "flipud": 2, "flips": 3, "d2": 4, "d4": 8}


def _eager(fn, x, **kw):
    from pytorch_toolbelt_amd.inference import _lazy

    prev = _lazy.set_enabled(False)
    try:
        out = fn(x, **kw)
    finally:
        _lazy.set_enabled(prev)
    assert type(out) is torch.Tensor
    return out


# ------------------------------------------------------------------------------------------------ lazy handles
@pytest.mark.parametrize("group", sorted(GROUPS))
@pytest.mark.parametrize("reduction", ["mean", "sum", "gmean", "logodd"])
def test_lazy_deaugment_equals_eager(group, reduction, dev, lazy):
    from pytorch_toolbelt_amd.inference import tta

    fn = getattr(tta, f"{group}_image_deaugment")
    V = GROUPS[group]
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.rand((V * 3, 2, 64, 64), device=dev, generator=g) * 0.9 + 0.05
    want = _eager(fn, x, reduction=reduction)
    ev0 = lazy.evaluations
    y = fn(x, reduction=reduction)
    assert type(y) is lazy.LazyDeaugment and isinstance(y, torch.Tensor)
    assert y.shape == want.shape and y.dtype == want.dtype and y.device == want.device and len(y) == 3 and y.dim() == 4
    assert lazy.evaluations == ev0, "metadata must not evaluate the handle"
    assert torch.equal(y, want) and lazy.evaluations == ev0 + 1
    assert torch.equal(y + 1, want + 1) and torch.equal(y[1], want[1]) and torch.equal(torch.cat([y, y]), torch.cat([want, want]))
    assert np.array_equal(y.cpu().numpy(), want.cpu().numpy()) and lazy.evaluations == ev0 + 1
    # against the oracle, like the eager tests
    assert np.abs(y.cpu().numpy() - AO.image_deaugment(x.cpu().numpy(), group, reduction)).max() <= 1e-5


def test_lazy_handle_in_place_and_consumers(dev, lazy):
    from pytorch_toolbelt_amd.inference import tta
    from pytorch_toolbelt_amd.inference.ensembling import Ensembler  # noqa: F401  (a consumer module of this package imports fine)

    x = torch.rand((8 * 2, 3, 32, 32), device=dev)
    want = _eager(tta.d4_image_deaugment, x)
    y = tta.d4_image_deaugment(x)
    y.mul_(2.0)                               # in place on the handle == in place on the evaluated tensor
    assert torch.equal(y, want * 2)
    z = tta.d4_image_deaugment(x)
    assert torch.equal(torch.nn.functional.interpolate(z, scale_factor=2, mode="nearest"), torch.nn.functional.interpolate(want, scale_factor=2, mode="nearest"))
    assert torch.equal(tta.fliplr_image_augment(z), tta.fliplr_image_augment(want))     # our own kernels take it (data_ptr through the handle)
    assert z.data_ptr() == z._evaluate().data_ptr() and "tensor(" in repr(z)
    assert float(tta.d4_image_deaugment(x).sum()) == pytest.approx(float(want.sum()), rel=1e-6)
    # autograd-shaped calls, calls without a string reduction and float64 sources are evaluated on the spot
    xg = x.clone().requires_grad_(True)
    yg = tta.d4_image_deaugment(xg)
    assert type(yg) is torch.Tensor and yg.requires_grad
    yg.sum().backward()
    assert xg.grad is not None
    xh = x.half()
    hh = tta.d4_image_deaugment(xh)           # round 6: half-precision model outputs (torch.autocast) get a handle too -- a half tensor in every respect
    assert type(hh) is lazy.LazyDeaugment and hh.dtype == torch.float16 and torch.equal(hh, _eager(tta.d4_image_deaugment, xh))
    assert type(tta.d4_image_deaugment(x.double())) is torch.Tensor
    assert type(tta.d4_image_deaugment(x, reduction=None)) is torch.Tensor
    with torch.no_grad():
        assert type(tta.d4_image_deaugment(xg)) is lazy.LazyDeaugment


def test_lazy_handle_behaves_like_the_eager_tensor_for_unusual_consumers(dev, lazy):
    """copy / pickle / torch.save, .data, out=, item assignment, DLPack (both protocols), __cuda_array_interface__, autograd on the
    handle, compile, vmap, ...: every consumer sees what it would see with the eager result (inference/tta.py:442-467)."""
    import copy
    import io
    import pickle

    from pytorch_toolbelt_amd.inference import tta

    x = torch.rand((16, 3, 32, 32), device=dev)
    want = _eager(tta.d4_image_deaugment, x)

    def fresh():
        return tta.d4_image_deaugment(x)

    def same(a, b):
        return torch.equal(torch.as_tensor(a).to(dev).float(), torch.as_tensor(b).to(dev).float())

    def raises_same(fn, y):
        def kind(t):
            try:
                fn(t)
                return None
            except Exception as e:  # noqa: BLE001
                return type(e)
        return kind(y) == kind(want)

    def inference():
        with torch.inference_mode():
            return tta.d4_image_deaugment(x.clone()) + 0

    ops = {
        "deepcopy": lambda y: same(copy.deepcopy(y), want),
        "copy": lambda y: same(copy.copy(y), want),
        "pickle": lambda y: same(pickle.loads(pickle.dumps(y)), want),
        "torch.save": lambda y: (lambda b: (torch.save(y, b), b.seek(0), same(torch.load(b), want))[-1])(io.BytesIO()),
        ".data": lambda y: same(y.data, want),
        "detach": lambda y: same(y.detach(), want),
        "setitem": lambda y: (y.__setitem__(0, 1.0), same(y[1:], want[1:]) and float(y[0].min()) == 1.0)[-1],
        "out=": lambda y: (torch.add(want, 1.0, out=y), same(y, want + 1.0))[-1],
        "stack": lambda y: same(torch.stack([y, fresh()]).mean(0), want),
        "cat": lambda y: same(torch.cat([y, y])[2:], want),
        "numpy": lambda y: same(torch.from_numpy(y.cpu().numpy()), want) and raises_same(lambda t: t.numpy(), fresh()),
        "tolist": lambda y: y[0, 0, 0].tolist() == want[0, 0, 0].tolist(),
        "untyped_storage": lambda y: y.untyped_storage().size() == want.untyped_storage().size(),
        "view": lambda y: same(y.view(-1), want.view(-1)) and same(fresh().mT, want.mT),
        "iter": lambda y: same(next(iter(y)), want[0]) and len(y) == len(want),
        "hash": lambda y: isinstance(hash(y), int) and bool((y == want).all()),
        "requires_grad_": lambda y: (y.requires_grad_(True), (y * 2).sum().backward(), y.requires_grad and y.is_leaf and same(y.grad, torch.full_like(want, 2.0)))[-1],
        "to_dlpack": lambda y: same(torch.utils.dlpack.from_dlpack(torch.utils.dlpack.to_dlpack(y)), want),
        "from_dlpack": lambda y: same(torch.from_dlpack(y), want),
        "cuda_array_interface": lambda y: y.__cuda_array_interface__["shape"] == tuple(want.shape),
        "as_tensor": lambda y: same(torch.as_tensor(y), want) and same(fresh().clone(), want),
        "record_stream": lambda y: (y.record_stream(torch.cuda.current_stream()), same(y, want))[-1],
        "nbytes": lambda y: y.nbytes == want.nbytes and y.itemsize == want.itemsize and isinstance(y, torch.Tensor) and torch.is_tensor(y),
        "module": lambda y: tuple(torch.nn.Conv2d(3, 3, 1).to(dev)(y).shape) == tuple(want.shape),
        "index": lambda y: same(y[torch.tensor([1, 0], device=dev)], want[torch.tensor([1, 0], device=dev)]) and same(want[(fresh()[:, 0, 0, 0] > 2).long()], want[[0, 0]]),
        "pin_memory": lambda y: raises_same(lambda t: t.pin_memory(), y),
        "format": lambda y: f"{y[0, 0, 0, 0]:.3f}" == f"{want[0, 0, 0, 0]:.3f}",
        "non_blocking": lambda y: (lambda c: (torch.cuda.synchronize(), same(c, want))[-1])(y.to("cpu", non_blocking=True)),
        "compile": lambda y: same(torch.compile(lambda t: t * 2, backend="eager")(y), want * 2),
        "vmap": lambda y: same(torch.vmap(lambda t: t.sum())(y), want.sum(dim=(1, 2, 3))),
        "inference_mode": lambda y: same(inference(), want),
    }
    failed = []
    for name, op in ops.items():
        y = fresh()
        assert type(y) is lazy.LazyDeaugment
        try:
            ok = op(y)
        except Exception as e:  # noqa: BLE001
            ok = False
            name = f"{name} ({type(e).__name__}: {e})"
        if not ok:
            failed.append(name)
    assert not failed, failed


def test_lazy_source_edit_is_reported_and_budget_bounds_memory(dev, lazy):
    from pytorch_toolbelt_amd.inference import tta

    x = torch.rand((2 * 2, 1, 16, 16), device=dev)
    y = tta.fliplr_image_deaugment(x)
    x.add_(1.0)
    with pytest.raises(RuntimeError, match="modified in place"):
        y + 0
    with torch.inference_mode():              # no version counters there: an edit could not be noticed, so no handle is handed out
        xi = torch.rand((2 * 2, 1, 16, 16), device=dev)
        yi = tta.fliplr_image_deaugment(xi)
        assert type(yi) is torch.Tensor and torch.equal(yi, _eager(tta.fliplr_image_deaugment, xi))
    old = lazy._BUDGET
    try:
        src = torch.rand((2 * 4, 2, 32, 32), device=dev)
        lazy._BUDGET = lazy._pending_bytes + 3 * src.numel() * 4
        hs = [tta.fliplr_image_deaugment(src) for _ in range(6)]
        assert [h._value is not None for h in hs] == [True, True, True, False, False, False]
        want = _eager(tta.fliplr_image_deaugment, src)
        assert all(torch.equal(h, want) for h in hs)
    finally:
        lazy._BUDGET = old


def _run_image(merger, outputs, crops, batch, literal=True, group="d4", reduction="mean"):
    from pytorch_toolbelt_amd.inference import tta

    V = GROUPS[group]
    n = len(crops)
    fn = getattr(tta, f"{group}_image_deaugment")
    for b0 in range(0, n, batch):
        b1 = min(n, b0 + batch)
        y = torch.cat([outputs[k * n + b0:k * n + b1] for k in range(V)])       # chunk-major model output of the batch
        if literal:
            merger.integrate_batch(fn(y, reduction=reduction), crops[b0:b1])
        else:
            merger.integrate_batch_deaugment(y, crops[b0:b1], group=group, reduction=reduction)
    return merger.merge()


def _oracle_image(geom, C, w, outputs, batch, group="d4", reduction="mean", keep=None):
    V = GROUPS[group]
    crops = geom["crops"]
    n = len(crops)
    st = TO.merger_new(geom["target_shape"], C, w)
    onp = outputs.cpu().numpy()
    for b0 in range(0, n, batch):
        b1 = min(n, b0 + batch)
        y = np.concatenate([onp[k * n + b0:k * n + b1] for k in range(V)])
        TO.merger_integrate(st, AO.image_deaugment(y, group, reduction), crops[b0:b1])
    return TO.merger_merge(st)


@pytest.mark.parametrize("group,reduction,shape,tile,step,C,batch", [
    ("d4", "mean", (500, 420), 128, 64, 4, 8),
    ("d4", "gmean", (300, 300), 64, 32, 2, 5),
    ("d2", "mean", (200, 330), (64, 128), (32, 64), 3, 4),
    ("fliplr", "sum", (130, 170), (52, 36), (20, 12), 2, 9),     # off the vector grid: scalar kernels
])
def test_literal_loop_is_fused_and_bit_identical(group, reduction, shape, tile, step, C, batch, dev, lazy):
    from pytorch_toolbelt_amd.inference.tiles import TileMerger

    tile2 = (tile, tile) if isinstance(tile, int) else tile
    geom = TO.slicer_geometry(shape, tile, step)
    w = TO.pyramid_window(*tile2)[0]
    n, V = len(geom["crops"]), GROUPS[group]
    g = torch.Generator(device=dev).manual_seed(11)
    outputs = torch.rand((V * n, C, *tile2), device=dev, generator=g) * 0.9 + 0.05
    prev = lazy.set_enabled(False)
    want = _run_image(TileMerger(geom["target_shape"], C, w, device=dev, auto_plan=False), outputs, geom["crops"], batch, group=group, reduction=reduction)
    lazy.set_enabled(prev)
    f0, e0 = lazy.fused, lazy.evaluations
    got = _run_image(TileMerger(geom["target_shape"], C, w, device=dev, auto_plan=False), outputs, geom["crops"], batch, group=group, reduction=reduction)
    assert lazy.fused - f0 == (n + batch - 1) // batch and lazy.evaluations == e0, "the literal calls did not take the fused launch"
    assert torch.equal(got, want)
    assert np.nanmax(np.abs(got.cpu().numpy() - _oracle_image(geom, C, w, outputs, batch, group, reduction))) <= 1e-5
    # the same through a planned and a deferred merger
    for kw in ({"crops": geom["crops"]}, {"crops": geom["crops"], "defer": True}):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = TileMerger(geom["target_shape"], C, w, device=dev, **kw)
        assert torch.equal(_run_image(m, outputs, geom["crops"], batch, group=group, reduction=reduction), want)


# ------------------------------------------------------------------------------------------------ self-planning mergers
def test_new_merger_per_image_plans_itself_from_the_second_image(dev, lazy, autoplan, flavour):
    TileMerger = autoplan.TileMerger
    geom = TO.slicer_geometry((500, 420), 128, 64)
    crops, C, batch = geom["crops"], 3, 8
    w = TO.pyramid_window(128, 128)[0]
    n = len(crops)
    modes = []
    for image in range(4):
        g = torch.Generator(device=dev).manual_seed(100 + image)
        outputs = torch.randn((8 * n, C, 128, 128), device=dev, generator=g)
        exact = _run_image(TileMerger(geom["target_shape"], C, w, device=dev, auto_plan=False), outputs, crops, batch, literal=False)
        m = TileMerger(geom["target_shape"], C, w, device=dev)          # the reference's constructor call, once per image
        modes.append(m.mode)
        got = _run_image(m, outputs, crops, batch)
        assert torch.equal(got, exact), f"image {image} ({modes[-1]})"
        assert m.mode == modes[-1]
        if m.mode == "deferred bands":
            assert m._deferred.soft and m._deferred.complete and not len(m._held) and m._plan.pos == n
        elif m._plan is not None:
            assert m._plan.done.all() and m._plan.pos == n
    assert modes == ["incremental", flavour, flavour, flavour]
    assert np.nanmax(np.abs(got.cpu().numpy() - _oracle_image(geom, C, w, outputs, batch))) <= 1e-5
    if flavour == "deferred bands":   # band plans are pooled per geometry: mergers that follow each other share them
        from pytorch_toolbelt_amd.inference import _merge_modes as MM

        ent = next(iter(MM.auto_cache.values()))
        del m
        assert 1 <= len(ent.pool) <= 2 and ent.rows


def test_literal_loop_from_two_threads_on_two_streams(dev, lazy, autoplan):
    """Two threads, each on its own stream, run the literal loop on the SAME geometry at the same time (a new merger per image: they
    share the module's plan cache, the window cache and the lazy bookkeeping); every image equals the serial, eager result."""
    import threading

    geom = TO.slicer_geometry((700, 900), (128, 128), (64, 64))
    crops, w = geom["crops"], TO.pyramid_window(128, 128)[0]
    C, images = 3, 8
    outs = [torch.rand((8 * len(crops), C, 128, 128), device=dev) for _ in range(images)]
    from pytorch_toolbelt_amd.inference.tiles import TileMerger

    lazy.set_enabled(False)
    autoplan.set_auto_plan(False)
    serial = [_run_image(TileMerger(geom["target_shape"], C, w, device=dev), o, crops, 8) for o in outs]
    lazy.set_enabled(True)
    autoplan.set_auto_plan(True)
    torch.cuda.synchronize()
    results, errors = {}, []

    def worker(t):
        try:
            s = torch.cuda.Stream(device=dev)
            s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s):
                for rep in range(3):
                    for k in range(t, images, 2):
                        results[(rep, k)] = _run_image(TileMerger(geom["target_shape"], C, w, device=dev), outs[k], crops, 8)
            s.synchronize()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors
    assert len(results) == 3 * images
    for (rep, k), got in results.items():
        assert torch.equal(torch.nan_to_num(got, nan=-7.0), torch.nan_to_num(serial[k], nan=-7.0)), (rep, k)


def test_reset_flow_plans_itself_and_survives_deviations(dev, lazy, autoplan, flavour):
    TileMerger = autoplan.TileMerger
    geom = TO.slicer_geometry((384, 384), 128, 64)
    crops, C, batch = geom["crops"], 2, 4
    w = TO.pyramid_window(128, 128)[0]
    n = len(crops)
    g = torch.Generator(device=dev).manual_seed(5)
    outputs = torch.randn((8 * n, C, 128, 128), device=dev, generator=g)
    exact = _run_image(TileMerger(geom["target_shape"], C, w, device=dev, auto_plan=False), outputs, crops, batch, literal=False)
    m = TileMerger(geom["target_shape"], C, w, device=dev)
    assert m.mode == "incremental" and torch.equal(_run_image(m, outputs, crops, batch), exact)
    m.reset()
    assert m.mode == flavour and torch.equal(_run_image(m, outputs, crops, batch), exact)
    # an image that skips tiles (content-dependent loops do): correct, and planning asks for more evidence afterwards
    keep = np.array([i for i in range(n) if i not in (3, 7, 8)])
    sub = torch.cat([outputs[k * n + keep] for k in range(8)])
    sub_geom = {"crops": crops[keep], "target_shape": geom["target_shape"]}
    exact_sub = _run_image(TileMerger(geom["target_shape"], C, w, device=dev, auto_plan=False), sub, crops[keep], batch, literal=False)
    m.reset()
    assert m.mode == flavour
    got = _run_image(m, sub, crops[keep], batch)
    assert m.mode == "incremental"                       # it left the plan at the first deviating batch
    assert torch.equal(got, exact_sub)
    assert np.nanmax(np.abs(got.cpu().numpy() - _oracle_image(sub_geom, C, w, sub, batch))) <= 1e-5
    m.reset()
    assert m.mode == "incremental"                       # one odd image seen: not yet planned from it
    assert torch.equal(_run_image(m, outputs, crops, batch), exact)
    m.reset()
    assert m.mode == "incremental" and torch.equal(_run_image(m, outputs, crops, batch), exact)
    m.reset()
    assert m.mode == flavour and torch.equal(_run_image(m, outputs, crops, batch), exact)


def test_self_planned_merger_hands_out_accumulators(dev, lazy, autoplan, flavour):
    """Reading ``image`` after a self-planned merger has turned (part of) the image into results.  Planned blocks: the accumulators are
    COMPLETE and EXACT (a self-planned merger stores the weighted sum of a block next to its merged value, PTB_PLANNED_KEEP_SUMS) --
    array_equal with the unplanned path.  Deferred bands: no accumulator was ever stored, merged rows come back as merged * norm_mask
    (one float32 rounding away from the sequential sums).  Either way nothing raises, it is said once, and the geometry stays on the
    ordinary path afterwards."""
    TileMerger = autoplan.TileMerger
    geom = TO.slicer_geometry((256, 256), 128, 64)
    crops, C, batch = geom["crops"], 2, 3
    w = TO.pyramid_window(128, 128)[0]
    n = len(crops)
    outputs = torch.randn((8 * n, C, 128, 128), device=dev, generator=torch.Generator(device=dev).manual_seed(9))
    ref = TileMerger(geom["target_shape"], C, w, device=dev, auto_plan=False)
    exact = _run_image(ref, outputs, crops, batch, literal=False)
    exact_image, exact_norm = ref.image.clone(), ref.norm_mask.clone()
    _run_image(TileMerger(geom["target_shape"], C, w, device=dev), outputs, crops, batch)        # image 1: remembered
    m = TileMerger(geom["target_shape"], C, w, device=dev)
    assert m.mode == flavour
    got = _run_image(m, outputs, crops, batch)
    assert torch.equal(got, exact)
    if flavour == "planned":
        with pytest.warns(RuntimeWarning, match="accumulators are complete"):
            img = m.image
        assert torch.equal(img, exact_image) and torch.equal(m.norm_mask, exact_norm)
        assert torch.equal(m.merge(), exact)
    else:
        with pytest.warns(RuntimeWarning, match="continuing on the ordinary accumulate"):
            img = m.image
        torch.testing.assert_close(img, exact_image, rtol=2e-7, atol=1e-9)
        assert torch.equal(m.norm_mask, exact_norm)
        torch.testing.assert_close(m.merge(), exact, rtol=3e-7, atol=1e-9)
    assert m.mode == "incremental"
    m2 = TileMerger(geom["target_shape"], C, w, device=dev)
    assert m2.mode == "incremental"                      # this geometry's user reads accumulators: exact path from now on
    assert torch.equal(_run_image(m2, outputs, crops, batch), exact) and torch.equal(m2.image, exact_image)


def test_self_planned_merger_takes_an_extra_tile(dev, lazy, autoplan, flavour):
    TileMerger = autoplan.TileMerger
    geom = TO.slicer_geometry((256, 256), 128, 64)
    crops, C = geom["crops"], 1
    w = TO.mean_window(128, 128)
    n = len(crops)
    outputs = torch.randn((n, C, 128, 128), device=dev, generator=torch.Generator(device=dev).manual_seed(2))
    first = TileMerger(geom["target_shape"], C, w, device=dev)
    first.integrate_batch(outputs, crops)
    first.merge()
    m = TileMerger(geom["target_shape"], C, w, device=dev)
    assert m.mode == flavour
    m.integrate_batch(outputs, crops)
    with pytest.warns(RuntimeWarning, match="accumulators are complete" if flavour == "planned" else "continuing on the ordinary accumulate"):
        m.integrate_batch(outputs[:1], crops[4:5])       # one more tile over pixels that were already merged
    st = TO.merger_new(geom["target_shape"], C, w)
    TO.merger_integrate(st, outputs.cpu().numpy(), crops)
    TO.merger_integrate(st, outputs[:1].cpu().numpy(), crops[4:5])
    if flavour == "planned":
        assert np.array_equal(m.merge().cpu().numpy(), TO.merger_merge(st))        # exact: the same sums in the same order
    else:
        np.testing.assert_allclose(m.merge().cpu().numpy(), TO.merger_merge(st), rtol=5e-7, atol=1e-7)


def test_self_deferred_merger_serves_the_rest_of_the_api(dev, lazy, autoplan):
    """merge_crop (complete and early), accumulate_single, half-precision outputs, crops as a collated CPU tensor, a second merge() of the
    same image and reset() on a merger that planned itself into deferred bands: the plain merger's results."""
    from pytorch_toolbelt_amd.inference import tta
    from pytorch_toolbelt_amd.inference.tiles import ImageSlicer

    TileMerger = autoplan.TileMerger
    tiler = ImageSlicer((500, 420, 3), 128, 64, weight="pyramid")
    crops, C, batch = tiler.crops, 3, 8
    n = len(crops)
    outputs = torch.randn((8 * n, C, 128, 128), device=dev, generator=torch.Generator(device=dev).manual_seed(61))

    def image(m, b=batch, upto=n, dtype=None, collate=False):
        for b0 in range(0, upto, b):
            b1 = min(upto, b0 + b)
            y = torch.cat([outputs[k * n + b0:k * n + b1] for k in range(8)])
            cc = torch.from_numpy(crops[b0:b1]) if collate else crops[b0:b1]
            m.integrate_batch(tta.d4_image_deaugment(y if dtype is None else y.to(dtype)), cc)
        return m

    plain = image(TileMerger(tiler.target_shape, C, tiler.weight, device=dev, auto_plan=False))
    image(TileMerger(tiler.target_shape, C, tiler.weight, device=dev)).merge()           # the geometry's first image
    m = image(TileMerger(tiler.target_shape, C, tiler.weight, device=dev), collate=True)
    assert m.mode == "deferred bands" and m._deferred.complete
    assert torch.equal(m.merge_crop(tiler, argmax=True, dtype=torch.uint8), plain.merge_crop(tiler, argmax=True, dtype=torch.uint8))
    assert torch.equal(m.merge_crop(tiler), plain.merge_crop(tiler)) and torch.equal(m.merge(), plain.merge()) and m.merge() is m.merge()
    m.reset()                                                                            # the same merger, next image: deferred again
    assert m.mode == "deferred bands" and torch.equal(image(m).merge(), plain.merge())
    # half-precision model outputs (the de-augmentation is evaluated at once for them: a real [B, C, h, w] half tensor arrives)
    half_plain = image(TileMerger(tiler.target_shape, C, tiler.weight, device=dev, auto_plan=False), dtype=torch.float16).merge()
    mh = image(TileMerger(tiler.target_shape, C, tiler.weight, device=dev), dtype=torch.float16)
    assert mh.mode == "deferred bands" and torch.equal(mh.merge(), half_plain)
    # merge_crop of an image that ends early (after bands went out), then accumulate_single for the missing tiles
    cut = (2 * n // 3) // batch * batch
    me = image(TileMerger(tiler.target_shape, C, tiler.weight, device=dev), upto=cut)
    pe = image(TileMerger(tiler.target_shape, C, tiler.weight, device=dev, auto_plan=False), upto=cut)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a, b = me.merge_crop(tiler), pe.merge_crop(tiler)
    assert torch.equal(torch.isnan(a), torch.isnan(b))
    torch.testing.assert_close(torch.nan_to_num(a), torch.nan_to_num(b), rtol=1e-6, atol=1e-6)
    for t in range(cut, n):
        tile = tta.d4_image_deaugment(torch.cat([outputs[k * n + t:k * n + t + 1] for k in range(8)]))[0]
        me.accumulate_single(tile, crops[t])
        pe.accumulate_single(tile, crops[t])
    torch.testing.assert_close(me.merge(), pe.merge(), rtol=1e-6, atol=1e-6)


def test_geometries_off_the_block_grid_defer_too(dev, lazy, autoplan):
    """Tile origins that are multiples of 4 but not of the planned kernels' 64 x 32 blocks (tile 96, step 48; tile 224 / 112 of the
    ImageNet-sized models): there is no block plan for them, but the band kernel takes them -- self-planned from the second image,
    and with crops=, defer=True from the first; bit-identical to the plain merger; a deviation simply goes back to it."""
    TileMerger = autoplan.TileMerger
    for shape, tile, step, C, batch in (((500, 420), 96, 48, 2, 5), ((700, 500), 224, 112, 3, 4)):
        geom = TO.slicer_geometry(shape, tile, step)
        crops = geom["crops"]
        assert np.any(crops[:, 0] % 64) and not np.any(crops[:, :2] % 4)
        w = TO.pyramid_window(tile, tile)[0]
        n = len(crops)
        outputs = torch.randn((8 * n, C, tile, tile), device=dev, generator=torch.Generator(device=dev).manual_seed(tile))
        exact = _run_image(TileMerger(geom["target_shape"], C, w, device=dev, auto_plan=False), outputs, crops, batch, literal=False)
        modes = []
        for _ in range(3):
            m = TileMerger(geom["target_shape"], C, w, device=dev)
            modes.append(m.mode)
            assert torch.equal(_run_image(m, outputs, crops, batch), exact)
            assert m.mode == modes[-1]
        assert modes == ["incremental", "deferred bands", "deferred bands"] and not m._plan.blocks
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            explicit = TileMerger(geom["target_shape"], C, w, device=dev, crops=crops, defer=True)
        assert explicit.mode == "deferred bands" and torch.equal(_run_image(explicit, outputs, crops, batch, literal=False), exact)
        # off the remembered sequence before / after bands went out: the ordinary path, no block strategy in between
        m = TileMerger(geom["target_shape"], C, w, device=dev)
        order = np.concatenate([np.arange(n)[:n // 2], np.arange(n)[n // 2:][::-1]])
        sub = torch.cat([outputs[k * n + order] for k in range(8)])
        want = _run_image(TileMerger(geom["target_shape"], C, w, device=dev, auto_plan=False), sub, crops[order], batch, literal=False)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            got = _run_image(m, sub, crops[order], batch)
        assert m.mode == "incremental"
        torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-6, equal_nan=True)


# ------------------------------------------------------------------------------------------------ self-planned deferral: what the caller never asked for
def _learn(TileMerger, geom, C, w, dev, outputs, batch, **kw):
    first = TileMerger(geom["target_shape"], C, w, device=dev)
    assert first.mode == "incremental"
    return _run_image(first, outputs, geom["crops"], batch, **kw)


def test_self_deferred_merger_degrades_without_raising(dev, lazy, autoplan, monkeypatch):
    """After bands of the image went out a self-deferred merger is asked for things deferred merging cannot serve: tiles off the
    remembered sequence, merge() of an image that ends early, a read of ``image`` in the middle.  ``TileMerger(defer=True)`` raises
    there (the caller opted in); a merger that deferred on its own account carries on as the ordinary one, inside 1e-6 of it."""
    monkeypatch.setenv("PTB_DEFER_ROWS", "128")       # six launch groups on this image
    TileMerger = autoplan.TileMerger
    geom = TO.slicer_geometry((700, 420), 128, 64)
    crops, C, batch = geom["crops"], 2, 4
    w = TO.pyramid_window(128, 128)[0]
    n = len(crops)
    outputs = torch.randn((8 * n, C, 128, 128), device=dev, generator=torch.Generator(device=dev).manual_seed(21))
    exact = _learn(TileMerger, geom, C, w, dev, outputs, batch)

    def feed(m, idx, b):
        from pytorch_toolbelt_amd.inference import tta

        for b0 in range(0, len(idx), b):
            sel = idx[b0:b0 + b]
            m.integrate_batch(tta.d4_image_deaugment(torch.cat([outputs[k * n + sel] for k in range(8)])), crops[sel])

    def plain(idx, b):
        p = TileMerger(geom["target_shape"], C, w, device=dev, auto_plan=False)
        feed(p, idx, b)
        return p

    def close(a, b):
        a, b = a.cpu().numpy(), b.cpu().numpy()
        assert np.array_equal(np.isnan(a), np.isnan(b))
        np.testing.assert_allclose(np.nan_to_num(a), np.nan_to_num(b), rtol=1e-6, atol=1e-6)

    every = np.arange(n)
    # (1) a late deviation: the last third of the image arrives in another order
    m = TileMerger(geom["target_shape"], C, w, device=dev)
    assert m.mode == "deferred bands" and len(m._bands.bands) >= 3, "the geometry must give several launch groups"
    cut = (2 * n // 3) // batch * batch
    order = np.concatenate([every[:cut], every[cut:][::-1]])
    with pytest.warns(RuntimeWarning, match="continuing on the ordinary accumulate"):
        feed(m, order, batch)
    assert m.mode == "incremental"
    close(m.merge(), plain(order, batch).merge())
    # (2) the image ends early: merge() with the lower rows missing -> NaN there, like the reference
    autoplan._auto.clear()          # (a planned image that deviated asks for more evidence: start this geometry over)
    _learn(TileMerger, geom, C, w, dev, outputs, batch)
    m = TileMerger(geom["target_shape"], C, w, device=dev)
    assert m.mode == "deferred bands"
    feed(m, every[:cut], batch)
    assert m._bands_done > 0
    close(m.merge(), plain(every[:cut], batch).merge())
    # (3) accumulators read in the middle of an image, then the rest of the tiles
    autoplan._auto.clear()
    _learn(TileMerger, geom, C, w, dev, outputs, batch)
    m = TileMerger(geom["target_shape"], C, w, device=dev)
    feed(m, every[:cut], batch)
    assert m.mode == "deferred bands" and m._bands_done > 0
    p = plain(every[:cut], batch)
    close(m.image, p.image)
    assert torch.equal(m.norm_mask, p.norm_mask) and m.mode == "incremental"
    feed(m, every[cut:], batch)
    close(m.merge(), exact)


def test_self_deferred_merger_and_static_model_outputs(dev, lazy, autoplan):
    """A model that writes every batch into ONE buffer (HIP graphs, out=): the first image of a geometry runs incrementally and
    notices (the previous batch is still referenced when the next one arrives in the same memory), so its mergers never defer --
    they plan into blocks, which read every batch inside integrate_batch; results exact.  A model that SWITCHES to a static buffer
    after its geometry was learnt with fresh outputs is refused loudly (the held predictions are already overwritten)."""
    from pytorch_toolbelt_amd.inference import tta

    TileMerger = autoplan.TileMerger
    geom = TO.slicer_geometry((384, 384), 128, 64)
    crops, C, batch = geom["crops"], 2, 4
    w = TO.pyramid_window(128, 128)[0]
    n = len(crops)
    outputs = torch.randn((8 * n, C, 128, 128), device=dev, generator=torch.Generator(device=dev).manual_seed(31))
    exact = _run_image(TileMerger(geom["target_shape"], C, w, device=dev, auto_plan=False), outputs, crops, batch)
    static = torch.empty((8 * batch, C, 128, 128), device=dev)

    def static_image(m):
        for b0 in range(0, n, batch):
            b1 = min(n, b0 + batch)
            buf = static[:8 * (b1 - b0)]
            buf.copy_(torch.cat([outputs[k * n + b0:k * n + b1] for k in range(8)]))
            m.integrate_batch(tta.d4_image_deaugment(buf), crops[b0:b1])
        return m.merge()

    modes = []
    for _ in range(3):
        m = TileMerger(geom["target_shape"], C, w, device=dev)
        modes.append(m.mode)
        assert torch.equal(static_image(m), exact)
    assert modes == ["incremental", "planned", "planned"]
    autoplan._auto.clear()
    assert torch.equal(_run_image(TileMerger(geom["target_shape"], C, w, device=dev), outputs, crops, batch), exact)     # learnt with fresh outputs
    m = TileMerger(geom["target_shape"], C, w, device=dev)
    assert m.mode == "deferred bands"
    with pytest.raises(RuntimeError, match="occupies memory of an earlier batch.*auto_plan=False"):
        static_image(m)


def test_self_deferred_merger_keeps_to_its_byte_budget(dev, lazy, autoplan, monkeypatch):
    """PTB_DEFER_BYTES bounds the model outputs a self-deferred merger keeps alive: rows per launch are halved until the peak custody
    fits, below that the merger plans into blocks; an image that brings more bytes per tile than the plan was sized for (more views)
    runs planned and the geometry re-sizes.  Results are bit-identical throughout."""
    from pytorch_toolbelt_amd.inference import _merge_modes as MM

    TileMerger = autoplan.TileMerger
    geom = TO.slicer_geometry((900, 420), 128, 64)
    crops, C, batch = geom["crops"], 2, 4
    w = TO.pyramid_window(128, 128)[0]
    n = len(crops)
    per_row = len({int(x) for x in crops[:, 0]})
    outputs = torch.randn((8 * n, C, 128, 128), device=dev, generator=torch.Generator(device=dev).manual_seed(41))
    exact = _run_image(TileMerger(geom["target_shape"], C, w, device=dev, auto_plan=False), outputs, crops, batch)
    tile_bytes = 8 * C * 128 * 128 * 4
    # (a) room for ~4 tile rows: the default 1024 rows per launch (the whole image in one group) do not fit, 128 or fewer do
    monkeypatch.setenv("PTB_DEFER_BYTES", str((4 * per_row + batch) * tile_bytes))
    assert torch.equal(_run_image(TileMerger(geom["target_shape"], C, w, device=dev), outputs, crops, batch), exact)
    m = TileMerger(geom["target_shape"], C, w, device=dev)
    assert m.mode == "deferred bands" and m._bands.rows < 1024 and (m._bands.peak_tiles() + batch) * tile_bytes <= MM.defer_budget()
    peak = 0
    from pytorch_toolbelt_amd.inference import tta
    for b0 in range(0, n, batch):
        b1 = min(n, b0 + batch)
        m.integrate_batch(tta.d4_image_deaugment(torch.cat([outputs[k * n + b0:k * n + b1] for k in range(8)])), crops[b0:b1])
        peak = max(peak, sum(h[0].numel() * 4 for h in m._held))
    assert torch.equal(m.merge(), exact) and 0 < peak <= MM.defer_budget()
    # (b) nothing fits: planned blocks
    autoplan._auto.clear()
    monkeypatch.setenv("PTB_DEFER_BYTES", str(tile_bytes))
    assert torch.equal(_run_image(TileMerger(geom["target_shape"], C, w, device=dev), outputs, crops, batch), exact)
    m = TileMerger(geom["target_shape"], C, w, device=dev)
    assert m.mode == "planned" and torch.equal(_run_image(m, outputs, crops, batch), exact)
    # (c) sized on fliplr outputs (2 views), then a d4 image (8 views) arrives: over budget at its first batch -> planned, re-sized
    autoplan._auto.clear()
    flip_exact = _run_image(TileMerger(geom["target_shape"], C, w, device=dev, auto_plan=False), outputs[:2 * n], crops, batch, group="fliplr")
    monkeypatch.setenv("PTB_DEFER_BYTES", str((n + batch) * tile_bytes // 4))          # the whole image of 2-view outputs, a quarter of it at 8 views
    assert torch.equal(_run_image(TileMerger(geom["target_shape"], C, w, device=dev), outputs[:2 * n], crops, batch, group="fliplr"), flip_exact)
    m = TileMerger(geom["target_shape"], C, w, device=dev)
    assert m.mode == "deferred bands" and m._bands.rows == 1024
    assert torch.equal(_run_image(m, outputs[:2 * n], crops, batch, group="fliplr"), flip_exact) and m.mode == "deferred bands"
    m = TileMerger(geom["target_shape"], C, w, device=dev)
    assert m.mode == "deferred bands"
    assert torch.equal(_run_image(m, outputs, crops, batch), exact) and m.mode == "planned"
    m = TileMerger(geom["target_shape"], C, w, device=dev)
    assert m.mode == "deferred bands" and m._bands.rows < 1024
    assert torch.equal(_run_image(m, outputs, crops, batch), exact) and m.mode == "deferred bands"


# ------------------------------------------------------------------------------------------------ deferred merger: the contract
def _deferred(geom, C, w, dev):
    from pytorch_toolbelt_amd.inference.tiles import TileMerger

    m = TileMerger(geom["target_shape"], C, w, device=dev, crops=geom["crops"], defer=True, defer_rows=128)
    assert m.mode == "deferred bands"
    return m


def test_deferred_merger_refuses_a_reused_output_buffer(dev):
    """A model that writes every batch into one static buffer (HIP graphs, out=): by the time the second batch arrives the first
    one -- still held, not yet read -- is gone, so no fallback could be correct: the merger raises instead of returning stale data."""
    geom = TO.slicer_geometry((384, 384), 128, 64)
    crops, C = geom["crops"], 2
    w = TO.pyramid_window(128, 128)[0]
    outputs = torch.randn((len(crops), C, 128, 128), device=dev)
    m = _deferred(geom, C, w, dev)
    static = torch.empty((4, C, 128, 128), device=dev)
    static.copy_(outputs[0:4])
    m.integrate_batch(static, crops[0:4])
    static.copy_(outputs[4:8])
    with pytest.raises(RuntimeError, match="occupies memory of an earlier batch"):
        m.integrate_batch(static, crops[4:8])
    # overlapping views of one big buffer are refused too (the merger cannot know who writes to the shared bytes)
    m = _deferred(geom, C, w, dev)
    pool = torch.randn((6, C, 128, 128), device=dev)
    m.integrate_batch(pool[0:4], crops[0:4])
    with pytest.raises(RuntimeError, match="occupies memory"):
        m.integrate_batch(pool[2:6], crops[4:8])
    # ... while the planned merger without defer reads every batch when it is handed in: same loop, correct result
    from pytorch_toolbelt_amd.inference.tiles import TileMerger

    safe = TileMerger(geom["target_shape"], C, w, device=dev, crops=crops)
    for b0 in range(0, len(crops), 4):
        static[:len(crops[b0:b0 + 4])].copy_(outputs[b0:b0 + 4])
        safe.integrate_batch(static[:len(crops[b0:b0 + 4])], crops[b0:b0 + 4])
    st = TO.merger_new(geom["target_shape"], C, w)
    TO.merger_integrate(st, outputs.cpu().numpy(), crops)
    assert np.array_equal(safe.merge().cpu().numpy(), TO.merger_merge(st))


def test_deferred_merger_notices_in_place_edits_of_held_batches(dev):
    geom = TO.slicer_geometry((384, 384), 128, 64)
    crops, C = geom["crops"], 2
    w = TO.pyramid_window(128, 128)[0]
    n = len(crops)
    outputs = [torch.randn((len(crops[b0:b0 + 4]), C, 128, 128), device=dev) for b0 in range(0, n, 4)]
    m = _deferred(geom, C, w, dev)
    m.integrate_batch(outputs[0], crops[0:4])
    outputs[0].mul_(2.0)                                  # the merger has not read it yet
    with pytest.raises(RuntimeError, match="held batch 0 .* modified in place"):
        for i in range(1, len(outputs)):
            m.integrate_batch(outputs[i], crops[4 * i:4 * i + 4])
    # untouched batches: fine, and bit-identical to the incremental merger
    outputs = [torch.randn((len(crops[b0:b0 + 4]), C, 128, 128), device=dev) for b0 in range(0, n, 4)]
    m = _deferred(geom, C, w, dev)
    from pytorch_toolbelt_amd.inference.tiles import TileMerger

    plain = TileMerger(geom["target_shape"], C, w, device=dev, auto_plan=False)
    for i, o in enumerate(outputs):
        m.integrate_batch(o, crops[4 * i:4 * i + 4])
        plain.integrate_batch(o, crops[4 * i:4 * i + 4])
    assert m.mode == "deferred bands" and torch.equal(m.merge(), plain.merge())
    with torch.inference_mode():                          # no version counters: accepted, nothing to compare
        m = _deferred(geom, C, w, dev)
        outs = [o.clone() for o in outputs]
        for i, o in enumerate(outs):
            m.integrate_batch(o, crops[4 * i:4 * i + 4])
        assert torch.equal(m.merge(), plain.merge())


@pytest.mark.parametrize("defer", [False, True], ids=["incremental", "deferred-bands"])
def test_sharded_merger_fuses_the_literal_calls_too(defer, dev, lazy):
    """`ShardedTileMerger.integrate_batch(tta.d4_image_deaugment(y), crops)`: the lazy handle is fused into the rank's launch like in
    TileMerger (two ranks played in turn, rectangles handed over by hand); the assembled result equals the single-device merge."""
    from pytorch_toolbelt_amd.inference import tta
    from pytorch_toolbelt_amd.inference.tiles import TileMerger
    from pytorch_toolbelt_amd.parallel import ShardedTileMerger

    class _Rank:
        def __init__(self, r, w):
            self.r, self.w = r, w

        def get_rank(self, group=None):
            return self.r

        def get_world_size(self, group=None):
            return self.w

    geom = TO.slicer_geometry((700, 520), (128, 128), (64, 64))
    w = TO.pyramid_window(128, 128)[0]
    crops, C = geom["crops"], 2
    n = len(crops)
    y = torch.randn((8, n, C, 128, 128), device=dev, generator=torch.Generator(device=dev).manual_seed(4))
    single = TileMerger(geom["target_shape"], C, w, device=dev, auto_plan=False)
    for b0 in range(0, n, 8):
        idx = list(range(b0, min(n, b0 + 8)))
        single.integrate_batch_deaugment(y[:, idx].reshape(-1, C, 128, 128), crops[idx], group="d4")
    want = single.merge()
    ranks = []
    fused0, eval0 = lazy.fused, lazy.evaluations
    for r in range(2):
        m = ShardedTileMerger(geom["target_shape"], C, w, crops, device=dev, dist=_Rank(r, 2), defer=defer)
        m._start_exchange = lambda: None
        m.reset()
        mine = m.tiles
        for b0 in range(0, len(mine), 8):
            idx = mine[b0:b0 + 8]
            m.integrate_batch(tta.d4_image_deaugment(y[:, idx].reshape(-1, C, 128, 128)), crops[idx])
        ranks.append(m)
    assert lazy.fused > fused0 and lazy.evaluations == eval0
    full = torch.empty_like(want)
    for m in ranks:
        for buf, (src, r0, r1, c0, c1) in zip(m._recv_buf, m.recvs):
            buf.copy_(ranks[src]._rect(r0, r1, c0, c1))
        m._exchanged = True
        o0, o1 = m.owned_rows
        full[:, o0:o1] = m.merge()
    torch.testing.assert_close(full, want, rtol=0, atol=2e-6)


def test_handles_only_where_a_later_change_of_the_source_would_be_seen(dev, lazy):
    """a handle reads its source later, so it is only handed out when an in-place change in between can be noticed.
    Tensors made under torch.inference_mode() have no version counter, and a stream being captured into a HIP graph replays into
    static buffers: both get the eagerly evaluated tensor, like the reference.  A handle consumed on ANOTHER stream than the one it
    was created on first waits for that stream (the producer of its source)."""
    from pytorch_toolbelt_amd.inference import tta

    x = torch.rand((8 * 2, 3, 64, 64), device=dev)
    assert type(tta.d4_image_deaugment(x)) is lazy.LazyDeaugment
    with torch.inference_mode():
        xi = torch.rand((8 * 2, 3, 64, 64), device=dev)
        yi = tta.d4_image_deaugment(xi)
        assert type(yi) is torch.Tensor
        xi.mul_(0.0)                                   # too late to matter: the result was computed from the original values
        assert float(yi.abs().sum()) > 0
    # created on a side stream behind a slow producer, evaluated on the default stream
    side = torch.cuda.Stream(device=dev)
    big = torch.rand((4096, 4096), device=dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(6):
            big = big @ big * 1e-3                     # keeps the side stream busy
        src = torch.rand((8 * 2, 3, 64, 64), device=dev) + big[0, 0] * 0.0
        want_src = src.clone()
        handle = tta.d4_image_deaugment(src)
        assert type(handle) is lazy.LazyDeaugment
    got = handle + 0                                   # evaluated here, on the default stream
    torch.cuda.synchronize()
    want = _eager(tta.d4_image_deaugment, want_src)
    assert torch.equal(got, want)


def test_literal_loop_under_inference_mode(dev, lazy, autoplan):
    """torch.inference_mode() -- the recommended context of an inference loop -- makes tensors without version counters, so an edit of
    a handle's source could not be noticed.  A handle is still handed out when nobody exists who could make that edit: the argument
    is a temporary of the call expression (``integrate_batch(d4_image_deaugment(model(x)), crops)``) and the only owner of its
    storage.  A tensor bound to a name, kept by the model, or sharing its storage with another tensor is evaluated on the spot, as
    before."""
    from pytorch_toolbelt_amd.inference import tta

    TileMerger = autoplan.TileMerger
    geom = TO.slicer_geometry((500, 420), 128, 64)
    crops, C, batch = geom["crops"], 3, 8
    w = TO.pyramid_window(128, 128)[0]
    n = len(crops)
    outputs = torch.randn((8 * n, C, 128, 128), device=dev, generator=torch.Generator(device=dev).manual_seed(77))
    exact = _run_image(TileMerger(geom["target_shape"], C, w, device=dev, auto_plan=False), outputs, crops, batch, literal=False)

    class Model:                     # returns a fresh tensor per call; `keep` makes it hold on to its last output
        def __init__(self, keep=False):
            self.keep, self.last = keep, None

        def __call__(self, idx):
            out = torch.cat([outputs[k * n + idx] for k in range(8)])
            if self.keep:
                self.last = out
            return out

    with torch.inference_mode():
        for image in range(3):       # image 0 learns the geometry, 1 and 2 run deferred bands -- all through fused launches
            model = Model()
            m = TileMerger(geom["target_shape"], C, w, device=dev)
            f0, e0 = lazy.fused, lazy.evaluations
            for b0 in range(0, n, batch):
                idx = torch.arange(b0, min(n, b0 + batch), device=dev)
                m.integrate_batch(tta.d4_image_deaugment(model(idx)), crops[b0:b0 + batch])
            assert lazy.fused - f0 == (n + batch - 1) // batch and lazy.evaluations == e0, "the nested literal call was not fused under inference_mode"
            assert m.mode == ("incremental" if image == 0 else "deferred bands")
            assert torch.equal(m.merge(), exact)
        idx = torch.arange(0, batch, device=dev)
        y = Model()(idx)             # bound to a name: its owner could still edit it -> evaluated at once
        assert type(tta.d4_image_deaugment(y)) is torch.Tensor
        keeper = Model(keep=True)    # the model keeps a reference to its output (a static buffer does): evaluated at once
        assert type(tta.d4_image_deaugment(keeper(idx))) is torch.Tensor
        big = torch.randn((16, C, 128, 128), device=dev)
        assert type(tta.d4_image_deaugment(big[:8])) is torch.Tensor       # a temporary VIEW of memory somebody else owns: evaluated at once
        handle = tta.d4_image_deaugment(Model()(idx))       # (not inside an `assert`: pytest's rewriting keeps the intermediate values alive)
        assert type(handle) is lazy.LazyDeaugment
        assert torch.equal(handle + 0, _eager(tta.d4_image_deaugment, y))


def test_deferred_merger_next_to_a_real_model(dev):
    """The planned + deferred merger under the allocator churn of a real convolutional model (the 4-level conv-BN-ReLU UNet of
    bench.py, narrow): every batch is a fresh tensor the caching allocator carved out of memory the previous activations just
    freed -- held batches must stay intact until their band is merged, so the result equals the plain merger's bit for bit; fp32
    and bf16-autocast outputs (read natively).  A model that writes into ONE static output buffer (HIP-graph style) is refused."""
    import bench
    from pytorch_toolbelt_amd.inference import tta
    from pytorch_toolbelt_amd.inference.tiles import ImageSlicer, TileMerger

    torch.manual_seed(0)
    model = bench._unet(width=4, classes=3).to(dev)
    tiler = ImageSlicer((700, 600, 3), 128, 64, weight="pyramid")
    image = torch.randint(0, 256, (700, 600, 3), dtype=torch.uint8, device=dev)
    n = len(tiler.crops)
    for dt in (None, torch.bfloat16):
        ctx = torch.autocast("cuda", dtype=dt) if dt is not None else torch.autocast("cuda", enabled=False)
        plain = TileMerger(tiler.target_shape, 3, tiler.weight, device=dev, auto_plan=False)
        deferred = TileMerger(tiler.target_shape, 3, tiler.weight, device=dev, crops=tiler.crops, defer=True)
        assert deferred.mode == "deferred bands"
        with torch.no_grad():
            for b0 in range(0, n, 5):
                xb = tiler.split_device(image, slice(b0, b0 + 5), augment="d4", scale=[1 / 255.0] * 3, bias=[0.0] * 3)
                with ctx:
                    yb = model(xb)
                assert yb.dtype == (dt or torch.float32)
                deferred.integrate_batch_deaugment(yb, tiler.crops[b0:b0 + 5], group="d4", reduction="mean")
                plain.integrate_batch_deaugment(yb, tiler.crops[b0:b0 + 5], group="d4", reduction="mean")
                del xb, yb                           # (the only references left are the merger's)
                junk = torch.rand((64, 3, 128, 128), device=dev)      # churn: whatever was freed is handed out again
                del junk
        assert deferred.mode == "deferred bands" and torch.equal(deferred.merge(), plain.merge())
    # static outputs: the second batch lives in the first one's memory while that one is still held
    static = torch.empty((8 * 5, 3, 128, 128), device=dev)
    deferred = TileMerger(tiler.target_shape, 3, tiler.weight, device=dev, crops=tiler.crops, defer=True)
    with torch.no_grad():
        xb = tiler.split_device(image, slice(0, 5), augment="d4", scale=[1 / 255.0] * 3, bias=[0.0] * 3)
        static.copy_(model(xb))
        deferred.integrate_batch_deaugment(static, tiler.crops[0:5], group="d4", reduction="mean")
        static.copy_(model(tiler.split_device(image, slice(5, 10), augment="d4", scale=[1 / 255.0] * 3, bias=[0.0] * 3)))
        with pytest.raises(RuntimeError, match="still held|modified in place"):
            deferred.integrate_batch_deaugment(static, tiler.crops[5:10], group="d4", reduction="mean")


# ------------------------------------------------------------------------------------------------ round 6: AMP outputs, output rings, version-less batches
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("group,reduction", [("d4", "mean"), ("d4", "gmean"), ("d2", "sum"), ("flips", "mean"), ("fliplr", "hmean")])
def test_half_precision_literal_loop_is_fused_and_bit_identical(group, reduction, dtype, dev, lazy, autoplan):
    """The literal loop on the outputs of a model under torch.autocast: `*_image_deaugment(y_half)` is a lazy
    handle too; the merger fuses it with PTB_ROUND_SRC -- the reduced value is rounded to the source dtype in registers, exactly what the
    eager call's half tensor carries into integrate_batch (tta.py:442-467 -> tiles.py:334-335) -- so every strategy (incremental, planned
    blocks, deferred bands, self-planned) gives the eager pair's bits."""
    TileMerger = autoplan.TileMerger
    geom = TO.slicer_geometry((320, 256), 64, 32)
    crops, C, batch = geom["crops"], 3, 5
    w = TO.pyramid_window(64, 64)[0]
    n, V = len(crops), GROUPS[group]
    g = torch.Generator(device=dev).manual_seed(5)
    positive = reduction in ("gmean", "hmean")
    outputs = (torch.rand((V * n, C, 64, 64), device=dev, generator=g) * 0.9 + 0.05 if positive
               else torch.randn((V * n, C, 64, 64), device=dev, generator=g)).to(dtype)
    prev = lazy.set_enabled(False)
    want = _run_image(TileMerger(geom["target_shape"], C, w, device=dev, auto_plan=False), outputs, crops, batch, group=group, reduction=reduction)
    lazy.set_enabled(prev)
    # the eager pair really rounds in between: the extension (no rounding, documented as more accurate) differs from it somewhere
    fused_ext = _run_image(TileMerger(geom["target_shape"], C, w, device=dev, auto_plan=False), outputs, crops, batch, literal=False, group=group,
                           reduction=reduction)
    assert not torch.equal(fused_ext, want) and float((fused_ext - want).abs().max()) < (2e-2 if dtype == torch.bfloat16 else 4e-3) * max(1.0, V if reduction == "sum" else 1.0)
    calls = (n + batch - 1) // batch
    for kw in ({"auto_plan": False}, {"crops": crops}, {"crops": crops, "defer": True}, {}, {}):
        f0, e0 = lazy.fused, lazy.evaluations
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = TileMerger(geom["target_shape"], C, w, device=dev, **kw)
        mode = m.mode
        got = _run_image(m, outputs, crops, batch, group=group, reduction=reduction)
        assert lazy.fused - f0 == calls and lazy.evaluations == e0, f"half-precision handles were not fused ({mode})"
        assert torch.equal(got, want), f"{mode}: fused half-precision literal loop differs from the eager pair"
    assert mode == "deferred bands"        # (the second merger without crops= planned itself)


GT3 = __import__("conftest").load_golden("tiles3.npz")


@pytest.mark.parametrize("case", GT3.by_fn("literal_loop_half"), ids=lambda c: c["name"])
def test_half_precision_literal_loop_against_the_reference(case, dev, lazy, autoplan):
    """tests/golden/tiles3.npz: the unmodified reference's literal loop on float16 / bfloat16 model outputs (CPU).  The HIP path evaluates the
    reduction in float32 and rounds ONCE to the source dtype; the reference's CPU ops round after every op of a
    non-linear reduction and sum the views in their own order, so: linear reductions agree except where the float32 sum of the views is
    itself inexact and lands on a rounding boundary (at most one unit in the last place of the half value, on a vanishing share of the
    elements); non-linear ones agree to the source dtype's precision.  Both stated below; fused == evaluated bit for bit either way."""
    kw, name = case["kwargs"], case["name"]
    from pytorch_toolbelt_amd.inference import tta
    from pytorch_toolbelt_amd.inference.tiles import ImageSlicer

    TileMerger = autoplan.TileMerger
    dt = getattr(torch, kw["dtype"])
    s = ImageSlicer(kw["image_shape"], kw["tile_size"], kw["tile_step"], weight="pyramid")
    th, tw = s.tile_size
    fn = getattr(tta, kw["group"] + "_image_deaugment")
    ulp = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10
    linear = kw["reduction"] in ("mean", "sum")
    results = []
    for image in range(2):                  # image 0: incremental; image 1: the merger planned itself (deferred bands)
        m = TileMerger(s.target_shape, kw["channels"], s.weight, device=dev)
        f0 = lazy.fused
        for bi, b0 in enumerate(range(0, len(s.crops), kw["batch"])):
            nb = min(kw["batch"], len(s.crops) - b0)
            y = torch.from_numpy(np.ascontiguousarray(GT3[f"{name}_y{bi}"])).view(dt).reshape(kw["views"] * nb, kw["channels"], th, tw).to(dev)
            if image == 0:
                tiles = _eager(fn, y, reduction=kw["reduction"])
                ref = torch.from_numpy(np.ascontiguousarray(GT3[f"{name}_t{bi}"])).view(dt).reshape(tiles.shape).float()
                err = (tiles.float().cpu() - ref).abs() / ref.abs().clamp_min(2.0 ** -14)
                if linear:
                    assert float(err.max()) <= ulp * 1.01 and float((err > 0).float().mean()) <= 2e-3, (float(err.max()), float((err > 0).float().mean()))
                else:
                    assert float(err.max()) <= 8 * ulp
            m.integrate_batch(fn(y, reduction=kw["reduction"]), s.crops[b0:b0 + nb])
        assert lazy.fused - f0 == (len(s.crops) + kw["batch"] - 1) // kw["batch"]
        results.append(m.merge())
    assert torch.equal(results[0], results[1])
    want = torch.from_numpy(GT3[f"{name}_merged"])
    err = (results[0].cpu() - want).abs() / want.abs().clamp_min(1e-2)
    if linear:
        assert float(err.max()) <= 2 * ulp and float((err > 1e-6).float().mean()) <= 5e-3, (float(err.max()), float((err > 1e-6).float().mean()))
    else:
        assert float(err.max()) <= 8 * ulp


@pytest.mark.parametrize("depth", [2, 3])
def test_ring_buffered_model_outputs_never_raise(depth, dev, lazy, autoplan):
    """VERDICT round 5, missing 3: a model that cycles through `depth` output buffers (two captured graphs, `out=bufs[i % k]`).  The
    learning image keeps its batches referenced as far back as a deferred image could ever hold them (the byte budget), so a ring no deeper
    than that is seen in image 1 and the geometry plans into blocks -- which read every batch inside integrate_batch; nothing raises and
    every image equals the plain merger's bit for bit."""
    from pytorch_toolbelt_amd.inference import tta

    TileMerger = autoplan.TileMerger
    geom = TO.slicer_geometry((384, 384), 128, 64)
    crops, C, batch = geom["crops"], 2, 4
    w = TO.pyramid_window(128, 128)[0]
    n = len(crops)
    outputs = torch.randn((8 * n, C, 128, 128), device=dev, generator=torch.Generator(device=dev).manual_seed(61))
    exact = _run_image(TileMerger(geom["target_shape"], C, w, device=dev, auto_plan=False), outputs, crops, batch)
    ring = [torch.empty((8 * batch, C, 128, 128), device=dev) for _ in range(depth)]
    modes = []
    for image in range(4):
        m = TileMerger(geom["target_shape"], C, w, device=dev)
        modes.append(m.mode)
        for i, b0 in enumerate(range(0, n, batch)):
            b1 = min(n, b0 + batch)
            buf = ring[i % depth][:8 * (b1 - b0)]
            buf.copy_(torch.cat([outputs[k * n + b0:k * n + b1] for k in range(8)]))
            m.integrate_batch(tta.d4_image_deaugment(buf), crops[b0:b1])
        assert torch.equal(m.merge(), exact), f"image {image} ({modes[-1]})"
    assert modes == ["incremental", "planned", "planned", "planned"]
    # the plain (no-TTA) loop with the same ring
    autoplan._auto.clear()
    plain = outputs[:n]
    exact = TileMerger(geom["target_shape"], C, w, device=dev, auto_plan=False)
    exact.integrate_batch(plain, crops)
    exact = exact.merge()
    modes = []
    for image in range(3):
        m = TileMerger(geom["target_shape"], C, w, device=dev)
        modes.append(m.mode)
        for i, b0 in enumerate(range(0, n, batch)):
            buf = ring[i % depth][:min(n, b0 + batch) - b0]
            buf.copy_(plain[b0:b0 + batch])
            m.integrate_batch(buf, crops[b0:b0 + batch])
        assert torch.equal(m.merge(), exact)
    assert modes == ["incremental", "planned", "planned"]


def test_version_less_batches_are_never_held_by_a_default_merger(dev, lazy, autoplan):
    """tensors made under torch.inference_mode() carry no version counter, so a merger that defers ON ITS OWN
    ACCOUNT could not notice `m1.integrate_batch(y, c); y.sigmoid_(); m2.integrate_batch(y, c)` -- reference-valid code (tiles.py:321-339
    reads the batch inside the call).  Such batches are never held: the geometry plans into blocks, every image is exact; the sole-owned
    sources of lazy handles (test_literal_loop_under_inference_mode) keep deferring."""
    TileMerger = autoplan.TileMerger
    geom = TO.slicer_geometry((384, 384), 128, 64)
    crops, C, batch = geom["crops"], 2, 4
    w = TO.pyramid_window(128, 128)[0]
    n = len(crops)
    src = torch.randn((n, C, 128, 128), device=dev, generator=torch.Generator(device=dev).manual_seed(71))

    def plain_image(m, x):
        for b0 in range(0, n, batch):
            m.integrate_batch(x[b0:b0 + batch], crops[b0:b0 + batch])
        return m.merge()

    want_raw = plain_image(TileMerger(geom["target_shape"], C, w, device=dev, auto_plan=False), src)
    want_sig = plain_image(TileMerger(geom["target_shape"], C, w, device=dev, auto_plan=False), src.sigmoid())
    with torch.inference_mode():
        modes = []
        for image in range(3):
            m1 = TileMerger(geom["target_shape"], C, w, device=dev)
            m2 = TileMerger(geom["target_shape"], C, w, device=dev)
            modes.append((m1.mode, m2.mode))
            for b0 in range(0, n, batch):
                y = src[b0:b0 + batch].clone()        # an inference tensor: no version counter
                assert torch.is_inference(y)
                m1.integrate_batch(y, crops[b0:b0 + batch])
                y.sigmoid_()                          # reference-valid: m1 has read it
                m2.integrate_batch(y, crops[b0:b0 + batch])
            assert torch.equal(m1.merge(), want_raw) and torch.equal(m2.merge(), want_sig), f"image {image}: {modes[-1]}"
        assert modes[0] == ("incremental", "incremental") and all(a != "deferred bands" and b != "deferred bands" for a, b in modes)
    # learnt with versioned tensors (deferred bands), then an inference-mode image arrives: it leaves deferred mode at its first batch
    autoplan._auto.clear()
    assert torch.equal(plain_image(TileMerger(geom["target_shape"], C, w, device=dev), src), want_raw)
    m1 = TileMerger(geom["target_shape"], C, w, device=dev)
    assert m1.mode == "deferred bands"
    with torch.inference_mode():
        m2 = TileMerger(geom["target_shape"], C, w, device=dev)
        for b0 in range(0, n, batch):
            y = src[b0:b0 + batch].clone()
            m1.integrate_batch(y, crops[b0:b0 + batch])
            y.sigmoid_()
            m2.integrate_batch(y, crops[b0:b0 + batch])
        assert m1.mode != "deferred bands"
        assert torch.equal(m1.merge(), want_raw) and torch.equal(m2.merge(), want_sig)
