"""CPU: the host-tensor path of the package (inference/_host.py, losses/_host.py -- what a CPU tensor / TileMerger(device="cpu") takes,
like the reference's device-agnostic code) pinned to the SAME golden vectors of the unmodified reference the HIP kernels are pinned to
(tests/golden/*.npz).  Values, and where the fixtures carry them the reference's autograd gradients; permutations and the tile merger
bit for bit.  No kernel runs here: `_native.calls` must not move."""
import numpy as np
import pytest
import torch

from conftest import load_golden

GA, GT, GL = load_golden("tta.npz"), load_golden("tiles.npz"), load_golden("losses.npz")
GL2, GL4, GL5, GT3, GT4, GT2 = (load_golden(n) for n in ("losses2.npz", "losses4.npz", "losses5.npz", "tta3.npz", "tta4.npz", "tta2.npz"))
TOL = dict(rtol=1e-5, atol=1e-5)
CPU = torch.device("cpu")


@pytest.fixture(autouse=True)
def no_kernel_runs():
    from pytorch_toolbelt_amd import _native as N

    before = N.calls
    yield
    assert N.calls == before, "a native entry point was called on the host path"


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _tta():
    from pytorch_toolbelt_amd.inference import tta

    return tta


def _L():
    from pytorch_toolbelt_amd import losses

    return losses


# ------------------------------------------------------------------------------------------------ TTA
@pytest.mark.parametrize("case", GA.by_fn("image_augment"), ids=lambda c: c["name"])
def test_host_augment_bit_exact(case):
    out = getattr(_tta(), f"{case['kwargs']['group']}_image_augment")(_t(GA[case["inputs"][0]]))
    assert np.array_equal(out.numpy(), GA[case["output"]])


@pytest.mark.parametrize("case", GA.by_fn("image_deaugment"), ids=lambda c: c["name"])
def test_host_deaugment(case):
    kw = case["kwargs"]
    out = getattr(_tta(), f"{kw['group']}_image_deaugment")(_t(GA[case["inputs"][0]]), reduction=kw["reduction"])
    assert type(out) is torch.Tensor, "host tensors are evaluated eagerly (no lazy handle)"
    ref = GA[case["output"]]
    assert tuple(out.shape) == ref.shape
    if kw["reduction"] is None:
        assert np.array_equal(out.numpy(), ref)
    else:
        np.testing.assert_allclose(out.numpy(), ref, rtol=1e-5, atol=1e-5, equal_nan=True)


@pytest.mark.parametrize("case", GA.by_fn("labels_deaugment"), ids=lambda c: c["name"])
def test_host_labels(case):
    kw = case["kwargs"]
    tta = _tta()
    fn = tta.fivecrop_label_deaugment if kw["group"] == "fivecrop" else getattr(tta, f"{kw['group']}_labels_deaugment")
    np.testing.assert_allclose(fn(_t(GA[case["inputs"][0]]), reduction=kw["reduction"]).numpy(), GA[case["output"]], rtol=1e-5, atol=1e-6)


def test_host_fivecrop_multiscale_reductions():
    tta = _tta()
    from pytorch_toolbelt_amd.inference import functional as F

    assert np.array_equal(tta.fivecrop_image_augment(_t(GA["x_sq"]), (8, 10)).numpy(), GA["fivecrop_aug"])
    xm = _t(GA["x_ms"])
    for c in GA.by_fn("ms_image_augment"):
        outs = tta.ms_image_augment(xm, c["kwargs"]["size_offsets"], mode="bilinear", align_corners=c["kwargs"]["align_corners"])
        assert outs[1] is xm
        for o, k in zip(outs, c["output"]):
            np.testing.assert_allclose(o.numpy(), GA[k], rtol=1e-5, atol=1e-6)
    for c in GA.by_fn("ms_image_deaugment"):
        kw = c["kwargs"]
        out = tta.ms_image_deaugment([_t(GA[k]) for k in c["inputs"]], kw["size_offsets"], reduction=kw["reduction"], mode="bilinear",
                                     align_corners=kw["align_corners"], stride=kw["stride"])
        np.testing.assert_allclose(out.numpy(), GA[c["output"]], rtol=1e-5, atol=1e-6)
    st = _t(GA["red_stack"])
    for c in GA.by_fn("reduction"):
        np.testing.assert_allclose(getattr(F, c["kwargs"]["which"])(st, dim=0).numpy(), GA[c["output"]], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("case", GT3.cases, ids=lambda c: c["name"])
def test_host_long_stacks_values_and_gradients(case):
    from pytorch_toolbelt_amd.inference import functional as F

    kw = case["kwargs"]
    x = _t(GT3[case["inputs"][0]]).requires_grad_(True)
    if case["fn"] == "deaugment_averaging":
        out, period = _tta()._deaugment_averaging(x, kw["reduction"]), 5
    else:
        out, period = getattr(F, case["fn"])(x, dim=kw["dim"], eps=kw["eps"]), 3
    np.testing.assert_allclose(out.detach().numpy(), GT3[case["output"]], rtol=1e-5, atol=1e-6)
    (out * (torch.arange(out.numel(), dtype=torch.float32).reshape(out.shape) % period + 1.0)).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), GT3[case["output"] + "_grad"], rtol=2e-4, atol=1e-5)


@pytest.mark.parametrize("case", GT2.by_fn("ms_flips_image_deaugment"), ids=lambda c: c["name"])
def test_host_multiscale_flips(case):
    """Multiscale TTA whose every scale is flip-augmented: on host tensors the composition <group>_image_deaugment + ms_image_deaugment."""
    kw = case["kwargs"]
    tta = _tta()
    offs = [tuple(o) if isinstance(o, list) else o for o in kw["size_offsets"]]
    ins = [_t(GT2[f"fz_{kw['group']}_y{i}"]) for i in range(len(offs))]
    out = tta.ms_flips_image_deaugment(ins, offs, group=kw["group"], inner_reduction=kw["inner_reduction"], reduction=kw["reduction"],
                                       align_corners=kw["align_corners"])
    np.testing.assert_allclose(out.numpy(), GT2[case["name"]], rtol=1e-5, atol=1e-5)


def test_host_views_are_differentiable_and_dtype_agnostic():
    """'TTA functions are device-agnostic and respect gradients flow' (inference/tta.py:1-5): float64 in -> float64 out, autograd through
    augment and a non-linear de-augmentation."""
    tta = _tta()
    x = torch.rand((2, 3, 6, 6), dtype=torch.float64).requires_grad_(True)
    aug = tta.d4_image_augment(x)
    assert aug.dtype == torch.float64 and aug.shape[0] == 16
    out = tta.d4_image_deaugment(aug * 0.5 + 0.25, reduction="gmean")
    assert out.dtype == torch.float64
    out.sum().backward()
    assert x.grad is not None and torch.isfinite(x.grad).all() and float(x.grad.abs().sum()) > 0
    with pytest.raises(ValueError):
        tta.d4_image_augment(torch.rand(1, 1, 4, 6))
    with pytest.raises(RuntimeError):
        tta.d4_image_deaugment(torch.rand(7, 1, 4, 4))


# ------------------------------------------------------------------------------------------------ tile merger
@pytest.mark.parametrize("case", GT.by_fn("tile_merger"), ids=lambda c: c["name"])
def test_host_tile_merger_bit_exact(case):
    from pytorch_toolbelt_amd.inference.tiles import HostBackedTileMerger, ImageSlicer, TileMerger

    kw, n = case["kwargs"], case["name"]
    s = ImageSlicer(kw["image_shape"], kw["tile_size"], kw["tile_step"], weight=kw["weight"])
    m = TileMerger(s.target_shape, kw["channels"], s.weight)           # the reference's default: device="cpu"
    assert type(m) is HostBackedTileMerger
    pred = _t(GT[f"{n}_pred"])
    for b0 in range(0, len(pred), kw["batch"]):
        m.integrate_batch(pred[b0:b0 + kw["batch"]], s.crops[b0:b0 + kw["batch"]])
    assert np.array_equal(m.image.numpy(), GT[f"{n}_image"])
    assert np.array_equal(m.norm_mask.numpy(), GT[f"{n}_norm"])
    assert np.array_equal(m.merge().numpy(), GT[f"{n}_merged"])
    # accumulate_single + merge_() (in place, aliases image) + reset()
    m.reset()
    for tile, box in zip(pred, s.crops):
        m.accumulate_single(tile, box)
    merged = m.merge_()
    assert merged is m.image and np.array_equal(merged.numpy(), GT[f"{n}_merged"])


def test_host_tile_merger_dtype_and_extensions():
    """Accumulator dtype fidelity (reference tiles.py:306-308, 334-335): a float64 host merger accumulates IN float64; the HIP merger's
    extensions (integrate_batch_deaugment, merge_crop, crops= / defer=) are accepted so that one code base runs on both."""
    from pytorch_toolbelt_amd.inference.tiles import ImageSlicer, TileMerger

    s = ImageSlicer((100, 90, 3), 32, 16, weight="pyramid")
    g = torch.Generator().manual_seed(0)
    pred = torch.rand((len(s.crops), 2, 32, 32), generator=g, dtype=torch.float64) * 255
    m64 = TileMerger(s.target_shape, 2, s.weight, dtype=torch.float64, crops=s.crops, defer=True)
    m64.integrate_batch(pred, s.crops)
    assert m64.image.dtype == torch.float64 and m64.mode == "host"
    total = np.zeros((2,) + s.target_shape)
    mass = np.zeros((1,) + s.target_shape)
    for t, (x, y, w, h) in zip(pred.numpy(), s.crops):
        total[:, y:y + h, x:x + w] += t * s.weight
        mass[:, y:y + h, x:x + w] += s.weight
    assert np.array_equal(m64.merge().numpy(), total / mass), "float64 accumulators must give the float64 sums exactly"
    # fused de-augmentation = the two reference calls
    tta = _tta()
    views = torch.rand((8 * 3, 2, 32, 32), generator=g)
    a, b = TileMerger(s.target_shape, 2, s.weight), TileMerger(s.target_shape, 2, s.weight)
    a.integrate_batch_deaugment(views, s.crops[:3], group="d4", reduction="mean")
    b.integrate_batch(tta.d4_image_deaugment(views), s.crops[:3])
    assert torch.equal(a.image, b.image)
    m = TileMerger(s.target_shape, 2, s.weight)
    m.integrate_batch(pred.float(), s.crops)
    full = m.merge()
    crop = m.merge_crop(s, layout="hwc")
    assert crop.shape == (100, 90, 2) and torch.equal(crop, full[:, s.margin_top:s.margin_top + 100, s.margin_left:s.margin_left + 90].permute(1, 2, 0))
    assert m.merge_crop(s, argmax=True, dtype=torch.uint8).dtype == torch.uint8
    with pytest.raises(ValueError):
        m.integrate_batch(pred[:2].float(), s.crops[:3])


GTD = load_golden("tiles2.npz")


@pytest.mark.parametrize("case", GTD.by_fn("tile_merger_dtype"), ids=lambda c: c["name"])
def test_host_tile_merger_accumulates_in_the_callers_dtype_bit_exact(case):
    """tests/golden/tiles2.npz: the unmodified reference's TileMerger(dtype=float16 | bfloat16 | float64) -- image / norm_mask / weight
    in that dtype, every `+=` rounded to it (tiles.py:295-308, 330-339), fed float32 and same-dtype batches, the last tile through
    accumulate_single (which does not cast, :310-319).  The torch-op merger reproduces image, norm_mask and merge() bit for bit; it is
    also what a CUDA merger with half-precision accumulators becomes under `set_reference_accumulators(True)` / `set_strict_dropin()`
    (tests/test_tiles_gpu.py checks that routing against the same op sequence on the device)."""
    from pytorch_toolbelt_amd.inference.tiles import HostBackedTileMerger, ImageSlicer, TileMerger

    kw, n = case["kwargs"], case["name"]
    dt = getattr(torch, kw["dtype"])
    s = ImageSlicer(kw["image_shape"], kw["tile_size"], kw["tile_step"], weight="pyramid")
    pred = _t(GTD[n.split("_")[0] + "_pred"])
    x = pred if kw["feed"] == "float32" else pred.to(dt)
    m = TileMerger(s.target_shape, kw["channels"], s.weight, dtype=dt)
    assert type(m) is HostBackedTileMerger and m.image.dtype == dt and m.norm_mask.dtype == dt and m.weight.dtype == dt
    last = len(s.crops) - 1
    for b0 in range(0, last, kw["batch"]):
        m.integrate_batch(x[b0:min(last, b0 + kw["batch"])], s.crops[b0:min(last, b0 + kw["batch"])])
    m.accumulate_single(pred[last].to(dt), s.crops[last])

    def raw(t):
        return t.detach().contiguous().view(torch.uint8).numpy()

    assert np.array_equal(raw(m.image), GTD[f"{n}_image"])
    assert np.array_equal(raw(m.norm_mask), GTD[f"{n}_norm"])
    merged = m.merge()
    assert merged.dtype == dt and np.array_equal(raw(merged), GTD[f"{n}_merged"])


GTILES3 = load_golden("tiles3.npz")


def _from_raw(arr, dt, shape):
    return torch.from_numpy(np.ascontiguousarray(arr)).view(dt).reshape(shape)


@pytest.mark.parametrize("case", GTILES3.by_fn("literal_loop_half"), ids=lambda c: c["name"])
def test_host_literal_loop_on_half_precision_model_outputs_bit_exact(case):
    """tests/golden/tiles3.npz: `merger.integrate_batch(tta.<group>_image_deaugment(y), crops)` of the unmodified reference with float16 /
    bfloat16 model outputs (torch.autocast): the de-augmentation returns a HALF tensor -- every op of the reduction rounded to the source
    dtype (tta.py:442-467, functional.py:250-333) -- which integrate_batch widens (tiles.py:334-335).  The torch-op path replays the
    de-augmented tiles, the accumulator and merge() bit for bit."""
    from pytorch_toolbelt_amd.inference import tta
    from pytorch_toolbelt_amd.inference.tiles import ImageSlicer, TileMerger

    kw, n = case["kwargs"], case["name"]
    dt = getattr(torch, kw["dtype"])
    s = ImageSlicer(kw["image_shape"], kw["tile_size"], kw["tile_step"], weight="pyramid")
    th, tw = s.tile_size
    m = TileMerger(s.target_shape, kw["channels"], s.weight)
    fn = getattr(tta, kw["group"] + "_image_deaugment")
    for bi, b0 in enumerate(range(0, len(s.crops), kw["batch"])):
        nb = min(kw["batch"], len(s.crops) - b0)
        y = _from_raw(GTILES3[f"{n}_y{bi}"], dt, (kw["views"] * nb, kw["channels"], th, tw))
        tiles = fn(y, reduction=kw["reduction"])
        assert tiles.dtype == dt
        assert np.array_equal(tiles.contiguous().view(torch.uint8).numpy().reshape(-1), GTILES3[f"{n}_t{bi}"].reshape(-1))
        m.integrate_batch(tiles, s.crops[b0:b0 + nb])
    assert np.array_equal(m.image.numpy(), GTILES3[f"{n}_image"])
    assert np.array_equal(m.merge().numpy(), GTILES3[f"{n}_merged"], equal_nan=True)


# ------------------------------------------------------------------------------------------------ losses
def _kw(case, G=GL):
    kw = dict(case["kwargs"])
    if kw.pop("class_weights", None):
        kw["class_weights"] = _t(G["class_weights"])
    return kw


@pytest.mark.parametrize("case", GL.by_fn("focal_loss_with_logits"), ids=lambda c: c["name"])
def test_host_focal_functional(case):
    out = _L().focal_loss_with_logits(_t(GL[case["inputs"][0]]), _t(GL[case["inputs"][1]]), **_kw(case))
    np.testing.assert_allclose(out.numpy(), GL[case["output"]], **TOL)


@pytest.mark.parametrize("case", GL.by_fn("binary_focal_loss"), ids=lambda c: c["name"])
def test_host_binary_focal_module(case):
    out = _L().BinaryFocalLoss(**_kw(case))(_t(GL[case["inputs"][0]]), _t(GL[case["inputs"][1]]))
    np.testing.assert_allclose(out.numpy(), GL[case["output"]], **TOL)


@pytest.mark.parametrize("case", GL.by_fn("softmax_focal_loss_with_logits"), ids=lambda c: c["name"])
def test_host_softmax_focal(case):
    x, t = _t(GL[case["inputs"][0]]), _t(GL[case["inputs"][1]])
    np.testing.assert_allclose(_L().softmax_focal_loss_with_logits(x, t, **_kw(case)).numpy(), GL[case["output"]], **TOL)
    np.testing.assert_allclose(_L().CrossEntropyFocalLoss(**_kw(case))(x, t).numpy(), GL[case["output"]], **TOL)


@pytest.mark.parametrize("case", GL.by_fn("soft_dice_score", "soft_jaccard_score"), ids=lambda c: c["name"])
def test_host_soft_scores(case):
    out = getattr(_L(), case["fn"])(_t(GL[case["inputs"][0]]), _t(GL[case["inputs"][1]]), **case["kwargs"])
    np.testing.assert_allclose(out.numpy(), GL[case["output"]], **TOL)


@pytest.mark.parametrize("case", GL.by_fn("dice_loss", "jaccard_loss"), ids=lambda c: c["name"])
def test_host_region_losses(case):
    cls = _L().DiceLoss if case["fn"] == "dice_loss" else _L().JaccardLoss
    out = cls(**dict(case["kwargs"]))(_t(GL[case["inputs"][0]]), _t(GL[case["inputs"][1]]))
    np.testing.assert_allclose(out.numpy(), GL[case["output"]], **TOL)


@pytest.mark.parametrize("case", GL.by_fn("lovasz_softmax", "lovasz_hinge"), ids=lambda c: c["name"])
def test_host_lovasz(case):
    cls = _L().LovaszLoss if case["fn"] == "lovasz_softmax" else _L().BinaryLovaszLoss
    out = cls(**case["kwargs"])(_t(GL[case["inputs"][0]]), _t(GL[case["inputs"][1]]))
    np.testing.assert_allclose(out.numpy(), GL[case["output"]], **TOL)


_GRAD = {"grad_binary_focal": "BinaryFocalLoss", "grad_softmax_focal": "CrossEntropyFocalLoss", "grad_dice": "DiceLoss", "grad_jaccard": "JaccardLoss"}


@pytest.mark.parametrize("case", GL.by_fn(*_GRAD), ids=lambda c: c["name"])
def test_host_gradients(case):
    x = _t(GL[case["inputs"][0]]).requires_grad_(True)
    getattr(_L(), _GRAD[case["fn"]])(**dict(case["kwargs"]))(x, _t(GL[case["inputs"][1]])).backward()
    assert np.abs(x.grad.numpy() - GL[case["output"]]).max() <= 1e-5
    np.testing.assert_allclose(x.grad.numpy(), GL[case["output"]], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("case", GL2.cases, ids=lambda c: c["name"])
def test_host_pointwise_values_and_gradients(case):
    from test_losses2_gpu import run_case

    x = _t(GL2[case["inputs"][0]]).requires_grad_(True)
    val = run_case(case["fn"], case["kwargs"], x, _t(GL2[case["inputs"][1]]), CPU)
    want = GL2[case["name"]]
    assert tuple(val.shape) == want.shape
    np.testing.assert_allclose(val.detach().numpy(), want, rtol=2e-5, atol=1e-5)
    val.sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), GL2[case["name"] + "_grad"], rtol=2e-5, atol=1e-5)


@pytest.mark.parametrize("case", GL4.cases, ids=lambda c: c["name"])
def test_host_focal_softmax_activation(case):
    L = _L()
    kw = dict(case["kwargs"])
    if kw.pop("class_weights", None):
        kw["class_weights"] = _t(GL4[case["weights"]])
    x = _t(GL4[case["inputs"][0]]).requires_grad_(True)
    t = _t(GL4[case["inputs"][1]])
    out = L.BinaryFocalLoss(activation="softmax", **kw)(x, t) if case["fn"] == "focal_softmax_module" else L.focal_loss_with_logits(x, t, activation="softmax", **kw)
    np.testing.assert_allclose(out.detach().numpy(), GL4[case["output"]], rtol=1e-5, atol=1e-5)
    w = (torch.arange(out.numel(), dtype=torch.float32).reshape(out.shape) % 7 + 1.0) if out.dim() else None
    (out * w if w is not None else out).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), GL4[case["output"] + "_grad"], rtol=2e-4, atol=2e-6)


@pytest.mark.parametrize("case", GL5.cases, ids=lambda c: c["name"])
def test_host_lovasz_larger_cases_with_gradients(case):
    from pytorch_toolbelt_amd.losses import lovasz as LV

    kw = dict(case["kwargs"])
    x = _t(GL5[case["inputs"][0]]).requires_grad_(True)
    t = _t(GL5[case["inputs"][1]])
    if case["fn"] == "lovasz_softmax":
        out = LV._lovasz_softmax(x, t, classes=kw["classes"], per_image=kw["per_image"], ignore_index=kw["ignore_index"])
    else:
        out = LV._lovasz_hinge(x, t, per_image=kw["per_image"], ignore_index=kw["ignore_index"])
    np.testing.assert_allclose(out.detach().numpy(), GL5[case["output"]], rtol=1e-5, atol=1e-6)
    out.backward()
    np.testing.assert_allclose(x.grad.numpy(), GL5[case["grad"]], rtol=1e-4, atol=1e-7)


GL6 = load_golden("losses6.npz")


@pytest.mark.parametrize("case", GL6.cases, ids=lambda c: c["name"])
def test_host_binary_focal_fractional_gamma_with_ignore_index(case):
    """losses6.npz: BinaryFocalLoss(gamma=<non-integer>, ignore_index=k) on label targets.  The value equals the reference's; the
    reference's autograd gradient is NaN on the ignored entries whose base 1 - pt is negative (functional.py:70, 90-94: `0 * NaN`) --
    NAMED DEVIATION: this library's gradient is 0 there, on the host path and in the HIP kernels alike
    (tests/test_losses2_gpu.py), and equal to the reference's everywhere else."""
    L = _L()
    x = _t(GL6[case["inputs"][0]]).requires_grad_(True)
    t = _t(GL6[case["inputs"][1]])
    out = L.BinaryFocalLoss(**case["kwargs"])(x, t)
    np.testing.assert_allclose(out.detach().numpy(), GL6[case["output"]], rtol=1e-5, atol=1e-6)
    out.backward()
    got, want = x.grad.numpy(), GL6[case["output"] + "_grad"]
    bad = np.isnan(want)
    assert int(bad.sum()) == case["nan_grads"] and np.isfinite(got).all()
    ignored = np.broadcast_to((t.numpy() == case["kwargs"]["ignore_index"])[:, None], want.shape)
    assert not (bad & ~ignored).any() and (got[ignored] == 0).all()          # the reference's NaNs sit on ignored entries only; ours are exact zeros
    np.testing.assert_allclose(got[~bad], want[~bad], rtol=2e-4, atol=1e-7)


def test_host_fused_loss_equals_its_parts():
    L = _L()
    g = torch.Generator().manual_seed(2)
    x = torch.randn((2, 5, 12, 10), generator=g)
    y = torch.randint(0, 5, (2, 12, 10), generator=g)
    fused = L.FocalDiceJaccardLoss("multiclass")(x, y)
    parts = L.BinaryFocalLoss()(x, y) + L.DiceLoss("multiclass")(x, y) + L.JaccardLoss("multiclass")(x, y)
    assert abs(float(fused) - float(parts)) < 1e-6


# ------------------------------------------------------------------------------------------------ 3-D tiles
GV = load_golden("volumes.npz")


@pytest.mark.parametrize("case", GV.by_fn("vmerger"), ids=lambda c: c["name"])
def test_host_volume_merger_bit_exact(case):
    from pytorch_toolbelt_amd.inference.tiles_3d import HostBackedVolumeMerger, VolumeMerger, VolumeSlicer

    kw, n = case["kwargs"], case["name"]
    s = VolumeSlicer(kw["volume_shape"], kw["voxel_size"], kw["voxel_step"])
    m = VolumeMerger(s.target_shape, kw["channels"], GV[f"{n}_weight"])          # the reference's default device="cpu"
    assert type(m) is HostBackedVolumeMerger and isinstance(m, VolumeMerger)
    pred = _t(GV[f"{n}_pred"])
    for b0 in range(0, len(pred), kw["batch"]):
        m.integrate_batch(pred[b0:b0 + kw["batch"]], s.crops[b0:b0 + kw["batch"]])
    assert np.array_equal(m.volume.numpy(), GV[f"{n}_volume"])
    assert np.array_equal(m.norm_mask.numpy(), GV[f"{n}_norm"])
    assert np.array_equal(m.merge().numpy(), GV[f"{n}_merged"])
    with pytest.raises(ValueError):
        m.integrate_batch(pred[:1], list(s.crops[:1]) * 2)


def test_host_view_ops_of_any_dtype_and_rank_bit_exact():
    """tests/golden/tta5.npz through the host path: the views act on dims (2, 3) of a tensor of any dtype and any rank >= 4, like the
    reference's x.flip(3) / x.rot90(k, dims=(2, 3)) / x.transpose(2, 3) (inference/functional.py:47-132)."""
    from pytorch_toolbelt_amd.inference import functional as F

    tta = _tta()
    G = load_golden("tta5.npz")

    def tensor(key, dtype_name, shape):
        dt = getattr(torch, dtype_name)
        raw = torch.from_numpy(np.ascontiguousarray(G[key]))
        return (raw.view(torch.bool) if dt == torch.bool else raw.view(dt)).reshape(shape)

    assert len(G.cases) >= 500
    for case in G.cases:
        kw = case["kwargs"]
        x = tensor(case["inputs"][0], kw["dtype"], kw["shape"])
        fn = case["fn"]
        if fn.endswith("_deaugment_none"):
            got = getattr(tta, fn[:-5])(x, reduction=None)
        elif fn.endswith("_image_augment"):
            got = getattr(tta, fn)(x)
        else:
            got = getattr(F, fn)(x)
        want = tensor(case["output"], kw["dtype"], kw["out_shape"])
        assert got.dtype == want.dtype and tuple(got.shape) == tuple(want.shape) and torch.equal(got, want), case["name"]
    with pytest.raises(IndexError):
        F.torch_fliplr(torch.zeros((3, 8, 8)))


def test_host_multiscale_area_and_nearest_exact():
    """tests/golden/tta6.npz through the host path (F.interpolate itself, whatever the mode)."""
    tta = _tta()
    G = load_golden("tta6.npz")
    for case in G.cases:
        kw = case["kwargs"]
        offs = [tuple(o) if isinstance(o, list) else o for o in kw["size_offsets"]]
        if case["fn"] == "ms_image_augment_grad":
            outs = tta.ms_image_augment(_t(G["x"]), offs, mode=kw["mode"], align_corners=None)
            for i, o in enumerate(outs):
                np.testing.assert_allclose(o.numpy(), G[f"{case['name']}_{i}"], rtol=1e-6, atol=1e-6)
        else:
            ins = [_t(G[f"fm_{i}"]).requires_grad_(True) for i in range(len(offs))]
            out = tta.ms_image_deaugment(ins, offs, reduction=kw["reduction"], mode=kw["mode"], align_corners=None)
            np.testing.assert_allclose(out.detach().numpy(), G[case["name"]], **TOL)
            (out * (torch.arange(out.numel(), dtype=torch.float32).reshape(out.shape) % 7 + 1.0)).sum().backward()
            for i, t in enumerate(ins):
                np.testing.assert_allclose(t.grad.numpy(), G[f"{case['name']}_grad_{i}"], rtol=2e-4, atol=2e-5)
