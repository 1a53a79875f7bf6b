"""GPU parity: TTA augment / de-augment / reductions (HIP) vs golden vectors of the reference and the numpy oracle.

Pure permutations (augment, reduction=None, flips) are bit-exact; reductions are held to 1e-5 (fp32)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import tta_oracle as AO

pytestmark = pytest.mark.gpu

GA = load_golden("tta.npz")


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


def _composed(fn):
    """Evaluate ``fn()`` with lazy de-augmentation off: the reference's call-by-call composition (the comparison partner of the
    fused kernels; with lazy handles on, the same calls would be fused into the one-pass kernel themselves)."""
    from pytorch_toolbelt_amd.inference import _lazy

    prev = _lazy.set_enabled(False)
    try:
        return fn()
    finally:
        _lazy.set_enabled(prev)


def _tta():
    from pytorch_toolbelt_amd.inference import tta

    return tta


@pytest.mark.parametrize("case", GA.by_fn("image_augment"), ids=lambda c: c["name"])
def test_golden_augment(case, dev):
    fn = getattr(_tta(), f"{case['kwargs']['group']}_image_augment")
    out = fn(torch.from_numpy(GA[case["inputs"][0]]).to(dev))
    assert np.array_equal(out.cpu().numpy(), GA[case["output"]])


@pytest.mark.parametrize("case", GA.by_fn("image_deaugment"), ids=lambda c: c["name"])
def test_golden_deaugment(case, dev):
    kw = case["kwargs"]
    fn = getattr(_tta(), f"{kw['group']}_image_deaugment")
    out = fn(torch.from_numpy(GA[case["inputs"][0]]).to(dev), reduction=kw["reduction"]).cpu().numpy()
    ref = GA[case["output"]]
    assert out.shape == ref.shape
    if kw["reduction"] is None:
        assert np.array_equal(out, ref)
    else:
        np.testing.assert_allclose(out, ref, rtol=1e-5, atol=1e-5, equal_nan=True)


@pytest.mark.parametrize("case", GA.by_fn("labels_deaugment"), ids=lambda c: c["name"])
def test_golden_labels(case, dev):
    kw = case["kwargs"]
    tta = _tta()
    fn = tta.fivecrop_label_deaugment if kw["group"] == "fivecrop" else getattr(tta, f"{kw['group']}_labels_deaugment")
    out = fn(torch.from_numpy(GA[case["inputs"][0]]).to(dev), reduction=kw["reduction"]).cpu().numpy()
    np.testing.assert_allclose(out, GA[case["output"]], rtol=1e-5, atol=1e-6)


def test_golden_fivecrop_ms_reductions(dev):
    tta = _tta()
    from pytorch_toolbelt_amd.inference import functional as F

    x = torch.from_numpy(GA["x_sq"]).to(dev)
    assert np.array_equal(tta.fivecrop_image_augment(x, (8, 10)).cpu().numpy(), GA["fivecrop_aug"])
    xm = torch.from_numpy(GA["x_ms"]).to(dev)
    for c in GA.by_fn("ms_image_augment"):
        outs = tta.ms_image_augment(xm, c["kwargs"]["size_offsets"], mode="bilinear", align_corners=c["kwargs"]["align_corners"])
        assert outs[1] is xm  # offset 0 returns the input itself
        for o, k in zip(outs, c["output"]):
            np.testing.assert_allclose(o.cpu().numpy(), GA[k], rtol=1e-5, atol=1e-6)
    for c in GA.by_fn("ms_image_deaugment"):
        kw = c["kwargs"]
        fmaps = [torch.from_numpy(GA[k]).to(dev) for k in c["inputs"]]
        out = tta.ms_image_deaugment(fmaps, kw["size_offsets"], reduction=kw["reduction"], mode="bilinear", align_corners=kw["align_corners"], stride=kw["stride"])
        np.testing.assert_allclose(out.cpu().numpy(), GA[c["output"]], rtol=1e-5, atol=1e-6)
    st = torch.from_numpy(GA["red_stack"]).to(dev)
    for c in GA.by_fn("reduction"):
        out = getattr(F, c["kwargs"]["which"])(st, dim=0)
        np.testing.assert_allclose(out.cpu().numpy(), GA[c["output"]], rtol=1e-5, atol=1e-6)
    # reduction along another dim == oracle on the moved axis
    out = F.geometric_mean(st, dim=1).cpu().numpy()
    np.testing.assert_allclose(out, AO.geometric_mean(np.moveaxis(GA["red_stack"], 1, 0)), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("chunk_rows,scalar", [(64, 0), (32, 0), (16, 0), (64, 1)])
@pytest.mark.parametrize("group,shape", [("d4", (3, 2, 200, 200)), ("d4", (2, 3, 75, 75)), ("d2", (2, 2, 136, 68)), ("flips", (2, 2, 37, 53)), ("fliplr", (1, 5, 64, 260)), ("flipud", (3, 1, 130, 4))])
def test_roundtrip_and_oracle(group, shape, chunk_rows, scalar, dev, native):
    """reference tests/test_tta.py:31-68 (deaugment(augment(x)) == x) on ragged shapes + oracle comparison."""
    lib = native.load()
    lib.ptb_set_tunable(0, chunk_rows)
    lib.ptb_set_tunable(1, scalar)
    tta = _tta()
    rng = np.random.default_rng(7)
    x = rng.random(shape, dtype=np.float32)
    tx = torch.from_numpy(x).to(dev)
    aug = getattr(tta, f"{group}_image_augment")(tx)
    assert np.array_equal(aug.cpu().numpy(), AO.image_augment(x, group))
    back = getattr(tta, f"{group}_image_deaugment")(aug)
    np.testing.assert_allclose(back.cpu().numpy(), x, atol=1e-6, rtol=1e-6)
    V = aug.shape[0] // shape[0]
    y = (rng.random((V * shape[0],) + shape[1:], dtype=np.float32) * 0.9 + 0.05)
    ty = torch.from_numpy(y).to(dev)
    for red in ("mean", "sum", "gmean", "logodd", None):
        out = getattr(tta, f"{group}_image_deaugment")(ty, reduction=red).cpu().numpy()
        np.testing.assert_allclose(out, AO.image_deaugment(y, group, red), rtol=1e-5, atol=1e-5)
    # callable reduction receives the [V, B, ...] stack
    out = getattr(tta, f"{group}_image_deaugment")(ty, reduction=lambda t, dim: t.max(dim=dim)[0]).cpu().numpy()
    assert np.array_equal(out, AO.image_deaugment(y, group, None).max(axis=0))


def test_reference_kat_labels_and_wrappers(dev):
    """reference tests/test_tta.py:71-108 with the SumAll model, plus d4/fliplr image2mask with an identity model."""
    tta = _tta()
    x = torch.tensor([[1, 2, 3, 4], [5, 6, 7, 8], [9, 0, 1, 2], [3, 4, 5, 6]], device=dev).unsqueeze(0).unsqueeze(0).float()
    model = lambda t: t.sum(dim=[1, 2, 3])
    assert int(tta.d4_image2label(model, x)) == int(x.sum())
    assert int(tta.fliplr_image2label(model, x)) == int(x.sum())
    assert int(tta.fivecrop_image2label(model, x, (2, 2))) == ((1 + 2 + 5 + 6) + (3 + 4 + 7 + 8) + (9 + 0 + 3 + 4) + (1 + 2 + 5 + 6) + (6 + 7 + 0 + 1)) / 5
    assert int(tta.tencrop_image2label(model, x, (2, 2))) == (2 * ((1 + 2 + 5 + 6) + (3 + 4 + 7 + 8) + (9 + 0 + 3 + 4) + (1 + 2 + 5 + 6) + (6 + 7 + 0 + 1))) / 10
    img = torch.rand((4, 3, 224, 224), device=dev)
    ident = lambda t: t
    np.testing.assert_allclose(tta.d4_image2mask(ident, img).cpu().numpy(), img.cpu().numpy(), atol=1e-6, rtol=1e-6)
    np.testing.assert_allclose(tta.fliplr_image2mask(ident, img).cpu().numpy(), img.cpu().numpy(), atol=1e-6, rtol=1e-6)
    # d4_labels_deaugment reference quirk: chunk 6 dropped, chunk 7 twice -> mean of 0..7 is 3.625
    v = torch.arange(8, device=dev, dtype=torch.float32).view(8, 1)
    assert float(tta.d4_labels_deaugment(v)) == pytest.approx(3.625)


def test_errors(dev):
    tta = _tta()
    with pytest.raises(ValueError):
        tta.d4_image_augment(torch.rand((1, 1, 8, 12), device=dev))
    with pytest.raises(RuntimeError):
        tta.d4_image_deaugment(torch.rand((7, 1, 8, 8), device=dev))
    with pytest.raises(KeyError):
        tta.d4_image_deaugment(torch.rand((8, 1, 8, 8), device=dev), reduction="median")
    with pytest.raises(RuntimeError):
        tta.flips_labels_deaugment(torch.rand((4, 3), device=dev))
    with pytest.raises(ValueError):
        tta.ms_image_deaugment([torch.rand((1, 1, 8, 8), device=dev)], [0, 2])
    with pytest.raises(ValueError):
        tta.fivecrop_image_augment(torch.rand((1, 1, 8, 8), device=dev), (9, 4))


def test_functional_view_ops(dev):
    from pytorch_toolbelt_amd.inference import functional as F

    x = np.random.default_rng(9).random((2, 3, 20, 20), dtype=np.float32)
    t = torch.from_numpy(x).to(dev)
    table = {
        "torch_fliplr": x[..., :, ::-1], "torch_flipud": x[..., ::-1, :], "torch_rot180": x[..., ::-1, ::-1],
        "torch_transpose": np.swapaxes(x, 2, 3), "torch_transpose2": np.swapaxes(x, 2, 3),
        "torch_rot90_ccw": np.rot90(x, 1, axes=(2, 3)), "torch_rot90_cw": np.rot90(x, -1, axes=(2, 3)),
        "torch_rot90_ccw_transpose": np.swapaxes(np.rot90(x, 1, axes=(2, 3)), 2, 3),
        "torch_rot90_cw_transpose": np.swapaxes(np.rot90(x, -1, axes=(2, 3)), 2, 3),
        "torch_rot180_transpose": np.swapaxes(np.rot90(x, 2, axes=(2, 3)), 2, 3),
        "torch_transpose_rot90_ccw": np.rot90(np.swapaxes(x, 2, 3), 1, axes=(2, 3)),
        "torch_transpose_rot90_cw": np.rot90(np.swapaxes(x, 2, 3), -1, axes=(2, 3)),
        "torch_transpose_rot180": np.rot90(np.swapaxes(x, 2, 3), 2, axes=(2, 3)),
    }
    for name, want in table.items():
        assert np.array_equal(getattr(F, name)(t).cpu().numpy(), want), name
    assert F.torch_none(t) is t
    with pytest.warns(DeprecationWarning):
        assert np.array_equal(F.torch_rot90(t).cpu().numpy(), table["torch_rot90_ccw"])
    # non-square flips
    r = torch.rand((1, 2, 12, 40), device=dev)
    assert torch.equal(F.torch_fliplr(r), r.flip(3)) and torch.equal(F.torch_flipud(r), r.flip(2))
    # pad helpers are exact shape arithmetic (reference tests/test_utils_functional.py)
    p, pad = F.pad_image_tensor(r, 32)
    assert p.shape == (1, 2, 32, 64) and torch.equal(F.unpad_image_tensor(p, pad), r)
    p2, crop = F.pad_tensor_to_size(r, (16, 41))
    assert p2.shape == (1, 2, 16, 41) and torch.equal(p2[crop], r)


def test_autograd_adjoint(dev):
    """TTA functions respect gradient flow (reference tta.py:3-4): check <L x, g> == <x, L^T g> for the linear maps."""
    tta = _tta()
    for group in ("fliplr", "flips", "d2", "d4"):
        x = torch.randn((2, 3, 32, 32), device=dev, requires_grad=True)
        aug = getattr(tta, f"{group}_image_augment")(x)
        g = torch.randn_like(aug)
        aug.backward(g)
        # adjoint of augment = de-augment with sum
        want = getattr(tta, f"{group}_image_deaugment")(g, reduction="sum")
        torch.testing.assert_close(x.grad, want, rtol=1e-6, atol=1e-6)
        y = torch.randn_like(aug, requires_grad=True)
        out = getattr(tta, f"{group}_image_deaugment")(y)  # mean
        go = torch.randn_like(out)
        out.backward(go)
        V = aug.shape[0] // 2
        want = getattr(tta, f"{group}_image_augment")(go) / V
        torch.testing.assert_close(y.grad, want, rtol=1e-6, atol=1e-6)
    lg = torch.randn((16, 5), device=dev, requires_grad=True)
    tta.d2_labels_deaugment(lg).sum().backward()
    torch.testing.assert_close(lg.grad, torch.full_like(lg, 0.25))


def test_generalized_and_multiscale_wrappers(dev):
    tta = _tta()
    x = torch.rand((2, 3, 64, 64), device=dev)
    m = tta.GeneralizedTTA(torch.nn.Identity(), tta.d4_image_augment, tta.d4_image_deaugment)
    np.testing.assert_allclose(m(x).cpu().numpy(), x.cpu().numpy(), atol=1e-6)

    class TwoHeads(torch.nn.Module):
        def forward(self, image):
            return {"mask": image * 2, "edge": image + 1}

    m = tta.GeneralizedTTA(TwoHeads(), {"image": tta.d2_image_augment}, {"mask": tta.d2_image_deaugment, "edge": tta.d2_image_deaugment})
    out = m(image=x)
    np.testing.assert_allclose(out["mask"].cpu().numpy(), (x * 2).cpu().numpy(), atol=1e-6)
    with pytest.raises(ValueError):
        m(x)
    ms = tta.MultiscaleTTA(torch.nn.Identity(), size_offsets=[-16, 0, 16])
    y = ms(x)
    assert y.shape == x.shape
    want = AO.ms_image_deaugment(AO.ms_image_augment(x.cpu().numpy(), [-16, 0, 16], False), [-16, 0, 16], "mean", True)  # quirk Q2
    np.testing.assert_allclose(y.cpu().numpy(), want, rtol=1e-5, atol=1e-5)


def test_cfg5_multiscale_fliplr_gmean(dev):
    """BASELINE configs[4] pattern: scales 0.75/1.0/1.25 (pixel offsets -N/4, 0, +N/4), each wrapped in fliplr TTA,
    gmean merge -- reduced to 384x384 for the oracle, plus size-independent properties at 4096x4096."""
    tta = _tta()
    N_ = 384
    offs = [-N_ // 4, 0, N_ // 4]
    rng = np.random.default_rng(12)
    outs = [(rng.random((2 * 1, 3, N_ + o, N_ + o), dtype=np.float32) * 0.9 + 0.05) for o in offs]   # fliplr-augmented model outputs
    per_scale = [tta.fliplr_image_deaugment(torch.from_numpy(o).to(dev)) for o in outs]
    got = tta.ms_image_deaugment(per_scale, offs, reduction="gmean", mode="bilinear", align_corners=False)
    want = AO.ms_image_deaugment([AO.image_deaugment(o, "fliplr", "mean") for o in outs], offs, "gmean", False)
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-5, atol=1e-5)
    for red in ("mean", "sum", "hmean", "logodd"):
        got = tta.ms_image_deaugment(per_scale, offs, reduction=red, align_corners=True)
        want = AO.ms_image_deaugment([AO.image_deaugment(o, "fliplr", "mean") for o in outs], offs, red, True)
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-5, atol=1e-5)
    # full size: a constant map at every scale merges to that constant (bilinear and gmean both reproduce constants);
    # a horizontal ramp stays a ramp under align_corners=True resampling
    big = [torch.full((1, 4, 4096 + o, 4096 + o), 0.37, device=dev) for o in (-1024, 0, 1024)]
    out = tta.ms_image_deaugment(big, [-1024, 0, 1024], reduction="gmean", align_corners=False)
    assert out.shape == (1, 4, 4096, 4096) and float((out - 0.37).abs().max()) <= 1e-6
    ramps = [torch.linspace(0, 1, 4096 + o, device=dev).view(1, 1, 1, -1).expand(1, 4, 4096 + o, 4096 + o).contiguous() for o in (-1024, 0, 1024)]
    out = tta.ms_image_deaugment(ramps, [-1024, 0, 1024], reduction="mean", align_corners=True)
    ref = torch.linspace(0, 1, 4096, device=dev).view(1, 1, 1, -1).expand_as(out)
    assert float((out - ref).abs().max()) <= 2e-6


def test_half_precision_inputs(dev):
    """fp16 / bf16 tensors (AMP inference) are accepted: evaluated in fp32, returned in the input dtype."""
    tta = _tta()
    x = torch.rand((2, 3, 64, 64), device=dev)
    for dt, tol in ((torch.float16, 2e-3), (torch.bfloat16, 1.6e-2)):
        xa = tta.d4_image_augment(x.to(dt))
        assert xa.dtype == dt and torch.equal(xa, tta.d4_image_augment(x).to(dt))
        y = torch.rand((16, 3, 64, 64), device=dev).to(dt)
        out = tta.d4_image_deaugment(y)
        assert out.dtype == dt
        want = AO.image_deaugment(y.float().cpu().numpy(), "d4", "mean")
        np.testing.assert_allclose(out.float().cpu().numpy(), want, atol=tol, rtol=tol)
        lab = tta.d2_labels_deaugment(torch.rand((8, 5), device=dev).to(dt))
        assert lab.dtype == dt
    # integer and float64 images: the views are permutations -- any dtype, exactly its bits (test_view_ops_take_every_dtype_and_rank)
    for dt in (torch.uint8, torch.int16, torch.int32, torch.float64):
        xi = (torch.rand((2, 3, 16, 16), device=dev) * 200).to(dt)
        got = tta.d4_image_augment(xi)
        assert got.dtype == dt and torch.equal(got.float(), tta.d4_image_augment(xi.float()))


_VIEW_OPS = {   # the reference's chains, inference/functional.py:47-132
    "torch_none": lambda x: x,
    "torch_fliplr": lambda x: x.flip(3),
    "torch_flipud": lambda x: x.flip(2),
    "torch_rot90_ccw": lambda x: x.rot90(k=1, dims=(2, 3)),
    "torch_rot90_cw": lambda x: x.rot90(k=-1, dims=(2, 3)),
    "torch_rot180": lambda x: torch.rot90(x, k=2, dims=(2, 3)),
    "torch_transpose": lambda x: x.transpose(2, 3),
    "torch_transpose2": lambda x: x.transpose(3, 2),
    "torch_rot90_ccw_transpose": lambda x: x.rot90(k=1, dims=(2, 3)).transpose(2, 3),
    "torch_rot90_cw_transpose": lambda x: x.rot90(k=-1, dims=(2, 3)).transpose(2, 3),
    "torch_rot180_transpose": lambda x: x.rot90(k=2, dims=(2, 3)).transpose(2, 3),
    "torch_transpose_rot90_ccw": lambda x: x.transpose(2, 3).rot90(k=1, dims=(2, 3)),
    "torch_transpose_rot90_cw": lambda x: x.transpose(2, 3).rot90(k=-1, dims=(2, 3)),
    "torch_transpose_rot180": lambda x: x.transpose(2, 3).rot90(k=2, dims=(2, 3)),
}
_ALL_DTYPES = [torch.uint8, torch.int8, torch.int16, torch.int32, torch.int64, torch.bool, torch.float16, torch.bfloat16, torch.float32,
               torch.float64, torch.complex64, torch.complex128]


def _any_dtype(shape, dt, dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    if dt == torch.bool:
        return (torch.rand(shape, generator=g) < 0.5).to(dev)
    if dt.is_complex:
        return torch.complex(torch.randn(shape, generator=g, dtype=torch.float64), torch.randn(shape, generator=g, dtype=torch.float64)).to(dt).to(dev)
    if dt.is_floating_point:
        return torch.randn(shape, generator=g, dtype=torch.float64).to(dt).to(dev)
    info = torch.iinfo(dt)
    return torch.randint(max(info.min, -2**62), min(info.max, 2**62), shape, generator=g, dtype=torch.int64).to(dt).to(dev)


@pytest.mark.parametrize("dt", _ALL_DTYPES, ids=lambda d: str(d).replace("torch.", ""))
def test_view_ops_take_every_dtype_and_rank(dt, dev):
    """torch_fliplr ... torch_transpose_rot180, *_image_augment and *_image_deaugment(reduction=None) are index permutations of dims 2
    and 3 (inference/functional.py:47-132: x.flip(3), x.rot90(k, dims=(2, 3)), x.transpose(2, 3)): every dtype, every rank >= 4
    (dims beyond the fourth ride along), non-square planes -- torch.equal with the reference's own torch ops on the same CUDA tensor."""
    from pytorch_toolbelt_amd.inference import functional as F

    tta = _tta()
    for shape in ((2, 3, 24, 24), (1, 2, 70, 37), (2, 1, 9, 130), (1, 2, 12, 20, 3), (1, 1, 6, 6, 2, 5)):
        x = _any_dtype(shape, dt, dev, seed=len(shape) + shape[2])
        for name, ref in _VIEW_OPS.items():
            got = getattr(F, name)(x)
            want = ref(x)
            assert got.dtype == dt and got.shape == want.shape and torch.equal(got, want), (name, shape)
            assert name == "torch_none" or got.is_contiguous()
    # the augment / de-augment groups on square planes (d4 needs them square: tta.py:399-406), 4-D and 5-D
    def chunks(x):      # the reference's torch.cat lists, inference/tta.py:257-284, 319-341, 385-422, 470-484
        xt = x.transpose(2, 3)
        return {"fliplr": [x, x.flip(3)], "flipud": [x, x.flip(2)], "flips": [x, x.flip(3), x.flip(2)],
                "d2": [x, x.flip(3), x.flip(2), x.flip(2).flip(3)],
                "d4": [x, x.rot90(-1, (2, 3)), x.rot90(2, (2, 3)), x.rot90(1, (2, 3)), xt, xt.rot90(-1, (2, 3)), xt.rot90(2, (2, 3)), xt.rot90(1, (2, 3))]}

    for shape in ((2, 2, 16, 16), (1, 2, 10, 10, 3)):
        x = _any_dtype(shape, dt, dev, seed=7)
        for group, want in chunks(x).items():
            aug = getattr(tta, f"{group}_image_augment")(x)
            V = len(want)
            assert aug.dtype == dt and aug.shape[0] == V * shape[0]
            assert torch.equal(aug, torch.cat(want)), (group, shape)
            # de-augment without a reduction undoes every view: V copies of x, bit for bit
            back = getattr(tta, f"{group}_image_deaugment")(aug, reduction=None)
            assert back.dtype == dt and tuple(back.shape) == (V,) + tuple(shape) and all(torch.equal(back[k], x) for k in range(V)), (group, shape)
    with pytest.raises(ValueError, match="rows equal to number of cols"):
        tta.d4_image_augment(_any_dtype((1, 1, 8, 12), dt, dev))
    for bad in (torch.zeros((3, 8, 8), device=dev).to(dt), torch.zeros((8,), device=dev).to(dt)):      # dims 2 / 3 do not exist: torch's own error
        with pytest.raises(IndexError):
            F.torch_fliplr(bad)


def golden_tensor(G, key, dtype_name, shape):
    """An array of tests/golden/tta5.npz (raw bytes of the reference's tensor) as a torch tensor of its dtype and shape."""
    dt = getattr(torch, dtype_name)
    raw = torch.from_numpy(np.ascontiguousarray(G[key]))
    return (raw.view(torch.bool) if dt == torch.bool else raw.view(dt)).reshape(shape)


def test_view_ops_of_any_dtype_against_the_reference_goldens(dev):
    """tests/golden/tta5.npz: outputs of the UNMODIFIED reference's view ops / augment / de-augment(reduction=None) on integer, boolean,
    half, float64 and complex tensors, 4-D and 5-D, square and not (oracle/make_golden.py gen_tta5) -- replayed through the HIP
    permutation kernel: same dtype, same shape, same bytes."""
    from pytorch_toolbelt_amd.inference import functional as F

    tta = _tta()
    G = load_golden("tta5.npz")
    assert len(G.cases) >= 500
    for case in G.cases:
        kw = case["kwargs"]
        src_shape = kw["shape"]
        x = golden_tensor(G, case["inputs"][0], kw["dtype"], src_shape).to(dev)
        fn = case["fn"]
        if fn.endswith("_deaugment_none"):
            got = getattr(tta, fn[:-5])(x, reduction=None)
        elif fn.endswith("_image_augment"):
            got = getattr(tta, fn)(x)
        else:
            got = getattr(F, fn)(x)
        want = golden_tensor(G, case["output"], kw["dtype"], kw["out_shape"])
        assert got.dtype == want.dtype and tuple(got.shape) == tuple(want.shape), case["name"]
        assert torch.equal(got.cpu().contiguous().view(torch.uint8) if got.dtype != torch.bool else got.cpu(), want.contiguous().view(torch.uint8) if want.dtype != torch.bool else want), case["name"]


def test_float64_and_nd_tta_reductions_on_the_device(dev):
    """float64 model outputs are de-augmented and reduced IN float64 (the reference is dtype-agnostic; round 4 computed them in
    float32), 5-D float32 outputs go through the permutation + the HIP stack reduction; gradients flow through both."""
    tta = _tta()
    y = torch.rand((16, 2, 12, 12), device=dev, dtype=torch.float64) * 0.9 + 0.05
    for red in ("mean", "sum", "gmean", "hmean", "logodd"):
        got = tta.d4_image_deaugment(y, reduction=red)
        want = AO.image_deaugment(y.cpu().numpy(), "d4", red)
        assert got.dtype == torch.float64 and want.dtype == np.float64
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-12, atol=1e-13)
    y5 = torch.rand((8, 2, 10, 10, 3), device=dev) * 0.9 + 0.05
    got = tta.d4_image_deaugment(y5, reduction="gmean")
    want = np.stack([AO.image_deaugment(y5[..., k].cpu().numpy(), "d4", "gmean") for k in range(3)], axis=-1)
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-5, atol=1e-6)
    for t in (y.clone().requires_grad_(True), y5.clone().requires_grad_(True)):
        out = tta.d4_image_deaugment(t, reduction="mean")
        (out * out).sum().backward()
        ref = t.detach().clone().requires_grad_(True)
        from pytorch_toolbelt_amd.inference import _host

        o2 = _host.deaug_reduce(ref, list(tta.DEAUGMENT_VIEWS["d4"]), 1)
        (o2 * o2).sum().backward()
        torch.testing.assert_close(t.grad, ref.grad, rtol=1e-5, atol=1e-6)
        a = tta.d4_image_augment(t[:1])
        assert a.requires_grad


@pytest.mark.parametrize("reduction", ["gmean", "hmean", "harmonic1p", "logodd", "log1p"])
@pytest.mark.parametrize("group,shape", [("d4", (16, 2, 48, 48)), ("d2", (8, 3, 20, 36)), ("flips", (6, 1, 15, 22))])
def test_nonlinear_reduction_gradients(reduction, group, shape, dev):
    """TTA "respects gradient flow" (reference tta.py:3-4) also through the non-linear reductions: compare the HIP
    backward with torch autograd of the reference formula (inference/functional.py:250-333) on the de-augmented stack."""
    tta = _tta()
    g = torch.Generator().manual_seed(3)
    y = (torch.rand(shape, generator=g) * 0.9 + 0.05).to(dev)
    y1 = y.clone().requires_grad_(True)
    out = getattr(tta, f"{group}_image_deaugment")(y1, reduction=reduction)
    y2 = y.clone().requires_grad_(True)
    stack = getattr(tta, f"{group}_image_deaugment")(y2, reduction=None)   # [V, B, ...], differentiable permutation
    ref = {
        "gmean": lambda t: t.log().mean(0).exp(),
        "hmean": lambda t: torch.reciprocal(torch.reciprocal(t.clamp_min(1e-6)).mean(0).clamp_min(1e-6)),
        "harmonic1p": lambda t: torch.reciprocal(torch.reciprocal(t + 1).mean(0)) - 1,
        "logodd": lambda t: torch.sigmoid(torch.log(t.clamp(1e-6, 1 - 1e-6) / (1 - t.clamp(1e-6, 1 - 1e-6))).mean(0)),
        "log1p": lambda t: torch.log1p(t).mean(0).exp() - 1,
    }[reduction](stack)
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-6)
    w = torch.rand_like(out)
    (out * w).sum().backward()
    (ref * w).sum().backward()
    torch.testing.assert_close(y1.grad, y2.grad, rtol=1e-4, atol=1e-6)
    # generic [T, B, ...] stacks (labels, ensembling) go through the same kernel
    lg = (torch.rand((4 * 3, 7), generator=g) * 0.9 + 0.05).to(dev)
    l1 = lg.clone().requires_grad_(True)
    tta.d2_labels_deaugment(l1, reduction=reduction).sum().backward()
    l2 = lg.clone().requires_grad_(True)
    st = l2.view(4, 3, 7)
    {"gmean": lambda t: t.log().mean(0).exp(), "hmean": lambda t: 1 / (1 / t.clamp_min(1e-6)).mean(0).clamp_min(1e-6),
     "harmonic1p": lambda t: 1 / (1 / (t + 1)).mean(0) - 1,
     "logodd": lambda t: torch.sigmoid(torch.log(t / (1 - t)).mean(0)), "log1p": lambda t: torch.log1p(t).mean(0).exp() - 1}[reduction](st).sum().backward()
    torch.testing.assert_close(l1.grad, l2.grad, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("world", [2, 4, 7])
@pytest.mark.parametrize("align_corners", [True, False])
@pytest.mark.parametrize("reduction", ["mean", "gmean"])
def test_multiscale_row_strips_equal_the_full_result(world, align_corners, reduction, dev):
    """cfg5 over several GPUs: every rank reduces its own strip of output rows from source row strips with
    the halo `ms_strip_plan` prescribes -- no collective; concatenated, the strips equal the single-GPU call bit for bit."""
    from pytorch_toolbelt_amd.inference import tta
    from pytorch_toolbelt_amd.parallel import ms_image_deaugment_strip, ms_strip_plan

    torch.manual_seed(world)
    H, W = 96, 128
    offsets = [-24, 0, 32, 13]
    maps = [torch.rand((2, 3, H + o, W + o), device=dev) * 0.9 + 0.05 for o in offsets]
    full = tta.ms_image_deaugment(maps, offsets, reduction=reduction, align_corners=align_corners)
    heights = [m.shape[2] for m in maps]
    plan = ms_strip_plan(heights, H, world, align_corners)
    assert plan[0]["out"][0] == 0 and plan[-1]["out"][1] == H
    pieces = []
    for entry in plan:
        strips = [m[:, :, s0:s1].contiguous() for m, (s0, s1) in zip(maps, entry["src"])]
        for (s0, s1), h in zip(entry["src"], heights):
            assert 0 <= s0 < s1 <= h and s1 - s0 < h or world == 1 or h <= 40
        pieces.append(ms_image_deaugment_strip(strips, heights, entry["src"], entry["out"], (H, W), reduction, align_corners))
    assert torch.equal(torch.cat(pieces, dim=2), full)


@pytest.mark.parametrize("world", [2, 4, 7])
@pytest.mark.parametrize("align_corners", [True, False])
@pytest.mark.parametrize("inner,outer", [("mean", "mean"), ("gmean", "gmean")])
def test_flips_multiscale_row_strips_equal_the_full_result(world, align_corners, inner, outer, dev):
    """BASELINE configs[4] over several GPUs with the ONE-PASS kernel: every rank reduces its strip of output rows from the row
    strips of every view of every scale (`ptb_ms_flip_deaug_reduce_strip`); concatenated, the strips equal the single-GPU one-pass call
    bit for bit (ragged sizes, four scales incl. the same-size one); a strip that lacks a row its taps read is refused, and so are
    groups whose views flip rows."""
    from pytorch_toolbelt_amd.inference import tta
    from pytorch_toolbelt_amd.parallel import ms_flips_image_deaugment_strip, ms_strip_plan

    torch.manual_seed(world)
    H, W = 200, 264
    offsets = [-48, 0, 52, 24]
    ys = [torch.rand((2 * 2, 3, H + o, W + o), device=dev) * 0.9 + 0.05 for o in offsets]       # fliplr: 2 views x batch 2
    full = tta.ms_flips_image_deaugment(ys, offsets, group="fliplr", inner_reduction=inner, reduction=outer, align_corners=align_corners)
    heights = [y.shape[2] for y in ys]
    plan = ms_strip_plan(heights, H, world, align_corners)
    pieces = []
    for entry in plan:
        strips = [y[:, :, s0:s1].contiguous() for y, (s0, s1) in zip(ys, entry["src"])]
        pieces.append(ms_flips_image_deaugment_strip(strips, heights, entry["src"], entry["out"], (H, W), group="fliplr", inner_reduction=inner,
                                                     reduction=outer, align_corners=align_corners))
    assert torch.equal(torch.cat(pieces, dim=2), full)
    entry = plan[1]
    short = [(s0 + 2, s1) for s0, s1 in entry["src"]]
    strips = [y[:, :, s0:s1].contiguous() for y, (s0, s1) in zip(ys, short)]
    with pytest.raises(RuntimeError, match="outside"):
        ms_flips_image_deaugment_strip(strips, heights, short, entry["out"], (H, W), group="fliplr", inner_reduction=inner, reduction=outer,
                                       align_corners=align_corners)
    strips = [y[:, :, s0:s1].contiguous() for y, (s0, s1) in zip(ys, entry["src"])]
    with pytest.raises(NotImplementedError):
        ms_flips_image_deaugment_strip(strips, heights, entry["src"], entry["out"], (H, W), group="flipud", inner_reduction=inner, reduction=outer,
                                       align_corners=align_corners)


# ------------------------------------------------------------------ multiscale: nearest mode, gradients, flips fused per scale
GT2 = load_golden("tta2.npz")


def _offs2(kw):
    return [tuple(o) if isinstance(o, list) else o for o in kw["size_offsets"]]


@pytest.mark.parametrize("case", GT2.by_fn("ms_image_augment_grad"), ids=lambda c: c["name"])
def test_ms_image_augment_values_and_gradients(case, dev):
    """ms_image_augment (bilinear both align_corners, nearest) against the unmodified reference: every scale's values and
    the autograd gradient w.r.t. the input (adjoint resize kernels: 'TTA respects gradient flow', tta.py:3-4)."""
    tta = _tta()
    kw = case["kwargs"]
    x = torch.from_numpy(GT2["x"]).to(dev).requires_grad_(True)
    outs = tta.ms_image_augment(x, _offs2(kw), mode=kw["mode"], align_corners=kw["align_corners"])
    tot = 0
    for i, o in enumerate(outs):
        np.testing.assert_allclose(o.detach().cpu().numpy(), GT2[f"{case['name']}_{i}"], rtol=1e-5, atol=1e-6)
        tot = tot + (o * (torch.arange(o.numel(), dtype=torch.float32, device=dev).reshape(o.shape) % 5 + 1.0)).sum()
    tot.backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), GT2[case["name"] + "_grad"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("case", GT2.by_fn("ms_image_deaugment_grad"), ids=lambda c: c["name"])
def test_ms_image_deaugment_values_and_gradients(case, dev):
    tta = _tta()
    kw = case["kwargs"]
    offs = _offs2(kw)
    ins = [torch.from_numpy(GT2[f"fm_{i}"]).to(dev).requires_grad_(True) for i in range(len(offs))]
    out = tta.ms_image_deaugment(ins, offs, reduction=kw["reduction"], mode=kw["mode"], align_corners=kw["align_corners"])
    np.testing.assert_allclose(out.detach().cpu().numpy(), GT2[case["name"]], rtol=1e-5, atol=1e-5)
    (out * (torch.arange(out.numel(), dtype=torch.float32, device=dev).reshape(out.shape) % 7 + 1.0)).sum().backward()
    for i, t in enumerate(ins):
        np.testing.assert_allclose(t.grad.cpu().numpy(), GT2[f"{case['name']}_grad_{i}"], rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("case", GT2.by_fn("ms_flips_image_deaugment"), ids=lambda c: c["name"])
def test_ms_flips_image_deaugment_matches_reference_composition(case, dev):
    """The one-pass flips + multiscale merge (extension) equals the reference's composition ms_image_deaugment([<group>_image_
    deaugment(y) ...]) -- here on odd sizes (36 columns: the fused kernel takes them; (4, -4) offsets: mixed ratios)."""
    from pytorch_toolbelt_amd import _native as N

    tta = _tta()
    kw = case["kwargs"]
    offs = _offs2(kw)
    ys = [torch.from_numpy(GT2[f"fz_{kw['group']}_y{i}"]).to(dev) for i in range(len(offs))]
    before = N.calls
    out = tta.ms_flips_image_deaugment(ys, offs, group=kw["group"], inner_reduction=kw["inner_reduction"], reduction=kw["reduction"],
                                       align_corners=kw["align_corners"])
    assert N.calls == before + 1, "the fused one-pass kernel did not take this configuration"
    np.testing.assert_allclose(out.cpu().numpy(), GT2[case["name"]], rtol=1e-5, atol=1e-5)


def test_ms_flips_fused_equals_composition_at_scale(dev):
    """1024 x 1024, scales 0.75 / 1.0 / 1.25, fliplr and d2: fused == composed (same kernels' arithmetic up to fp32 rounding),
    fallbacks (d4: transposing views; callable reduction; autograd) still give the composed result."""
    tta = _tta()
    torch.manual_seed(5)
    N_ = 1024
    offs = [-N_ // 4, 0, N_ // 4]
    for group, V in (("fliplr", 2), ("d2", 4)):
        ys = [torch.rand((V, 4, N_ + o, N_ + o), device=dev) * 0.9 + 0.05 for o in offs]
        for inner, outer, ac in (("gmean", "gmean", False), ("mean", "mean", True)):
            fused = tta.ms_flips_image_deaugment(ys, offs, group=group, inner_reduction=inner, reduction=outer, align_corners=ac)
            comp = _composed(lambda: tta.ms_image_deaugment([getattr(tta, f"{group}_image_deaugment")(y, reduction=inner) for y in ys], offs, reduction=outer, align_corners=ac))
            assert float((fused - comp).abs().max()) <= 2e-6
            # the same composition written with lazy handles on IS the one-pass kernel (row-preserving groups): one launch, same bits
            from pytorch_toolbelt_amd import _native as N
            before = N.calls
            lit = tta.ms_image_deaugment([getattr(tta, f"{group}_image_deaugment")(y, reduction=inner) for y in ys], offs, reduction=outer, align_corners=ac)
            assert N.calls == before + 1 and torch.equal(lit, fused), (group, inner)
    ys = [torch.rand((8, 2, 64 + o, 64 + o), device=dev) * 0.9 + 0.05 for o in (-16, 0, 16)]
    comp = tta.ms_image_deaugment([tta.d4_image_deaugment(y) for y in ys], [-16, 0, 16])
    assert torch.allclose(tta.ms_flips_image_deaugment(ys, [-16, 0, 16], group="d4"), comp)
    yg = [y[:4].clone().requires_grad_(True) for y in ys]
    out = tta.ms_flips_image_deaugment(yg, [-16, 0, 16], group="d2", inner_reduction="gmean", reduction="mean")
    out.sum().backward()
    assert all(t.grad is not None and torch.isfinite(t.grad).all() for t in yg)
    with pytest.raises(ValueError, match="align_corners"):
        tta.ms_image_augment(ys[1], [8], mode="nearest", align_corners=False)       # F.interpolate's own rule
    with pytest.raises(ValueError, match="align_corners"):
        tta.ms_image_augment(ys[1], [8], mode="area")                                # (area takes align_corners=None only, like nearest)
    with pytest.raises(NotImplementedError):
        tta.ms_image_augment(ys[1], [8], mode="trilinear", align_corners=False)      # not a 4-D mode: F.interpolate refuses it too


@pytest.mark.parametrize("shape", [(1024, 1024), (500, 700), (330, 260)])
def test_ms_flips_tile_shapes_are_bit_identical(shape, dev):
    """The fused flips + multiscale kernel gives the same bits whatever output tile a workgroup owns (64 x 32, 64 x 16, 64 x 64, the
    wide 128 x 16 of ptb_set_tunable(15, 128)): a pixel's arithmetic never depends on the tiling; ragged borders included."""
    from pytorch_toolbelt_amd import _native as N

    tta = _tta()
    lib = N.load()
    H, W = shape
    offs = [(-(H // 4) // 4 * 4, -(W // 4) // 4 * 4), 0, ((H // 4) // 4 * 4, (W // 4) // 4 * 4)]
    g = torch.Generator(device=dev).manual_seed(3)
    try:
        for group, V in (("fliplr", 2), ("flipud", 2)):
            ys = [torch.rand((V * 2, 3, H + (o[0] if o else 0), W + (o[1] if o else 0)), device=dev, generator=g) * 0.9 + 0.05 for o in offs]
            for inner, outer, ac in (("gmean", "gmean", False), ("mean", "mean", True), ("mean", "gmean", False)):
                outs = {}
                for name, (w_, rows) in {"64x32": (64, 32), "64x16": (64, 16), "64x64": (64, 64), "128x16": (128, 32)}.items():
                    assert lib.ptb_set_tunable(15, w_) == 0 and lib.ptb_set_tunable(6, rows) == 0
                    before = N.calls
                    outs[name] = tta.ms_flips_image_deaugment(ys, offs, group=group, inner_reduction=inner, reduction=outer, align_corners=ac)
                    assert N.calls == before + 1
                comp = _composed(lambda: tta.ms_image_deaugment([getattr(tta, f"{group}_image_deaugment")(y, reduction=inner) for y in ys], offs, reduction=outer, align_corners=ac))
                assert float((outs["64x32"] - comp).abs().max()) <= 2e-6
                for name, o in outs.items():
                    assert torch.equal(o, outs["64x32"]), (group, inner, outer, name)
    finally:
        lib.ptb_set_tunable(15, 128)
        lib.ptb_set_tunable(6, 32)


def test_ms_flips_gmean_of_extreme_values(dev):
    """The two-view gmean inside the fused kernel is sqrt(a b) with a scaled product; pairs whose product leaves the safe range --
    vanishing probabilities, zeros, values that are no probabilities -- take sqrt(a) sqrt(b): same result as the composed path's
    exp(mean(log)) to rounding, no underflow to 0, no overflow to inf."""
    tta = _tta()
    offs = [-64, 0, 64]
    g = torch.Generator(device=dev).manual_seed(8)
    ys = [torch.rand((2, 2, 256 + o, 256 + o), device=dev, generator=g) * 0.9 + 0.05 for o in offs]
    for y in ys:      # sprinkle extremes over whole regions (a wave takes the exact branch as soon as one of its pairs is odd)
        y[0, :, 10:40, 20:90] = 1e-30
        y[1, :, 10:40, 20:90] = 3e-31
        y[:, :, 100:120, 5:60] = 1e-38
        y[0, :, 150:170, 100:160] = 0.0
        y[:, :, 200:220, 30:80] = 4e19
    fused = tta.ms_flips_image_deaugment(ys, offs, group="fliplr", inner_reduction="gmean", reduction="mean", align_corners=False)
    comp = _composed(lambda: tta.ms_image_deaugment([tta.fliplr_image_deaugment(y, reduction="gmean") for y in ys], offs, reduction="mean", align_corners=False))
    assert torch.isfinite(fused).all()
    torch.testing.assert_close(fused, comp, rtol=2e-5, atol=1e-37)
    # the smallest normal probabilities survive the inner gmean (their product, 4e-76, is far below fp32): check it directly
    same = torch.full((2, 1, 64, 64), 2e-38, device=dev)
    inner = tta.ms_flips_image_deaugment([same], [0], group="fliplr", inner_reduction="gmean", reduction="mean")
    assert float(inner.min()) > 1.9e-38 and float(inner.max()) < 2.1e-38


# ------------------------------------------------------------------ stacks longer than 8, reductions with their eps argument
GT3 = load_golden("tta3.npz")


@pytest.mark.parametrize("case", GT3.cases, ids=lambda c: c["name"])
def test_long_stacks_and_explicit_eps_values_and_gradients(case, dev):
    """`_deaugment_averaging` over 10 / 12 / 17 stacked predictions (tencrop, large ensembles) with every reduction, and
    harmonic_mean / logodd_mean with a caller-chosen eps along dim 0 and 1: values and autograd gradients of the unmodified
    reference (ptb_stack_reduce / ptb_stack_reduce_bwd, csrc/ptb_stack.hip)."""
    from pytorch_toolbelt_amd import _native as N
    from pytorch_toolbelt_amd.inference import functional as F

    kw = case["kwargs"]
    x = torch.from_numpy(GT3[case["inputs"][0]]).to(dev).requires_grad_(True)
    before = N.calls
    if case["fn"] == "deaugment_averaging":
        out = _tta()._deaugment_averaging(x, kw["reduction"])
        period = 5
    else:
        out = getattr(F, case["fn"])(x, dim=kw["dim"], eps=kw["eps"])
        period = 3
    assert N.calls > before, "no native launch"
    np.testing.assert_allclose(out.detach().cpu().numpy(), GT3[case["output"]], rtol=1e-5, atol=1e-6)
    (out * (torch.arange(out.numel(), dtype=torch.float32, device=dev).reshape(out.shape) % period + 1.0)).sum().backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), GT3[case["output"] + "_grad"], rtol=2e-4, atol=1e-5)


# ------------------------------------------------------------------ multiscale with mode="bicubic"
GT6 = load_golden("tta6.npz")


@pytest.mark.parametrize("case", GT6.cases, ids=lambda c: c["name"])
def test_multiscale_area_and_nearest_exact_values_and_gradients(case, dev):
    """ms_image_augment / ms_image_deaugment with mode="nearest-exact" and mode="area" (ptb_resize_nearest_exact, ptb_resize_area and
    their adjoints -- with "bilinear", "bicubic" and "nearest" every mode F.interpolate takes for a 4-D tensor; the reference forwards
    any, inference/tta.py:599-621, 645-689) against the unmodified reference: values of every scale / of the merged map, and the
    autograd gradients."""
    tta = _tta()
    kw = case["kwargs"]
    offs = _offs2(kw)
    if case["fn"] == "ms_image_augment_grad":
        x = torch.from_numpy(GT6["x"]).to(dev).requires_grad_(True)
        outs = tta.ms_image_augment(x, offs, mode=kw["mode"], align_corners=None)
        tot = 0
        for i, o in enumerate(outs):
            want = GT6[f"{case['name']}_{i}"]
            if kw["mode"] == "nearest-exact":
                assert np.array_equal(o.detach().cpu().numpy(), want)          # a gather: bit-exact
            else:
                np.testing.assert_allclose(o.detach().cpu().numpy(), want, rtol=1e-6, atol=1e-6)
            tot = tot + (o * (torch.arange(o.numel(), dtype=torch.float32, device=dev).reshape(o.shape) % 5 + 1.0)).sum()
        tot.backward()
        np.testing.assert_allclose(x.grad.cpu().numpy(), GT6[case["name"] + "_grad"], rtol=1e-4, atol=2e-5)
    else:
        ins = [torch.from_numpy(GT6[f"fm_{i}"]).to(dev).requires_grad_(True) for i in range(len(offs))]
        out = tta.ms_image_deaugment(ins, offs, reduction=kw["reduction"], mode=kw["mode"], align_corners=None)
        np.testing.assert_allclose(out.detach().cpu().numpy(), GT6[case["name"]], rtol=1e-5, atol=1e-5)
        (out * (torch.arange(out.numel(), dtype=torch.float32, device=dev).reshape(out.shape) % 7 + 1.0)).sum().backward()
        for i, t in enumerate(ins):
            np.testing.assert_allclose(t.grad.cpu().numpy(), GT6[f"{case['name']}_grad_{i}"], rtol=2e-4, atol=2e-5)


GT4 = load_golden("tta4.npz")


@pytest.mark.parametrize("case", GT4.cases, ids=lambda c: c["name"])
def test_multiscale_bicubic_values_and_gradients(case, dev):
    """ms_image_augment / ms_image_deaugment with mode="bicubic" (ptb_resize_bicubic and its adjoint) against the unmodified
    reference: values of every scale / of the merged map, and the autograd gradients."""
    tta = _tta()
    kw = case["kwargs"]
    offs = _offs2(kw)
    if case["fn"] == "ms_image_augment_grad":
        x = torch.from_numpy(GT4["x"]).to(dev).requires_grad_(True)
        outs = tta.ms_image_augment(x, offs, mode="bicubic", align_corners=kw["align_corners"])
        tot = 0
        for i, o in enumerate(outs):
            np.testing.assert_allclose(o.detach().cpu().numpy(), GT4[f"{case['name']}_{i}"], rtol=1e-5, atol=2e-6)
            tot = tot + (o * (torch.arange(o.numel(), dtype=torch.float32, device=dev).reshape(o.shape) % 5 + 1.0)).sum()
        tot.backward()
        np.testing.assert_allclose(x.grad.cpu().numpy(), GT4[case["name"] + "_grad"], rtol=1e-4, atol=2e-5)
    else:
        ins = [torch.from_numpy(GT4[f"fm_{i}"]).to(dev).requires_grad_(True) for i in range(len(offs))]
        out = tta.ms_image_deaugment(ins, offs, reduction=kw["reduction"], mode="bicubic", align_corners=kw["align_corners"])
        np.testing.assert_allclose(out.detach().cpu().numpy(), GT4[case["name"]], rtol=1e-5, atol=1e-5)
        (out * (torch.arange(out.numel(), dtype=torch.float32, device=dev).reshape(out.shape) % 7 + 1.0)).sum().backward()
        for i, t in enumerate(ins):
            np.testing.assert_allclose(t.grad.cpu().numpy(), GT4[f"{case['name']}_grad_{i}"], rtol=2e-4, atol=2e-5)
