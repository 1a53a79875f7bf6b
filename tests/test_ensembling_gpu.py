"""GPU parity of inference/ensembling: Ensembler / ApplySigmoidTo / ApplySoftmaxTo through the HIP
list-reduce kernel vs the golden vectors of the reference (tests/golden/ensembling.npz) and the numpy oracle.
Tolerance 1e-5 absolute (BASELINE north_star) on O(1) values; linear reductions without activation are bit-exact."""
import numpy as np
import pytest
import torch
from torch import nn

from conftest import load_golden
from oracle import ensembling_oracle as NO

pytestmark = pytest.mark.gpu

GN = load_golden("ensembling.npz")
TOL = dict(rtol=1e-5, atol=1e-5)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture()
def native():
    from pytorch_toolbelt_amd import _native as N

    lib = N.load()
    yield N
    lib.ptb_set_tunable(1, 0)


class Affine(nn.Module):
    """Same stand-in model as oracle/make_golden.py:_Affine."""

    def __init__(self, k, b, kind):
        super().__init__()
        self.k, self.b, self.kind = k, b, kind

    def forward(self, x):
        y = x * self.k + self.b
        if self.kind == "tensor":
            return y
        if self.kind == "list":
            return [y, y * 0.5 - 0.25]
        return {"logits": y, "aux": y * 0.5 - 0.25}


def build(kw):
    from pytorch_toolbelt_amd.inference import ensembling as E

    models = [Affine(k, b, kw["kind"]) for k, b in kw["coeffs"]]
    key = "logits" if kw["kind"] == "dict" else 0
    if kw["wrap"] == "sigmoid":
        models = [E.ApplySigmoidTo(m, output_key=key, temperature=kw["temperature"]) for m in models]
    elif kw["wrap"] == "softmax":
        models = [E.ApplySoftmaxTo(m, output_key=key, dim=1, temperature=kw["temperature"]) for m in models]
    return E.Ensembler(models, reduction=kw["reduction"], outputs=kw["outputs"])


@pytest.mark.parametrize("scalar", [0, 1])
@pytest.mark.parametrize("case", GN.by_fn("ensembler"), ids=lambda c: f"{c['name']}-{c['kwargs']['kind']}-{c['kwargs']['wrap']}-{c['kwargs']['reduction']}")
def test_golden_ensembler(case, scalar, dev, native):
    kw, n = case["kwargs"], case["name"]
    native.load().ptb_set_tunable(1, scalar)
    ens = build(kw)
    x = torch.from_numpy(GN[kw["input"]]).to(dev)
    before = native.calls
    with torch.no_grad():
        out = ens(x)
    assert native.calls > before, "the HIP ensemble kernel did not run"
    if kw["keys"] is None:
        np.testing.assert_allclose(out.cpu().numpy(), GN[f"{n}_out"], **TOL)
    else:
        assert list(out.keys() if isinstance(out, dict) else range(len(out))) == kw["keys"]
        for key in kw["keys"]:
            np.testing.assert_allclose(out[key].cpu().numpy(), GN[f"{n}_out_{key}"], **TOL)


def test_linear_ensemble_is_bit_exact_and_reads_in_place(dev, native):
    from pytorch_toolbelt_amd.inference.ensembling import Ensembler

    torch.manual_seed(0)
    outs = [torch.randn((3, 7, 33, 20), device=dev) for _ in range(6)]

    class Fixed(nn.Module):
        def __init__(self, t):
            super().__init__()
            self.t = t

        def forward(self, x):
            return self.t

    for red in ("sum", "mean"):
        ens = Ensembler([Fixed(t) for t in outs], reduction=red)
        got = ens(outs[0])
        stack = torch.stack(outs)
        want = stack.sum(0) if red == "sum" else stack.mean(0)
        seq = outs[0].clone()   # the kernel adds in list order, like a python loop over the models
        for t in outs[1:]:
            seq = seq + t
        if red == "sum":
            assert torch.equal(got, seq)
        else:                    # (torch divides by a scalar as a multiplication by its reciprocal: 1 ulp apart)
            assert torch.allclose(got, seq / len(outs), rtol=1e-6, atol=1e-7)
        assert torch.allclose(got, want, atol=1e-6)


@pytest.mark.parametrize("shape,dim", [((2, 20, 9, 7), 1), ((3, 6, 10, 10), 1), ((4, 5), 1), ((2, 3, 8, 8), -1), ((5, 8, 6), 0)])
@pytest.mark.parametrize("reduction", ["mean", "gmean"])
def test_softmax_variants_match_oracle(shape, dim, reduction, dev, native):
    """C > 16, HW % 4 != 0, 2-D logits, softmax over the last / first dim: the generic kernel and the factorisation."""
    from pytorch_toolbelt_amd.inference.ensembling import ApplySoftmaxTo, Ensembler

    rng = np.random.default_rng(1)
    xs = [rng.standard_normal(shape).astype(np.float32) * 3 for _ in range(3)]

    class Fixed(nn.Module):
        def __init__(self, t):
            super().__init__()
            self.t = t

        def forward(self, x):
            return {"logits": self.t.clone()}

    ens = Ensembler([ApplySoftmaxTo(Fixed(torch.from_numpy(x).to(dev)), dim=dim, temperature=0.7) for x in xs], reduction=reduction)
    before = native.calls
    got = ens(None)["logits"].cpu().numpy()
    assert native.calls == before + 1          # ONE fused launch
    want = NO.ensemble([NO.softmax_to(x, 0.7, dim) for x in xs], reduction)
    np.testing.assert_allclose(got, want, **TOL)


def test_standalone_wrappers_mixed_activations_half_and_many_models(dev, native):
    from pytorch_toolbelt_amd.inference import ensembling as E

    rng = np.random.default_rng(2)
    x = torch.from_numpy(rng.standard_normal((2, 4, 16, 12)).astype(np.float32)).to(dev)
    base = Affine(1.1, -0.2, "dict")
    # standalone wrappers == the reference's formulas
    s = E.ApplySigmoidTo(base, temperature=2.0)(x)["logits"]
    np.testing.assert_allclose(s.cpu().numpy(), NO.sigmoid_to((x * 1.1 - 0.2).cpu().numpy(), 2.0), **TOL)
    p = E.ApplySoftmaxTo(base, temperature=0.5)(x)["logits"]
    np.testing.assert_allclose(p.cpu().numpy(), NO.softmax_to((x * 1.1 - 0.2).cpu().numpy(), 0.5, 1), **TOL)
    # one sigmoid-wrapped and one softmax-wrapped model: activations applied per model, then reduced
    ens = E.Ensembler([E.ApplySigmoidTo(Affine(1.0, 0.0, "dict")), E.ApplySoftmaxTo(Affine(0.5, 0.1, "dict"))], reduction="mean", outputs=["logits"])
    got = ens(x)["logits"].cpu().numpy()
    xn = x.cpu().numpy()
    want = NO.ensemble([NO.sigmoid_to(xn), NO.softmax_to((xn * np.float32(0.5)).astype(np.float32) + np.float32(0.1))], "mean")
    np.testing.assert_allclose(got, want, **TOL)
    # half inputs are evaluated in fp32 and cast back
    ens = E.Ensembler([Affine(1.0, 0.0, "tensor"), Affine(0.5, 0.25, "tensor")], reduction="mean")
    h = ens(x.half())
    assert h.dtype == torch.float16
    np.testing.assert_allclose(h.float().cpu().numpy(), ((x.half() + (x.half() * 0.5 + 0.25)).float() / 2).cpu().numpy(), atol=2e-3)
    # more than 16 models: stack path (still the HIP stack reduce underneath)
    many = E.Ensembler([Affine(1.0, 0.01 * i, "tensor") for i in range(20)], reduction="mean")
    want = x + float(np.mean([np.float32(0.01 * i) for i in range(20)]))
    assert torch.allclose(many(x), want, atol=1e-5)
    # reduction None keeps the stack, a callable is applied to it (ensembling.py:112, tta.py:63-96)
    assert E.Ensembler([Affine(1.0, 0.0, "tensor"), Affine(2.0, 0.0, "tensor")], reduction=None)(x).shape == (2, *x.shape)
    med = E.Ensembler([Affine(1.0, 0.0, "tensor"), Affine(2.0, 0.0, "tensor"), Affine(3.0, 0.0, "tensor")], reduction=lambda t, dim: t.median(dim=dim).values)(x)
    assert torch.allclose(med, x * 2.0)
    with pytest.raises(KeyError):
        E.Ensembler([Affine(1.0, 0.0, "tensor")], reduction="median")(x)


def test_autograd_flows_through_the_ensemble(dev):
    from pytorch_toolbelt_amd.inference import ensembling as E

    torch.manual_seed(3)
    x = torch.randn((2, 3, 8, 8), device=dev, requires_grad=True)
    ens = E.Ensembler([E.ApplySigmoidTo(Affine(1.0, 0.0, "dict")), E.ApplySigmoidTo(Affine(0.5, 0.2, "dict"))], reduction="gmean", outputs=["logits"])
    y = ens(x)["logits"]
    y.sum().backward()
    xr = x.detach().clone().requires_grad_(True)
    a, b = torch.sigmoid(xr), torch.sigmoid(xr * 0.5 + 0.2)
    ref = torch.exp((torch.log(a) + torch.log(b)) / 2)
    ref.sum().backward()
    assert torch.allclose(y, ref, atol=1e-6) and torch.allclose(x.grad, xr.grad, atol=1e-5)
