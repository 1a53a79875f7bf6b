"""CPU: pin the numpy oracle (oracle/) against the golden vectors produced by the unmodified reference
(tests/golden/*.npz, generator oracle/make_golden.py) and the reference's own known-answer tests."""
import hashlib

import numpy as np
import pytest

from conftest import load_golden
from oracle import losses_oracle as LO
from oracle import tiles_oracle as TO
from oracle import tta_oracle as AO

GT = load_golden("tiles.npz")
GA = load_golden("tta.npz")
GL = load_golden("losses.npz")


# ------------------------------------------------------------------ tiles
@pytest.mark.parametrize("case", GT.by_fn("geometry"), ids=lambda c: c["name"])
def test_geometry_bit_exact(case):
    kw = case["kwargs"]
    g = TO.slicer_geometry(kw["image_shape"], kw["tile_size"], kw["tile_step"], kw.get("image_margin", 0))
    n = case["name"]
    assert np.array_equal(g["crops"], GT[f"{n}_crops"])
    assert np.array_equal(g["bbox_crops"], GT[f"{n}_bbox"])
    meta = [*g["margins"], *g["target_shape"], *g["tile_size"], *g["tile_step"]]
    assert meta == GT[f"{n}_meta"].tolist()


@pytest.mark.parametrize("case", GT.by_fn("pyramid"), ids=lambda c: c["name"])
def test_pyramid_bitwise(case):
    w, h = case["kwargs"]["width"], case["kwargs"]["height"]
    W, Dc, De = TO.pyramid_window(w, h)
    n = case["name"]
    assert np.array_equal(W, GT[f"{n}_W"]) and np.array_equal(Dc, GT[f"{n}_Dc"]) and np.array_equal(De, GT[f"{n}_De"])


@pytest.mark.parametrize("case", GT.by_fn("pyramid_digest"), ids=lambda c: c["name"])
def test_pyramid_digest(case):
    w, h = case["kwargs"]["width"], case["kwargs"]["height"]
    W, _, _ = TO.pyramid_window(w, h)
    assert hashlib.sha256(np.ascontiguousarray(W).tobytes()).hexdigest() == str(GT[f"{case['name']}_sha256"])
    assert np.array_equal(W, W.T)  # reference tests/test_tiles.py:51


@pytest.mark.parametrize("case", GT.by_fn("split_merge"), ids=lambda c: c["name"])
def test_split_cut_merge(case):
    kw, n = case["kwargs"], case["name"]
    img = GT[f"{n}_image"]
    g = TO.slicer_geometry(img.shape, kw["tile_size"], kw["tile_step"])
    w = TO.pyramid_window(*kw["tile_size"])[0] if kw["weight"] == "pyramid" else TO.mean_window(*kw["tile_size"])
    tiles = TO.split(img, g)
    assert np.array_equal(np.stack(tiles), GT[f"{n}_tiles"])
    assert np.array_equal(np.stack([TO.cut_patch(img, g, i) for i in range(len(tiles))]), GT[f"{n}_cut"])
    assert np.array_equal(GT[f"{n}_iter_tiles"], GT[f"{n}_tiles"]) and np.array_equal(GT[f"{n}_iter_coords"], g["crops"])
    assert np.array_equal(TO.slicer_merge(tiles, g, w, img.shape, np.float32), GT[f"{n}_merge_f32"])
    assert np.array_equal(TO.slicer_merge(tiles, g, w, img.shape, np.uint8), GT[f"{n}_merge_u8"])
    assert np.array_equal(TO.slicer_merge(list(GT[f"{n}_ftiles"]), g, w, img.shape, np.float32), GT[f"{n}_fmerge"])


@pytest.mark.parametrize("case", GT.by_fn("tile_merger"), ids=lambda c: c["name"])
def test_tile_merger(case):
    kw, n = case["kwargs"], case["name"]
    g = TO.slicer_geometry(kw["image_shape"], kw["tile_size"], kw["tile_step"])
    w = TO.pyramid_window(*kw["tile_size"])[0] if kw["weight"] == "pyramid" else TO.mean_window(*kw["tile_size"])
    st = TO.merger_new(g["target_shape"], kw["channels"], w)
    pred = GT[f"{n}_pred"]
    for b0 in range(0, len(pred), kw["batch"]):
        TO.merger_integrate(st, pred[b0:b0 + kw["batch"]], g["crops"][b0:b0 + kw["batch"]])
    # numpy fp32 mul+add == torch fp32 mul+add, same order: bit-exact
    assert np.array_equal(st["image"], GT[f"{n}_image"])
    assert np.array_equal(st["norm_mask"], GT[f"{n}_norm"])
    assert np.array_equal(TO.merger_merge(st), GT[f"{n}_merged"])


def test_merger_uncovered_is_nan():
    st = TO.merger_new((8, 8), 1, np.ones((4, 4), np.float32))
    TO.merger_integrate(st, np.ones((1, 1, 4, 4), np.float32), [(0, 0, 4, 4)])
    m = TO.merger_merge(st)
    assert np.isnan(m[0, 7, 7]) and m[0, 0, 0] == 1.0  # quirk Q5: no eps clamp


def test_geometry_errors():
    with pytest.raises(ValueError):
        TO.slicer_geometry((10, 10), 4, 0)           # default tile_step=0 always raises
    with pytest.raises(ValueError):
        TO.slicer_geometry((10, 10), 4, 5)
    with pytest.raises(ValueError):
        TO.slicer_geometry((10, 10), (4, 4, 4), 2)


# ------------------------------------------------------------------ tta
def _tol(out):
    return dict(rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("case", GA.by_fn("image_augment"), ids=lambda c: c["name"])
def test_image_augment(case):
    out = AO.image_augment(GA[case["inputs"][0]], case["kwargs"]["group"])
    assert np.array_equal(out, GA[case["output"]])


@pytest.mark.parametrize("case", GA.by_fn("image_deaugment"), ids=lambda c: c["name"])
def test_image_deaugment(case):
    out = AO.image_deaugment(GA[case["inputs"][0]], case["kwargs"]["group"], case["kwargs"]["reduction"])
    ref = GA[case["output"]]
    assert out.shape == ref.shape
    np.testing.assert_allclose(out, ref, equal_nan=True, **_tol(out))


@pytest.mark.parametrize("case", GA.by_fn("labels_deaugment"), ids=lambda c: c["name"])
def test_labels_deaugment(case):
    out = AO.labels_deaugment(GA[case["inputs"][0]], case["kwargs"]["group"], case["kwargs"]["reduction"])
    np.testing.assert_allclose(out, GA[case["output"]], **_tol(out))


def test_d4_labels_quirk_b7_twice():
    x = np.repeat(np.arange(8, dtype=np.float32), 1)[:, None]
    assert AO.labels_deaugment(x, "d4", "mean")[0, 0] == pytest.approx(3.625)


def test_fivecrop_and_ms(golden_tta):
    G = golden_tta
    assert np.array_equal(AO.fivecrop_image_augment(G["x_sq"], (8, 10)), G["fivecrop_aug"])
    for c in G.by_fn("ms_image_augment"):
        outs = AO.ms_image_augment(G["x_ms"], c["kwargs"]["size_offsets"], c["kwargs"]["align_corners"])
        for o, k in zip(outs, c["output"]):
            np.testing.assert_allclose(o, G[k], rtol=1e-5, atol=1e-6)
    for c in G.by_fn("ms_image_deaugment"):
        kw = c["kwargs"]
        out = AO.ms_image_deaugment([G[k] for k in c["inputs"]], kw["size_offsets"], kw["reduction"], kw["align_corners"], kw["stride"])
        np.testing.assert_allclose(out, G[c["output"]], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("case", GA.by_fn("reduction"), ids=lambda c: c["name"])
def test_reductions(case):
    fn = getattr(AO, case["kwargs"]["which"])
    np.testing.assert_allclose(fn(GA["red_stack"], 0), GA[case["output"]], rtol=2e-6, atol=1e-7)


def test_reference_kat_tta_identity():
    """reference tests/test_tta.py:31-68: deaugment(augment(x)) == x, atol=rtol=1e-6."""
    rng = np.random.default_rng(3)
    x = rng.random((4, 3, 24, 24), dtype=np.float32)
    for grp in ("fliplr", "flipud", "flips", "d2", "d4"):
        y = AO.image_deaugment(AO.image_augment(x, grp), grp, "mean")
        np.testing.assert_allclose(y, x, atol=1e-6, rtol=1e-6)


def test_reference_kat_tta_labels():
    """reference tests/test_tta.py:71-108 (SumAll model on a 4x4 integer matrix)."""
    x = np.array([[1, 2, 3, 4], [5, 6, 7, 8], [9, 0, 1, 2], [3, 4, 5, 6]], dtype=np.float32)[None, None]
    model = lambda t: t.sum(axis=(1, 2, 3))
    assert int(AO.labels_deaugment(model(AO.image_augment(x, "d4")), "d4")[0]) == int(x.sum())
    assert int(AO.labels_deaugment(model(AO.image_augment(x, "fliplr")), "fliplr")[0]) == int(x.sum())
    five = AO.labels_deaugment(model(AO.fivecrop_image_augment(x, (2, 2))), "fivecrop")
    assert int(five[0]) == ((1 + 2 + 5 + 6) + (3 + 4 + 7 + 8) + (9 + 0 + 3 + 4) + (1 + 2 + 5 + 6) + (6 + 7 + 0 + 1)) / 5


# ------------------------------------------------------------------ losses
def _inputs(G, case):
    return [G[k] for k in case["inputs"]]


def _kw(G, case):
    kw = dict(case["kwargs"])
    if kw.pop("class_weights", None):
        kw["class_weights"] = G["class_weights"]
    return kw


LOSS_FNS = {
    "focal_loss_with_logits": LO.focal_loss_with_logits,
    "binary_focal_loss": LO.binary_focal_loss,
    "softmax_focal_loss_with_logits": LO.softmax_focal_loss_with_logits,
    "soft_dice_score": LO.soft_dice_score,
    "soft_jaccard_score": LO.soft_jaccard_score,
    "dice_loss": lambda a, b, **kw: LO.dice_loss(a, b, **kw),
    "jaccard_loss": lambda a, b, **kw: LO.jaccard_loss(a, b, **kw),
    "lovasz_hinge": LO.lovasz_hinge,
}


@pytest.mark.parametrize("case", GL.by_fn(*LOSS_FNS), ids=lambda c: c["name"])
def test_losses(case):
    a, b = _inputs(GL, case)
    out = LOSS_FNS[case["fn"]](a, b, **_kw(GL, case))
    np.testing.assert_allclose(out, GL[case["output"]], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("case", GL.by_fn("lovasz_softmax"), ids=lambda c: c["name"])
def test_lovasz_softmax(case):
    a, b = _inputs(GL, case)
    kw = dict(case["kwargs"])
    out = LO.lovasz_softmax(a, b, per_image=kw.get("per_image", False), ignore_index=kw.get("ignore"))
    np.testing.assert_allclose(out, GL[case["output"]], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize(
    "y_true,y_pred,expected",
    [([1, 1, 1, 1], [1, 1, 1, 1], 1.0), ([0, 1, 1, 0], [0, 1, 1, 0], 1.0), ([1, 1, 1, 1], [1, 1, 0, 0], 0.5)],
)
def test_reference_kat_jaccard(y_true, y_pred, expected):
    """reference tests/test_losses.py:37-50."""
    assert float(LO.soft_jaccard_score(np.float32(y_pred), np.float32(y_true), eps=1e-5)) == pytest.approx(expected, 1e-5)


@pytest.mark.parametrize(
    "y_true,y_pred,expected",
    [([1, 1, 1, 1], [1, 1, 1, 1], 1.0), ([0, 1, 1, 0], [0, 1, 1, 0], 1.0), ([1, 1, 1, 1], [1, 1, 0, 0], 2.0 / 3.0)],
)
def test_reference_kat_dice(y_true, y_pred, expected):
    """reference tests/test_losses.py:69-80."""
    assert float(LO.soft_dice_score(np.float32(y_pred), np.float32(y_true), eps=1e-5)) == pytest.approx(expected, 1e-5)


def test_reference_kat_region_losses():
    """reference tests/test_losses.py:83-209: ideal -> 0, worst -> 1, 1 - 1/3, empty class -> 0."""
    eps = 1e-5
    f = np.float32
    for fn in (LO.dice_loss, LO.jaccard_loss):
        assert fn(f([1, 1, 1]).reshape(1, 1, 1, -1), np.array([1, 1, 1]).reshape(1, 1, 1, -1), "binary", from_logits=False) == pytest.approx(0, abs=eps)
        assert fn(f([0, 0, 0]).reshape(1, 1, 1, -1), np.array([0, 0, 0]).reshape(1, 1, 1, -1), "binary", from_logits=False) == pytest.approx(0, abs=eps)
        assert fn(f([1, 1, 1]).reshape(1, 1, -1), np.array([0, 0, 0]).reshape(1, 1, 1, -1), "binary", from_logits=False) == pytest.approx(0, abs=eps)
        assert fn(f([1, 0, 1]).reshape(1, 1, -1), np.array([0, 1, 0]).reshape(1, 1, 1, -1), "binary", from_logits=False) == pytest.approx(1, abs=eps)
    yp = f([[[1, 0, 1, 0], [0, 1, 0, 1]]])
    assert LO.jaccard_loss(yp, np.array([[1, 1, 0, 0]]), "multiclass", from_logits=False) == pytest.approx(1 - 1 / 3, abs=eps)
    yp = f([[[0, 1, 1, 0], [0, 1, 1, 0]]])
    yt = f([[[1, 1, 0, 0], [1, 1, 0, 0]]])
    assert LO.jaccard_loss(yp, yt, "multilabel", from_logits=False) == pytest.approx(1 - 1 / 3, abs=eps)


def test_reference_kat_focal_ordering():
    """reference tests/test_losses.py:11-34."""
    t = np.array([1, 0, 1])
    assert LO.focal_loss_with_logits(np.float32([10, -10, 10]), t) < LO.focal_loss_with_logits(np.float32([-1, 2, 0]), t)
    good = np.float32([[0, 10, 0], [10, 0, 0], [0, 0, 10]])
    bad = np.float32([[0, -10, 0], [0, 10, 0], [0, 0, 10]])
    lab = np.array([1, 0, 2])
    assert LO.softmax_focal_loss_with_logits(good, lab) < LO.softmax_focal_loss_with_logits(bad, lab)


# ------------------------------------------------------------------ loop edges
from oracle import edges_oracle as EO  # noqa: E402

GE = load_golden("edges.npz")


def _edge_geom(kw):
    return TO.slicer_geometry(kw["image_shape"], kw["tile_size"], kw["tile_step"], kw.get("image_margin", 0))


@pytest.mark.parametrize("case", GE.by_fn("tiles_to_batch"), ids=lambda c: c["name"])
def test_edges_front_bit_exact(case):
    kw, n = case["kwargs"], case["name"]
    img = GE[f"{n}_image"]
    scale = GE[f"{n}_scale"] if kw.get("affine") else None
    bias = GE[f"{n}_bias"] if kw.get("affine") else None
    got = EO.tiles_to_batch(img, _edge_geom(kw), kw.get("indices"), scale, bias, kw.get("value", 0), kw.get("augment"))
    want = GE[f"{n}_out"]
    assert got.dtype == np.float32 and got.shape == want.shape
    assert np.array_equal(got, want)


@pytest.mark.parametrize("case", GE.by_fn("merge_crop"), ids=lambda c: c["name"])
def test_edges_back_bit_exact(case):
    kw, n = case["kwargs"], case["name"]
    g = _edge_geom(kw)
    w = TO.pyramid_window(*g["tile_size"])[0] if kw["weight"] == "pyramid" else TO.mean_window(*g["tile_size"])
    st = TO.merger_new(g["target_shape"], kw["channels"], w)
    pred = GE[f"{n}_pred"]
    for b0 in range(0, len(pred), kw["batch"]):
        TO.merger_integrate(st, pred[b0:b0 + kw["batch"]], g["crops"][b0:b0 + kw["batch"]])
    assert np.array_equal(EO.merge_crop(st, g, kw["image_shape"], "hwc", "float32"), GE[f"{n}_hwc_f32"])
    assert np.array_equal(EO.merge_crop(st, g, kw["image_shape"], "hwc", "uint8"), GE[f"{n}_hwc_u8"])
    assert np.array_equal(EO.merge_crop(st, g, kw["image_shape"], "hwc", "argmax_i64"), GE[f"{n}_argmax"])
    chw = EO.merge_crop(st, g, kw["image_shape"], "chw", "float32")
    assert np.array_equal(np.moveaxis(chw, 0, -1), GE[f"{n}_hwc_f32"])


def test_cast_u8_matches_numpy_astype():
    import warnings

    v = np.array([-300.2, -1.5, -0.5, 0.0, 0.99, 1.0, 127.5, 255.99, 256.0, 300.7, 65536.5, 1e10, -1e10, np.nan, np.inf, -np.inf],
                 dtype=np.float32)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = v.astype(np.uint8)
    got = EO.cast_u8(v)
    inside = np.abs(v) < 2147483648.0
    assert np.array_equal(got[inside], want[inside])   # outside int32 the C cast is undefined; the oracle pins 0
    assert (got[~inside] == 0).all()


# ------------------------------------------------------------------ ensembling
from oracle import ensembling_oracle as NO  # noqa: E402

GN = load_golden("ensembling.npz")


def affine_outputs(x, k, b, kind):
    """The golden generator's stand-in model (oracle/make_golden.py:_Affine) in float32 numpy."""
    y = (x * np.float32(k)).astype(np.float32) + np.float32(b)
    aux = (y * np.float32(0.5)).astype(np.float32) - np.float32(0.25)
    return y if kind == "tensor" else ({"logits": y, "aux": aux} if kind == "dict" else [y, aux])


@pytest.mark.parametrize("case", GN.by_fn("ensembler"), ids=lambda c: f"{c['name']}-{c['kwargs']['kind']}-{c['kwargs']['wrap']}-{c['kwargs']['reduction']}")
def test_ensembling_oracle(case):
    kw, n = case["kwargs"], case["name"]
    x = GN[kw["input"]]
    outs = [affine_outputs(x, k, b, kw["kind"]) for k, b in kw["coeffs"]]
    act_key = None if kw["wrap"] is None else ("logits" if kw["kind"] == "dict" else 0)

    def activated(o, key):
        v = o if key is None else o[key]
        if kw["wrap"] is not None and key == act_key:
            return NO.sigmoid_to(v, kw["temperature"]) if kw["wrap"] == "sigmoid" else NO.softmax_to(v, kw["temperature"], 1)
        return v

    if kw["keys"] is None:
        got = NO.ensemble([activated(o, None) for o in outs], kw["reduction"])
        np.testing.assert_allclose(got, GN[f"{n}_out"], rtol=2e-6, atol=2e-6)
    else:
        for key in kw["keys"]:
            got = NO.ensemble([activated(o, key) for o in outs], kw["reduction"])
            np.testing.assert_allclose(got, GN[f"{n}_out_{key}"], rtol=2e-6, atol=2e-6)


# ------------------------------------------------------------------ remaining elementwise losses
from oracle import pointwise_oracle as PO  # noqa: E402

GL2 = load_golden("losses2.npz")


def pointwise_kwargs(kw, C):
    """Materialise the 'chan' / 'scalar' weight markers of oracle/make_golden.py:gen_losses2."""
    k2 = dict(kw)
    for key, vec in (("weight", "wvec"), ("pos_weight", "pwvec")):
        if k2.get(key) == "chan":
            k2[key] = GL2[vec].reshape(C, 1, 1)
        elif k2.get(key) == "scalar":
            k2[key] = np.float32(1.7)
    return k2


POINTWISE_ORACLES = {"soft_bce": PO.soft_bce, "balanced_bce": PO.balanced_bce, "qfl": PO.quality_focal, "wing": PO.wing,
                     "logcosh": PO.log_cosh, "soft_ce": PO.soft_ce}


@pytest.mark.parametrize("case", GL2.cases, ids=lambda c: c["name"])
def test_pointwise_loss_oracle(case):
    a, b = GL2[case["inputs"][0]], GL2[case["inputs"][1]]
    kw = pointwise_kwargs(case["kwargs"], a.shape[1] if a.ndim > 1 else 1)
    got = POINTWISE_ORACLES[case["fn"]](a, b, **kw)
    want = GL2[case["name"]]
    assert np.shape(got) == want.shape
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-5)


# ------------------------------------------------------------------ focal, activation="softmax" (functional.py:61-66)
GL4 = load_golden("losses4.npz")


@pytest.mark.parametrize("case", GL4.cases, ids=lambda c: c["name"])
def test_focal_softmax_activation_oracle(case):
    kw = dict(case["kwargs"])
    if kw.pop("class_weights", None):
        kw["class_weights"] = GL4[case["weights"]]
    x, t = GL4[case["inputs"][0]], GL4[case["inputs"][1]]
    fn = LO.binary_focal_loss if case["fn"] == "focal_softmax_module" else LO.focal_loss_with_logits
    out = fn(x, t, activation="softmax", **kw)
    np.testing.assert_allclose(out, GL4[case["output"]], rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------ multiscale: nearest mode, flips inside every scale
GT2 = load_golden("tta2.npz")


def _offs(kw):
    return [tuple(o) if isinstance(o, list) else o for o in kw["size_offsets"]]


@pytest.mark.parametrize("case", GT2.cases, ids=lambda c: c["name"])
def test_multiscale_modes_and_flip_composition_oracle(case):
    kw = case["kwargs"]
    offs = _offs(kw)
    if case["fn"] == "ms_image_augment_grad":
        outs = AO.ms_image_augment(GT2["x"], offs, kw["align_corners"], mode=kw["mode"])
        for i, o in enumerate(outs):
            np.testing.assert_allclose(o, GT2[f"{case['name']}_{i}"], rtol=1e-5, atol=1e-6)
    elif case["fn"] == "ms_image_deaugment_grad":
        out = AO.ms_image_deaugment([GT2[f"fm_{i}"] for i in range(len(offs))], offs, kw["reduction"], kw["align_corners"], mode=kw["mode"])
        np.testing.assert_allclose(out, GT2[case["name"]], rtol=1e-5, atol=1e-6)
    else:
        ys = [GT2[f"fz_{kw['group']}_y{i}"] for i in range(len(offs))]
        out = AO.ms_image_deaugment([AO.image_deaugment(y, kw["group"], kw["inner_reduction"]) for y in ys], offs, kw["reduction"], kw["align_corners"])
        np.testing.assert_allclose(out, GT2[case["name"]], rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------ stacks longer than 8, reductions with their eps argument
GT3 = load_golden("tta3.npz")


@pytest.mark.parametrize("case", GT3.cases, ids=lambda c: c["name"])
def test_long_stacks_and_eps_oracle(case):
    kw = case["kwargs"]
    x = GT3[case["inputs"][0]]
    if case["fn"] == "deaugment_averaging":
        out = AO.deaugment_averaging(x, kw["reduction"])
    else:
        out = getattr(AO, case["fn"])(x, axis=kw["dim"], eps=kw["eps"])
    np.testing.assert_allclose(out, GT3[case["output"]], rtol=1e-5, atol=1e-6)


GT6 = load_golden("tta6.npz")


@pytest.mark.parametrize("case", GT6.cases, ids=lambda c: c["name"])
def test_multiscale_area_and_nearest_exact_oracle(case):
    """oracle.tta_oracle.nearest_exact_resize / area_resize against the unmodified reference (F.interpolate's remaining 4-D modes)."""
    kw = case["kwargs"]
    offs = _offs(kw)
    if case["fn"] == "ms_image_augment_grad":
        for i, o in enumerate(AO.ms_image_augment(GT6["x"], offs, None, mode=kw["mode"])):
            np.testing.assert_allclose(o, GT6[f"{case['name']}_{i}"], rtol=1e-6, atol=1e-6)
    else:
        out = AO.ms_image_deaugment([GT6[f"fm_{i}"] for i in range(len(offs))], offs, kw["reduction"], None, mode=kw["mode"])
        np.testing.assert_allclose(out, GT6[case["name"]], rtol=1e-5, atol=2e-6)


GT4 = load_golden("tta4.npz")


@pytest.mark.parametrize("case", GT4.cases, ids=lambda c: c["name"])
def test_multiscale_bicubic_oracle(case):
    kw = case["kwargs"]
    offs = _offs(kw)
    if case["fn"] == "ms_image_augment_grad":
        for i, o in enumerate(AO.ms_image_augment(GT4["x"], offs, kw["align_corners"], mode="bicubic")):
            np.testing.assert_allclose(o, GT4[f"{case['name']}_{i}"], rtol=1e-5, atol=2e-6)
    else:
        out = AO.ms_image_deaugment([GT4[f"fm_{i}"] for i in range(len(offs))], offs, kw["reduction"], kw["align_corners"], mode="bicubic")
        np.testing.assert_allclose(out, GT4[case["name"]], rtol=1e-5, atol=2e-6)


GL5 = load_golden("losses5.npz")


@pytest.mark.parametrize("case", GL5.cases, ids=lambda c: c["name"])
def test_lovasz_oracle_against_larger_reference_cases(case):
    """losses5.npz: _lovasz_softmax with every `classes` form / per_image / ignore_index and _lovasz_hinge of the unmodified reference
    (losses/lovasz.py:37-49, :92-140) at 2 x 4 x 70 x 61 -- the numpy restatement reproduces the values."""
    a, b = GL5[case["inputs"][0]], GL5[case["inputs"][1]]
    kw = dict(case["kwargs"])
    if case["fn"] == "lovasz_softmax":
        out = LO.lovasz_softmax(a, b, classes=kw["classes"], per_image=kw["per_image"], ignore_index=kw["ignore_index"])
    else:
        out = LO.lovasz_hinge(a, b, per_image=kw["per_image"], ignore_index=kw["ignore_index"])
    np.testing.assert_allclose(out, GL5[case["output"]], rtol=1e-5, atol=1e-6)


GL6 = load_golden("losses6.npz")


@pytest.mark.parametrize("case", GL6.cases, ids=lambda c: c["name"])
def test_oracle_binary_focal_fractional_gamma_with_ignore_index(case):
    """losses6.npz: the oracle restates the reference faithfully here -- np.power of the negative base of an ignored entry is NaN and
    np.where masks it, exactly like functional.py:70, 90-94 -- so the VALUE matches; the fixture also records how many gradient entries
    of the reference are NaN (the corner the library deviates on, see tests/test_losses2_gpu.py)."""
    with np.errstate(invalid="ignore"):
        got = LO.binary_focal_loss(GL6[case["inputs"][0]], GL6[case["inputs"][1]], **case["kwargs"])
    np.testing.assert_allclose(got, GL6[case["output"]], rtol=1e-5, atol=1e-6)
    assert int(np.isnan(GL6[case["output"] + "_grad"]).sum()) == case["nan_grads"]
    assert (case["nan_grads"] > 0) == (float(case["kwargs"]["gamma"]) != int(case["kwargs"]["gamma"]))
