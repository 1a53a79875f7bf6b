"""CPU: ImageSlicer host logic of the product vs the golden vectors of the unmodified reference (bit-exact)."""
import hashlib

import numpy as np
import pytest

from conftest import load_golden
from pytorch_toolbelt_amd.inference.tiles import ImageSlicer, compute_pyramid_patch_weight_loss

GT = load_golden("tiles.npz")


@pytest.mark.parametrize("case", GT.by_fn("geometry"), ids=lambda c: c["name"])
def test_geometry(case):
    kw, n = case["kwargs"], case["name"]
    s = ImageSlicer(kw["image_shape"], kw["tile_size"], kw["tile_step"], image_margin=kw.get("image_margin", 0))
    assert s.crops.dtype == np.int64 and np.array_equal(s.crops, GT[f"{n}_crops"])
    assert np.array_equal(s.bbox_crops, GT[f"{n}_bbox"])
    meta = [s.margin_left, s.margin_right, s.margin_top, s.margin_bottom, *s.target_shape, *s.tile_size, *s.tile_step]
    assert meta == GT[f"{n}_meta"].tolist()


@pytest.mark.parametrize("case", GT.by_fn("pyramid"), ids=lambda c: c["name"])
def test_pyramid(case):
    n = case["name"]
    W, Dc, De = compute_pyramid_patch_weight_loss(**case["kwargs"])
    assert np.array_equal(W, GT[f"{n}_W"]) and np.array_equal(Dc, GT[f"{n}_Dc"]) and np.array_equal(De, GT[f"{n}_De"])


@pytest.mark.parametrize("case", GT.by_fn("pyramid_digest"), ids=lambda c: c["name"])
def test_pyramid_digest(case):
    W, _, _ = compute_pyramid_patch_weight_loss(**case["kwargs"])
    assert hashlib.sha256(np.ascontiguousarray(W).tobytes()).hexdigest() == str(GT[f"{case['name']}_sha256"])


@pytest.mark.parametrize("case", GT.by_fn("split_merge"), ids=lambda c: c["name"])
def test_split_cut_merge(case):
    kw, n = case["kwargs"], case["name"]
    img = GT[f"{n}_image"]
    s = ImageSlicer(img.shape, kw["tile_size"], kw["tile_step"], weight=kw["weight"])
    tiles = s.split(img)
    assert np.array_equal(np.stack(tiles), GT[f"{n}_tiles"])
    assert all(t.base is not None for t in tiles)  # views of the padded copy, no per-tile allocation
    assert np.array_equal(np.stack([s.cut_patch(img, i) for i in range(len(tiles))]), GT[f"{n}_cut"])
    it = list(s.iter_split(img))
    assert np.array_equal(np.stack([t for t, _ in it]), GT[f"{n}_iter_tiles"])
    assert np.array_equal(np.stack([c for _, c in it]), GT[f"{n}_iter_coords"])
    assert np.array_equal(s.merge(tiles, dtype=np.float32), GT[f"{n}_merge_f32"])
    assert np.array_equal(s.merge(tiles, dtype=np.uint8), GT[f"{n}_merge_u8"])
    assert np.array_equal(s.merge(list(GT[f"{n}_ftiles"]), dtype=np.float32), GT[f"{n}_fmerge"])


def test_reference_tests_tiles():
    """reference tests/test_tiles.py:13-26,47-55 (their images are all-zero by construction)."""
    for shape, ts, st, w in [((500, 500, 3), 51, 26, "mean"), ((563, 512, 3), (128, 128), (128, 128), "mean"), ((1000, 1000, 3), (512, 512), (256, 256), "pyramid")]:
        image = np.random.random(shape).astype(np.uint8)
        tiler = ImageSlicer(image.shape, tile_size=ts, tile_step=st, weight=w)
        np.testing.assert_allclose(tiler.weight, tiler.weight.T)
        np.testing.assert_equal(tiler.merge(tiler.split(image), dtype=np.uint8), image)


def test_errors():
    with pytest.raises(ValueError):
        ImageSlicer((10, 10), 4)  # default tile_step=0
    with pytest.raises(ValueError):
        ImageSlicer((10, 10), 4, 5)
    with pytest.raises(ValueError):
        ImageSlicer((10, 10), (4, 4, 4), 2)
    with pytest.raises(KeyError):
        ImageSlicer((10, 10), 4, 2, weight="gauss")
    s = ImageSlicer((10, 10, 1), 4, 2)
    with pytest.raises(ValueError):
        list(s.iter_split(np.zeros((9, 10, 1))))
    with pytest.raises(AssertionError):
        s.split(np.zeros((9, 10, 1)))
    with pytest.raises(ValueError):
        s.merge([np.zeros((4, 4, 1))])
    w = np.full((4, 4), 2.0)
    assert ImageSlicer((10, 10), 4, 2, weight=w).weight is w


def test_pickle_roundtrip():
    import pickle

    s = ImageSlicer((100, 90, 3), 32, 16, weight="pyramid")
    s2 = pickle.loads(pickle.dumps(s))
    assert np.array_equal(s.crops, s2.crops) and np.array_equal(s.weight, s2.weight)


# ------------------------------------------------------------------ non-constant borders: OpenCV's documented tables
_CV_BORDER_TABLES = {        # cv2.BorderTypes documentation: "abcdefgh" extended by six on the left and seven on the right
    1: ("aaaaaa", "hhhhhhh"),        # BORDER_REPLICATE    aaaaaa|abcdefgh|hhhhhhh
    2: ("fedcba", "hgfedcb"),        # BORDER_REFLECT      fedcba|abcdefgh|hgfedcb
    3: ("cdefgh", "abcdefg"),        # BORDER_WRAP         cdefgh|abcdefgh|abcdefg
    4: ("gfedcb", "gfedcba"),        # BORDER_REFLECT_101  gfedcb|abcdefgh|gfedcba
}


@pytest.mark.parametrize("code", sorted(_CV_BORDER_TABLES))
def test_border_types_follow_opencv_documented_tables(code):
    """`border_type` is forwarded to cv2.copyMakeBorder by the reference (tiles.py:161,182,220); cv2 is not available where the goldens
    are made, so the non-constant codes are pinned to the extension tables of OpenCV's documentation -- along both axes, through
    split(), iter_split() and cut_patch()."""
    from pytorch_toolbelt_amd.inference.tiles import ImageSlicer

    left, right = _CV_BORDER_TABLES[code]
    row = np.frombuffer(b"abcdefgh", dtype=np.uint8)
    want = np.frombuffer((left + "abcdefgh" + right).encode(), dtype=np.uint8)
    # horizontal: an 1 x 8 image with margins (6, 7, 0, 0); vertical: its transpose
    s = ImageSlicer((1, 8), tile_size=(1, 21), tile_step=(1, 21), image_margin=(6, 7, 0, 0))
    (tile,) = s.split(row[None, :], border_type=code)
    assert tile.shape == (1, 21) and bytes(tile[0]) == bytes(want)
    assert bytes(s.cut_patch(row[None, :], 0, border_type=code)[0]) == bytes(want)
    (lazy_tile, _xy), = list(s.iter_split(row[None, :], border_type=code))
    assert bytes(lazy_tile[0]) == bytes(want)
    sv = ImageSlicer((8, 1), tile_size=(21, 1), tile_step=(21, 1), image_margin=(0, 0, 6, 7))
    (vt,) = sv.split(row[:, None], border_type=code)
    assert vt.shape == (21, 1) and bytes(vt[:, 0]) == bytes(want)
    # channels ride along untouched
    rgb = np.stack([row, row[::-1], row], axis=-1)[None]
    (t3,) = s.split(rgb, border_type=code)
    assert bytes(t3[0, :, 0]) == bytes(want) and bytes(t3[0, :, 2]) == bytes(want)
    with pytest.raises(NotImplementedError):
        s.split(row[None, :], border_type=7)
