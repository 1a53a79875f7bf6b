"""CPU: 3-D tiles -- the numpy oracle and this package's host-side ``VolumeSlicer`` against the golden
vectors of the reference's ``VolumeSlicer`` / ``VolumeMerger`` (tests/golden/volumes.npz).  Integer work: bit-exact."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import volumes_oracle as VO

GV = load_golden("volumes.npz")


def _starts(slicer, attr="crops"):
    return np.array([[r.start for r in roi] for roi in getattr(slicer, attr)], dtype=np.int64)


@pytest.mark.parametrize("case", GV.by_fn("vgeometry"), ids=lambda c: c["name"])
def test_geometry_bit_exact(case):
    from pytorch_toolbelt_amd.inference.tiles_3d import VolumeSlicer

    kw, n = case["kwargs"], case["name"]
    g = VO.slicer_geometry(kw["volume_shape"], kw["voxel_size"], kw["voxel_step"])
    s = VolumeSlicer(kw["volume_shape"], kw["voxel_size"], kw["voxel_step"])
    for starts, bbox, meta in ((g["starts"], g["bbox_starts"], [*g["pad_before"], *g["pad_after"], *g["target_shape"]]),
                               (_starts(s), _starts(s, "bbox_crops"), [*s.pad_before, *s.pad_after, *s.target_shape])):
        assert np.array_equal(starts, GV[f"{n}_starts"])
        assert np.array_equal(bbox, GV[f"{n}_bbox_starts"])
        assert [int(v) for v in meta] == GV[f"{n}_meta"][:9].tolist()
    assert np.array_equal(np.array([[r.stop for r in roi] for roi in s.crops]), GV[f"{n}_stops"])
    assert [int(v) for v in s.num_tiles] == GV[f"{n}_meta"][9:].tolist()


@pytest.mark.parametrize("case", GV.by_fn("vsplit"), ids=lambda c: c["name"])
def test_split_bit_exact(case):
    from pytorch_toolbelt_amd.inference.tiles_3d import VolumeSlicer

    kw, n = case["kwargs"], case["name"]
    vol = GV[f"{n}_volume"]
    g = VO.slicer_geometry(vol.shape, kw["voxel_size"], kw["voxel_step"])
    assert np.array_equal(np.stack(VO.split(vol, g, value=5)), GV[f"{n}_tiles"])
    s = VolumeSlicer(vol.shape, kw["voxel_size"], kw["voxel_step"])
    assert np.array_equal(np.stack(s.split(vol, value=5)), GV[f"{n}_tiles"])
    it = list(s.iter_split(vol, value=5))
    assert np.array_equal(np.stack([t for t, _ in it]), GV[f"{n}_tiles"]) and [r for _, r in it] == s.crops
    padded = np.pad(vol, [(int(b), int(a)) for b, a in zip(s.pad_before, s.pad_after)])
    assert np.array_equal(s.crop_to_orignal_size(padded), GV[f"{n}_crop"]) and np.array_equal(GV[f"{n}_crop"], vol)
    assert np.array_equal(VO.crop_to_original(padded, g, vol.shape), vol)
    with pytest.raises(ValueError):
        s.split(vol[1:])


@pytest.mark.parametrize("case", GV.by_fn("vmerger"), ids=lambda c: c["name"])
def test_merger_oracle_bit_exact(case):
    kw, n = case["kwargs"], case["name"]
    g = VO.slicer_geometry(kw["volume_shape"], kw["voxel_size"], kw["voxel_step"])
    st = VO.merger_new(g["target_shape"], kw["channels"], GV[f"{n}_weight"])
    pred = GV[f"{n}_pred"]
    for b0 in range(0, len(pred), kw["batch"]):
        VO.merger_integrate(st, pred[b0:b0 + kw["batch"]], g["starts"][b0:b0 + kw["batch"]])
    assert np.array_equal(st["volume"], GV[f"{n}_volume"]) and np.array_equal(st["norm_mask"], GV[f"{n}_norm"])
    assert np.array_equal(VO.merger_merge(st), GV[f"{n}_merged"])


def test_slicer_validation_weight_and_host_merge():
    from pytorch_toolbelt_amd.inference.tiles_3d import VolumeSlicer

    with pytest.raises(ValueError):
        VolumeSlicer((8, 8, 8), (4, 4), 2)
    with pytest.raises(ValueError):
        VolumeSlicer((8, 8, 8), 4, (2, 2, 5))
    with pytest.raises(ValueError):
        VolumeSlicer((8, 8, 8), 4, 0)
    s = VolumeSlicer((10, 11, 12), (4, 6, 8), (2, 3, 4))
    assert s.weight.shape == (4, 6, 8) and s.weight.dtype == np.float32 and (s.weight == 1).all()
    rng = np.random.default_rng(0)
    vol = rng.standard_normal((10, 11, 12))
    assert np.allclose(s.merge(s.split(vol), dtype=np.float64), vol)          # blending identical overlaps gives the volume back
    vol4 = rng.standard_normal((10, 11, 12, 2))
    assert np.allclose(s.merge(s.split(vol4), dtype=np.float64), vol4)
    with pytest.raises(ValueError):
        s.merge(s.split(vol)[:-1])
