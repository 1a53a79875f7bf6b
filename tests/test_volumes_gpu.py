"""GPU parity: ``VolumeMerger`` (HIP, through the C ABI) vs the reference's golden vectors and the numpy oracle.
Bit-exact: same sequential fp32 multiply-then-add order as the reference's loop."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import volumes_oracle as VO

pytestmark = pytest.mark.gpu

GV = load_golden("volumes.npz")


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


@pytest.mark.parametrize("scalar", [0, 1])
@pytest.mark.parametrize("case", GV.by_fn("vmerger"), ids=lambda c: c["name"])
def test_golden_volume_merger_bit_exact(case, scalar, dev, native):
    from pytorch_toolbelt_amd.inference.tiles_3d import VolumeMerger, VolumeSlicer

    native.load().ptb_set_tunable(1, scalar)
    kw, n = case["kwargs"], case["name"]
    s = VolumeSlicer(kw["volume_shape"], kw["voxel_size"], kw["voxel_step"])
    m = VolumeMerger(s.target_shape, kw["channels"], GV[f"{n}_weight"], device=dev)
    pred = torch.from_numpy(GV[f"{n}_pred"]).to(dev)
    before = native.calls
    for b0 in range(0, len(pred), kw["batch"]):
        m.integrate_batch(pred[b0:b0 + kw["batch"]], s.crops[b0:b0 + kw["batch"]])
    assert native.calls > before
    assert np.array_equal(m.volume.cpu().numpy(), GV[f"{n}_volume"])
    assert np.array_equal(m.norm_mask.cpu().numpy(), GV[f"{n}_norm"])
    assert np.array_equal(m.merge().cpu().numpy(), GV[f"{n}_merged"])


def test_roundtrip_errors_and_single(dev):
    from pytorch_toolbelt_amd.inference.tiles_3d import VolumeMerger, VolumeSlicer

    rng = np.random.default_rng(1)
    vol = rng.standard_normal((40, 50, 36)).astype(np.float32)
    s = VolumeSlicer(vol.shape, (16, 16, 16), (8, 8, 8))
    m = VolumeMerger(s.target_shape, 1, s.weight, device=dev)
    tiles = torch.from_numpy(np.stack(s.split(vol))[:, None]).to(dev)
    for b0 in range(0, len(tiles), 16):
        m.integrate_batch(tiles[b0:b0 + 16], s.crops[b0:b0 + 16])
    merged = m.merge()[0].cpu().numpy()
    assert np.allclose(s.crop_to_orignal_size(merged), vol, atol=1e-6)
    # accumulate_single (non-functional in the reference: implemented to its evident intent) == a batch of one
    a = VolumeMerger(s.target_shape, 1, s.weight, device=dev)
    b = VolumeMerger(s.target_shape, 1, s.weight, device=dev)
    a.accumulate_single(tiles[3], s.crops[3])
    b.integrate_batch(tiles[3:4], s.crops[3:4])
    assert torch.equal(a.volume, b.volume) and torch.equal(a.norm_mask, b.norm_mask)
    with pytest.raises(ValueError):
        m.integrate_batch(tiles[:2], s.crops[:3])
    with pytest.raises(RuntimeError):
        m.integrate_batch(tiles[:1, :, :8], s.crops[:1])
    with pytest.raises(RuntimeError):
        m.integrate_batch(tiles[:1], [(slice(0, 16), slice(0, 16), slice(40, 56))])       # outside the volume
    # the reference's default device="cpu": the torch-op merger on the host, which agrees with the HIP one
    host = VolumeMerger(s.target_shape, 1, s.weight)
    assert not host.volume.is_cuda and isinstance(host, VolumeMerger)
    # an oracle cross-check on a geometry with unaligned x origins (scalar path inside one merger)
    s2 = VolumeSlicer((13, 14, 15), (6, 7, 5), (3, 4, 5))
    w2 = (rng.random((6, 7, 5)) + 0.1).astype(np.float32)
    p2 = rng.standard_normal((len(s2.crops), 2, 6, 7, 5)).astype(np.float32)
    m2 = VolumeMerger(s2.target_shape, 2, w2, device=dev)
    m2.integrate_batch(torch.from_numpy(p2).to(dev), s2.crops)
    g = VO.slicer_geometry((13, 14, 15), (6, 7, 5), (3, 4, 5))
    st = VO.merger_integrate(VO.merger_new(g["target_shape"], 2, w2), p2, g["starts"])
    assert np.array_equal(m2.volume.cpu().numpy(), st["volume"]) and np.array_equal(m2.merge().cpu().numpy(), VO.merger_merge(st))
