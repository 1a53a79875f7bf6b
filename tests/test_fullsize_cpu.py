"""CPU: the numpy oracle at the FULL size of every BASELINE.json config against digests of the unmodified reference's outputs
(tests/golden/fullsize.npz, written by `python oracle/make_golden.py fullsize`).  The inputs are regenerated from
oracle/synth.py (an integer hash numpy and torch evaluate bit-identically).  Together with tests/test_fullsize_gpu.py
(HIP path vs the same digests and vs this oracle on every pixel) this pins the benchmarked configurations at their
stated sizes."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import losses_oracle as LO
from oracle import synth as SY
from oracle import tiles_oracle as TO
from oracle import tta_oracle as AO


@pytest.fixture(scope="module")
def full():
    return load_golden("fullsize.npz")


def _synth(shape, seed, kind="sym"):
    """synth_np's values through torch's multi-threaded CPU kernels (bit-identical, several times faster on 10^8 elements)."""
    return SY.synth_torch(shape, seed, kind, device="cpu").numpy()


def _check_digest(full, key, arr, tol):
    sh, sw = (int(v) for v in full[f"{key.split('_')[0]}_meta"][:2])
    sub, sums = SY.digest(arr, sh, sw)
    np.testing.assert_allclose(sub, full[f"{key}_sub"], rtol=0, atol=tol)
    want = full[f"{key}_sums"]
    assert abs(sums[0] - want[0]) <= tol * arr.size * 1e-2 + 1e-6 * abs(want[1])
    assert abs(sums[1] - want[1]) <= 1e-6 * abs(want[1])


def test_synth_numpy_equals_torch():
    for kind in ("sym", "unit"):
        a = SY.synth_np((3, 5, 64, 33), 77, kind)
        assert a.dtype == np.float32 and np.array_equal(a, SY.synth_torch((3, 5, 64, 33), 77, kind, device="cpu").numpy())
    assert SY.synth_np((4096,), 1, "unit").min() > 0
    lab = SY.labels_np((7, 9, 11), 5, 16)
    assert lab.dtype == np.int64 and np.array_equal(lab, SY.labels_torch((7, 9, 11), 5, 16, device="cpu").numpy())


def test_cfg1_oracle_vs_reference_digest(full):
    """BASELINE configs[0]: 1024x1024x3, 256/128, pyramid, CPU TileMerger and the fp64 host merge (bit-exact / 1e-6)."""
    geom = TO.slicer_geometry((1024, 1024, 3), 256, 128)
    weight = TO.pyramid_window(256, 256)[0]
    pred = SY.synth_np((49, 3, 256, 256), 101)
    st = TO.merger_new(geom["target_shape"], 3, weight)
    for b0 in range(0, 49, 8):
        TO.merger_integrate(st, pred[b0:b0 + 8], geom["crops"][b0:b0 + 8])
    merged = TO.merger_merge(st)
    sub, _ = SY.digest(merged, 13, 17)
    assert np.array_equal(sub, full["cfg1_sub"])          # same fp32 operation order -> same bits
    host = TO.slicer_merge([np.moveaxis(p, 0, -1) for p in pred], geom, weight, (1024, 1024, 3))
    _check_digest(full, "cfg1_host", np.moveaxis(host, -1, 0), 1e-6)
    assert np.abs(np.moveaxis(host, -1, 0) - merged).max() <= 1e-5


def test_cfg2_oracle_vs_reference_digest(full):
    """BASELINE configs[1]: 5000x5000x3, 512/256, 361 tiles, d4 mean de-augment + integrate in batches of 8, C = 4."""
    geom = TO.slicer_geometry((5000, 5000, 3), 512, 256)
    weight = TO.pyramid_window(512, 512)[0]
    crops = geom["crops"]
    assert len(crops) == 361
    st = TO.merger_new(geom["target_shape"], 4, weight)
    for k, b0 in enumerate(range(0, 361, 8)):
        nb = min(8, 361 - b0)
        TO.merger_integrate(st, AO.image_deaugment(_synth((8 * nb, 4, 512, 512), 2000 + k), "d4", "mean"), crops[b0:b0 + nb])
    merged = TO.merger_merge(st)
    _check_digest(full, "cfg2", merged, 1e-6)
    np.testing.assert_allclose(merged[:, [0, 255, 256, 2559, 2560, 5119], :], full["cfg2_rows"], rtol=0, atol=1e-6)


def test_cfg4_oracle_vs_reference_values(full):
    """BASELINE configs[3]: [32,16,512,512] logits + int64 labels: focal / Dice / Jaccard / CE-focal scalars (fp64 oracle)."""
    B, C, H, W = 32, 16, 512, 512
    x = _synth((B, C, H, W), 4001) * np.float32(2.0)
    lab = SY.labels_torch((B, H, W), 4002, C, device="cpu").numpy()
    # one quarter of the batch at a time keeps the fp64 temporaries small; the sums are exact enough in fp64
    assert float(LO.dice_loss(x, lab, "multiclass")) == pytest.approx(float(full["cfg4_dice"]), abs=1e-5)
    assert float(LO.jaccard_loss(x, lab, "multiclass")) == pytest.approx(float(full["cfg4_jaccard"]), abs=1e-5)
    parts = [float(LO.binary_focal_loss(x[i:i + 4], lab[i:i + 4], reduction="sum")) for i in range(0, B, 4)]
    assert sum(parts) / x.size == pytest.approx(float(full["cfg4_focal"]), abs=1e-6)
    parts = [float(LO.binary_focal_loss(x[i:i + 4], lab[i:i + 4], alpha=0.25, reduction="sum")) for i in range(0, B, 4)]
    assert sum(parts) / x.size == pytest.approx(float(full["cfg4_focal_alpha"]), abs=1e-6)


def test_cfg5_oracle_vs_reference_digest(full):
    """BASELINE configs[4]: scales 0.75 / 1.0 / 1.25 of 4096x4096, fliplr TTA inside every scale, gmean merge, C = 4."""
    offs = [-1024, 0, 1024]
    per_scale = [AO.image_deaugment(_synth((2, 4, 4096 + o, 4096 + o), 5000 + i, "unit"), "fliplr", "gmean") for i, o in enumerate(offs)]
    for ac in (False, True):
        out = AO.ms_image_deaugment(per_scale, offs, "gmean", ac)
        _check_digest(full, f"cfg5_ac{int(ac)}", out, 2e-6)


def test_torch_cpu_chain_equals_the_numpy_oracle():
    """oracle/torch_chain.py (what bench.py's cpu_baseline leg times on all host cores) computes the oracle's answer."""
    from oracle import torch_chain as TC

    geom = TO.slicer_geometry((300, 260, 3), 64, 32)
    weight = TO.pyramid_window(64, 64)[0]
    crops, C = geom["crops"], 3
    m = TC.Merger(geom["target_shape"], C, weight)
    st = TO.merger_new(geom["target_shape"], C, weight)
    for k, b0 in enumerate(range(0, len(crops), 5)):
        nb = min(5, len(crops) - b0)
        y = SY.synth_np((8 * nb, C, 64, 64), 300 + k)
        red = TC.image_deaugment(torch.from_numpy(y), "d4", "mean")
        want = AO.image_deaugment(y, "d4", "mean")
        np.testing.assert_allclose(red.numpy(), want, rtol=0, atol=1e-6)
        m.integrate_batch(red, crops[b0:b0 + nb])
        TO.merger_integrate(st, want, crops[b0:b0 + nb])
    np.testing.assert_allclose(m.merge().numpy(), TO.merger_merge(st), rtol=0, atol=1e-5)
    g = SY.synth_np((4, 2, 16, 24), 9, "unit")
    np.testing.assert_allclose(TC.image_deaugment(torch.from_numpy(g), "fliplr", "gmean").numpy(), AO.image_deaugment(g, "fliplr", "gmean"), atol=1e-6)
    cores, logical, model = TC.host_description()
    assert 1 <= cores <= logical and isinstance(model, str)
