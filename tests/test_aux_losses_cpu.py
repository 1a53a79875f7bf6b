"""CPU: the classification-shaped losses that are plain torch tensor algebra in this package (bi-tempered, soft F1,
focal cosine -- device agnostic) against values AND gradients produced by the unmodified reference
(tests/golden/losses3.npz, generator oracle/make_golden.py:gen_losses3)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

G3 = load_golden("losses3.npz")


def build(fn, kw):
    from pytorch_toolbelt_amd import losses as L

    return {"bitempered": L.BiTemperedLogisticLoss, "binary_bitempered": L.BinaryBiTemperedLogisticLoss,
            "binary_soft_f1": L.BinarySoftF1Loss, "soft_f1": L.SoftF1Loss, "focal_cosine": L.FocalCosineLoss}[fn](**kw)


@pytest.mark.parametrize("case", G3.cases, ids=lambda c: c["name"])
def test_matches_reference_values_and_gradients(case):
    x = torch.from_numpy(G3[case["inputs"][0]]).clone().requires_grad_(True)
    t = torch.from_numpy(G3[case["inputs"][1]])
    val = build(case["fn"], case["kwargs"])(x, t)
    np.testing.assert_allclose(val.detach().numpy(), G3[case["name"]], rtol=1e-5, atol=1e-6)
    val.sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), G3[case["name"] + "_grad"], rtol=1e-5, atol=1e-6)


def test_binary_bitempered_rejects_multichannel():
    from pytorch_toolbelt_amd.losses import BinaryBiTemperedLogisticLoss

    with pytest.raises(ValueError):
        BinaryBiTemperedLogisticLoss(0.8, 1.2)(torch.zeros(2, 2, 3, 3), torch.zeros(2, 2, 3, 3))
