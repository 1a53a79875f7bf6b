"""CPU: the torch-CPU op chains `bench.py` times as `cpu_baseline` (oracle/torch_chain.py) compute what the pinned numpy oracle
computes -- the timed baseline is known to be the right math (the headline chain is checked in tests/test_fullsize_cpu.py)."""
import numpy as np
import torch

from oracle import losses_oracle as LO
from oracle import torch_chain as TC
from oracle import tta_oracle as AO


def test_loss_chains_match_the_oracle():
    g = torch.Generator().manual_seed(0)
    x = torch.randn((3, 6, 24, 20), generator=g) * 2
    lab = torch.randint(0, 6, (3, 24, 20), generator=g)
    lab[lab == 4] = 1                                  # an absent class: masked out of Dice / Jaccard, skipped by Lovasz "present"
    focal, dice, jaccard = TC.binary_focal_multiclass_dice_jaccard(x, lab)
    xn, ln = x.numpy(), lab.numpy()
    assert abs(float(focal) - float(LO.binary_focal_loss(xn, ln))) <= 1e-6
    assert abs(float(dice) - float(LO.dice_loss(xn, ln, "multiclass"))) <= 1e-6
    assert abs(float(jaccard) - float(LO.jaccard_loss(xn, ln, "multiclass"))) <= 1e-6
    p = x.softmax(1)
    assert abs(float(TC.lovasz_softmax(p, lab)) - float(LO.lovasz_softmax(p.numpy(), ln))) <= 1e-5


def test_multiscale_fliplr_chain_matches_the_oracle():
    g = torch.Generator().manual_seed(1)
    offs = [-8, 0, 8]
    ys = [torch.rand((2, 3, 32 + o, 32 + o), generator=g) * 0.9 + 0.05 for o in offs]
    for ac in (False, True):
        got = TC.ms_fliplr_deaugment(ys, offs, "gmean", align_corners=ac).numpy()
        want = AO.ms_image_deaugment([AO.image_deaugment(y.numpy(), "fliplr", "gmean") for y in ys], offs, "gmean", ac)
        assert np.abs(got - want).max() <= 1e-5
