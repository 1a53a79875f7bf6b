"""CPU: the C-ABI library builds for gfx950, loads, and exports exactly the symbols include/ptb_hip.h declares."""
import os
import re

import pytest

from conftest import ROOT


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "ptb_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ptb_[a-z0-9_]+)\s*\(", text)))


def test_build_and_symbols():
    import __graft_entry__ as g

    g.build()
    from pytorch_toolbelt_amd import _native

    lib = _native.load()
    names = _header_symbols()
    assert len(names) >= 8
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ptb_hip.h but not exported"
    assert sorted(_native.SIGNATURES) == names, "ctypes SIGNATURES and include/ptb_hip.h disagree"
    assert lib.ptb_version() >= 100


def test_argument_validation_without_gpu():
    """Entry points validate arguments before touching the device, so these calls are safe without a GPU."""
    from pytorch_toolbelt_amd import _native as N

    lib = N.load()
    assert lib.ptb_set_tunable(0, 48) == -1
    assert lib.ptb_set_tunable(0, 64) == 0
    assert lib.ptb_set_tunable(7, 1) == -1
    assert lib.ptb_merge_div(None, None, None, 1, 16, None) == -1
    assert lib.ptb_deaug_reduce(None, None, 8, N.int_array([0] * 8), 1, 1, 1, 8, 8, None) == -1
    assert lib.ptb_resize_bilinear(None, None, 1, 4, 4, 8, 8, 0, None) == -1


def test_no_cpu_fallback():
    """The product refuses CPU tensors instead of computing them somewhere else."""
    import numpy as np
    import torch

    from pytorch_toolbelt_amd.inference import tta
    from pytorch_toolbelt_amd.inference.tiles import TileMerger

    with pytest.raises(RuntimeError, match="no CPU"):
        TileMerger((64, 64), 1, np.ones((32, 32), np.float32))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        tta.d4_image_augment(torch.rand(1, 1, 8, 8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        tta.fliplr_image_deaugment(torch.rand(2, 1, 8, 8))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "pytorch_toolbelt_amd")
    for dirpath, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"
                assert "/root/reference" not in src
