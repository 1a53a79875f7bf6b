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


def test_forced_rebuild_of_every_translation_unit(forced_build):
    """build(force=True): every .hip translation unit is really recompiled for gfx950 and linked (the content-hashed
    build() otherwise reuses the shipped library, so a compile error would go unnoticed) -- into a scratch directory, so the library
    this process has loaded is not overwritten.  The fresh library exports every symbol of include/ptb_hip.h."""
    import ctypes
    import glob

    import __graft_entry__ as g

    out = forced_build["lib_dir"]
    objs = glob.glob(os.path.join(out, "obj", "*.o"))
    assert len(objs) == len(g._sources()) >= 12
    fresh = ctypes.CDLL(os.path.join(out, "libptb_hip.so"))
    for n in _header_symbols():
        assert hasattr(fresh, n), f"{n} missing from the freshly built library"
    fresh.ptb_version.restype = ctypes.c_int
    assert fresh.ptb_version() >= 100
    shipped = os.path.getsize(g.LIB)
    assert abs(os.path.getsize(os.path.join(out, "libptb_hip.so")) - shipped) <= 0.02 * shipped, "the shipped library is not what the sources build"


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


def test_device_decides_and_the_hip_path_never_falls_back():
    """The contract of the two implementations: the DEVICE the caller names decides.  CPU tensors / device="cpu" take the host
    (torch-op) path like the reference does; anything CUDA takes the HIP kernels and fails LOUDLY when they cannot run -- a missing
    libptb_hip.so is an ImportError, a CUDA device on a box without a GPU is an error, and neither is ever answered by the host path."""
    import subprocess
    import sys

    import numpy as np
    import torch

    from pytorch_toolbelt_amd.inference import tta
    from pytorch_toolbelt_amd.inference.tiles import HostBackedTileMerger, ImageSlicer, TileMerger

    m = TileMerger((64, 64), 1, np.ones((32, 32), np.float32))                  # the reference's default device
    assert type(m) is HostBackedTileMerger and m.device.type == "cpu" and not m.image.is_cuda
    x = torch.rand(1, 1, 8, 8)
    assert torch.allclose(tta.d4_image_deaugment(tta.d4_image_augment(x), reduction="sum"), 8 * x)
    if not torch.cuda.is_available():
        with pytest.raises(Exception):                                          # no GPU here: a CUDA merger cannot be built, and is not faked
            TileMerger((64, 64), 1, np.ones((32, 32), np.float32), device="cuda")
    # entry points without a host form keep refusing host tensors
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ImageSlicer((64, 48, 3), (32, 16), (16, 16)).split_device(torch.zeros((64, 48, 3), dtype=torch.uint8))
    # a missing extension is an ImportError at the first native call -- checked in a fresh interpreter
    code = ("import os, sys; sys.path.insert(0, %r); os.environ['PTB_HIP_LIB'] = '/nonexistent/libptb_hip.so'\n"
            "from pytorch_toolbelt_amd import _native\n"
            "try:\n    _native.load()\nexcept ImportError as e:\n    print('IMPORT-ERROR', 'no non-HIP fallback' in str(e).replace('There is no', 'no'))\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "IMPORT-ERROR" in out.stdout, out.stdout + out.stderr
    # the host modules are torch-only: they import neither the native binding nor the oracle
    for rel in ("inference/_host.py", "losses/_host.py"):
        src = open(os.path.join(ROOT, "pytorch_toolbelt_amd", rel)).read()
        assert not re.search(r"^\s*(from|import)\s+(oracle|ctypes)\b", src, flags=re.M) and "_native" not in src, rel


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "pytorch_toolbelt_amd")
    for dirpath, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"
                assert "/root/reference" not in src
