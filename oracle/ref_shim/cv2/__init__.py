"""TEST-HARNESS ONLY: minimal stand-in for OpenCV so the *unmodified* reference
(/root/reference) can be imported in a container that has no cv2 wheel.

Only the zero-padding used by pytorch_toolbelt/inference/tiles.py:161,182,220
(cv2.copyMakeBorder with BORDER_CONSTANT) carries arithmetic; it is np.pad.
Everything else is just the integer constants the reference's import chain reads.
Never shipped in the product package.
"""
import numpy as np

BORDER_CONSTANT = 0
BORDER_REPLICATE = 1
BORDER_REFLECT = 2
BORDER_WRAP = 3
BORDER_REFLECT_101 = 4
BORDER_DEFAULT = 4
IMREAD_COLOR = 1
IMREAD_GRAYSCALE = 0
IMREAD_UNCHANGED = -1
IMREAD_ANYCOLOR = 4
IMREAD_ANYDEPTH = 2
COLOR_BGR2RGB = 4
COLOR_RGB2BGR = 4
COLOR_GRAY2RGB = 8
COLOR_BGR2GRAY = 6
FONT_HERSHEY_PLAIN = 1
FONT_HERSHEY_SIMPLEX = 0
LINE_AA = 16
THRESH_BINARY = 0
INTER_LINEAR = 1
INTER_NEAREST = 0
INTER_CUBIC = 2
INTER_AREA = 3
INTER_LANCZOS4 = 4


def copyMakeBorder(src, top, bottom, left, right, borderType=BORDER_CONSTANT, dst=None, value=0):
    assert borderType == BORDER_CONSTANT, "shim only restates constant (zero/value) padding"
    pad = [(int(top), int(bottom)), (int(left), int(right))] + [(0, 0)] * (src.ndim - 2)
    return np.pad(src, pad, mode="constant", constant_values=value)


def __getattr__(name):  # any other cv2 symbol: fail at call time, not import time
    def _missing(*a, **k):
        raise NotImplementedError(f"cv2.{name} is not available in the reference-import shim")
    return _missing
