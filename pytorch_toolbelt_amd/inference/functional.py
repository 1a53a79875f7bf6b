"""Geometric view ops and TTA reductions (drop-in for ``pytorch_toolbelt.inference.functional``).

The eight ``torch_*`` image transforms are the dihedral group D4 acting on dims (2, 3) of a ``[B, C, H, W]`` tensor
(reference inference/functional.py:47-145); here each one is a single HIP gather kernel that returns a *contiguous*
tensor (the reference returns strided views for rot90/transpose -- values are identical).  The mean variants
(reference inference/functional.py:250-333) reduce with a fused pre-transform / sum / post-transform HIP kernel.
Pad / unpad helpers are host-side shape arithmetic around ``torch.nn.functional.pad`` and are device-agnostic.
"""
import itertools
from collections.abc import Iterable, Sized
from typing import Tuple, Union

import torch
from torch import Tensor

from .. import _native as N
from ..utils.support import pytorch_toolbelt_deprecated
from . import _views as V

__all__ = [
    "geometric_mean",
    "harmonic_mean",
    "harmonic1p_mean",
    "logodd_mean",
    "log1p_mean",
    "pad_image_tensor",
    "pad_tensor_to_size",
    "torch_fliplr",
    "torch_flipud",
    "torch_none",
    "torch_rot180",
    "torch_rot270",
    "torch_rot90",
    "torch_rot90_ccw",
    "torch_rot90_ccw_transpose",
    "torch_rot90_cw",
    "torch_rot90_cw_transpose",
    "torch_transpose",
    "torch_transpose2",
    "torch_transpose_",
    "torch_transpose_rot90_ccw",
    "torch_transpose_rot90_cw",
    "unpad_image_tensor",
    "unpad_xyxy_bboxes",
]


def _one_view(x: Tensor, code: int) -> Tensor:
    return V.view_transform(x, [code], in_is_batch=True)


def torch_none(x: Tensor) -> Tensor:
    """Identity: returns the argument itself."""
    return x


def torch_rot90_ccw(x: Tensor) -> Tensor:
    """Quarter turn counter-clockwise: out[i][j] = x[j][N-1-i]."""
    return _one_view(x, N.ROT90_CCW)


def torch_rot90_cw(x: Tensor) -> Tensor:
    """Quarter turn clockwise: out[i][j] = x[N-1-j][i]."""
    return _one_view(x, N.ROT90_CW)


def torch_rot180(x: Tensor) -> Tensor:
    """Half turn: out[i][j] = x[H-1-i][W-1-j]."""
    return _one_view(x, N.ROT180)


def torch_flipud(x: Tensor) -> Tensor:
    """Vertical flip: out[i][j] = x[H-1-i][j]."""
    return _one_view(x, N.FLIPUD)


def torch_fliplr(x: Tensor) -> Tensor:
    """Horizontal flip: out[i][j] = x[i][W-1-j]."""
    return _one_view(x, N.FLIPLR)


def torch_transpose(x: Tensor) -> Tensor:
    """Main-diagonal transpose: out[i][j] = x[j][i]."""
    return _one_view(x, N.TRANSPOSE)


def torch_transpose2(x: Tensor) -> Tensor:
    """``x.transpose(3, 2)`` -- the same main-diagonal transpose."""
    return _one_view(x, N.TRANSPOSE)


def torch_transpose_(x: Tensor) -> Tensor:
    """In-place flavoured transpose of the reference (``x.transpose_(2, 3)``): the result is written back into ``x``
    (square inputs only, since the storage is reused).  Host tensors: ``x.transpose_(2, 3)`` itself."""
    if not x.is_cuda:
        return x.transpose_(2, 3)
    y = _one_view(x, N.TRANSPOSE)
    if x.shape[2] != x.shape[3]:
        raise ValueError("torch_transpose_ needs a square input on the native path")
    with torch.no_grad():
        x.copy_(y)
    return x


# compositions: (rot then transpose) and (transpose then rot) collapse to flips / anti-transpose
def torch_rot90_ccw_transpose(x: Tensor) -> Tensor:
    """rot90_ccw followed by transpose == horizontal flip."""
    return _one_view(x, N.FLIPLR)


def torch_rot90_cw_transpose(x: Tensor) -> Tensor:
    """rot90_cw followed by transpose == vertical flip."""
    return _one_view(x, N.FLIPUD)


def torch_rot180_transpose(x: Tensor) -> Tensor:
    """rot180 followed by transpose == anti-diagonal transpose: out[i][j] = x[N-1-j][N-1-i]."""
    return _one_view(x, N.ANTITRANSPOSE)


def torch_transpose_rot90_ccw(x: Tensor) -> Tensor:
    """transpose followed by rot90_ccw == vertical flip."""
    return _one_view(x, N.FLIPUD)


def torch_transpose_rot90_cw(x: Tensor) -> Tensor:
    """transpose followed by rot90_cw == horizontal flip."""
    return _one_view(x, N.FLIPLR)


def torch_transpose_rot180(x: Tensor) -> Tensor:
    """transpose followed by rot180 == anti-diagonal transpose."""
    return _one_view(x, N.ANTITRANSPOSE)


@pytorch_toolbelt_deprecated("Function torch_rot90 has been marked as deprecated. Please use torch_rot90_ccw instead")
def torch_rot90(x: Tensor):
    return torch_rot90_ccw(x)


@pytorch_toolbelt_deprecated("Function torch_rot270 has been marked as deprecated. Please use torch_rot90_cw instead")
def torch_rot270(x: Tensor):
    return torch_rot90_cw(x)


# ------------------------------------------------------------------------------------------- padding helpers
def pad_tensor_to_size(x: Tensor, size: Tuple[int, ...], mode="constant", value=0):
    """Centre ``x [B, C, *spatial]`` inside ``size`` (extra element goes after).  Returns (padded, crop slices)."""
    nd = len(size)
    if nd != x.dim() - 2:
        raise ValueError(f"Expected {nd} spatial dimensions, got {x.dim() - 2}")
    spatial = list(x.shape[-nd:])
    before = [(int(t) - int(s)) // 2 for t, s in zip(size, spatial)]
    after = [(int(t) - int(s)) - b for t, s, b in zip(size, spatial, before)]
    pad_args = tuple(itertools.chain(*reversed(list(zip(before, after)))))  # F.pad wants the last dim first
    padded = torch.nn.functional.pad(x, pad=pad_args, mode=mode, value=value)
    crop = [slice(None), slice(None)] + [slice(b, b + s) for b, s in zip(before, spatial)]
    return padded, crop


def pad_image_tensor(image_tensor: Tensor, pad_size: Union[int, Tuple[int, int]] = 32):
    """Pad ``[B, C, H, W]`` so H and W become multiples of ``pad_size``.  Returns (tensor, [left, right, top, bottom])."""
    if image_tensor.dim() != 4:
        raise ValueError("Tensor must have rank 4 ([B,C,H,W])")
    rows, cols = image_tensor.size(2), image_tensor.size(3)
    if isinstance(pad_size, Sized) and isinstance(pad_size, Iterable) and len(pad_size) == 2:
        mh, mw = (int(v) for v in pad_size)
    elif isinstance(pad_size, int):
        mh = mw = pad_size
    else:
        raise ValueError(f"Unsupported pad_size: {pad_size}, must be either tuple(pad_rows,pad_cols) or single int scalar.")

    def missing(n, m):
        if n > m:
            rem = n % m
            return m - rem if rem > 0 else 0
        return m - n

    pr, pc = missing(rows, mh), missing(cols, mw)
    if pr == 0 and pc == 0:
        return image_tensor, (0, 0, 0, 0)
    top, left = pr // 2, pc // 2
    pad = [left, pc - left, top, pr - top]
    return torch.nn.functional.pad(image_tensor, pad), pad


def unpad_image_tensor(image_tensor: Tensor, pad) -> Tensor:
    if image_tensor.dim() != 4:
        raise ValueError("Tensor must have rank 4 ([B,C,H,W])")
    left, right, top, bottom = pad
    rows, cols = image_tensor.size(2), image_tensor.size(3)
    return image_tensor[..., top:rows - bottom, left:cols - right]


def unpad_xyxy_bboxes(bboxes_tensor: Tensor, pad, dim=-1):
    left, _right, top, _bottom = pad
    shift = torch.tensor([left, top, left, top], dtype=bboxes_tensor.dtype).to(bboxes_tensor.device)
    if dim == -1:
        dim = bboxes_tensor.dim() - 1
    shape = [1] * bboxes_tensor.dim()
    shape[dim] = 4
    return bboxes_tensor - shift.view(shape)


# ------------------------------------------------------------------------------------------- reductions
def _reduce_dim(x: Tensor, dim: int, code: int, eps: float = V.DEFAULT_EPS) -> Tensor:
    if dim != 0:
        x = x.movedim(dim, 0)
    return V.stack_reduce(x.contiguous(), code, eps)


def geometric_mean(x: Tensor, dim: int) -> Tensor:
    """exp(mean(log x)) along ``dim`` -- for probabilities in (0, 1]; 0 gives 0, negatives give NaN."""
    return _reduce_dim(x, dim, N.RED_GMEAN)


def harmonic_mean(x: Tensor, dim: int, eps: float = 1e-6) -> Tensor:
    """1 / mean(1 / max(x, eps)) along ``dim`` (the result's denominator is clamped at eps too)."""
    return _reduce_dim(x, dim, N.RED_HMEAN, eps)


def harmonic1p_mean(x: Tensor, dim: int) -> Tensor:
    """1 / mean(1 / (x + 1)) - 1 along ``dim``."""
    return _reduce_dim(x, dim, N.RED_HARMONIC1P)


def logodd_mean(x: Tensor, dim: int, eps: float = 1e-6) -> Tensor:
    """sigmoid(mean(logit(clamp(x, eps, 1 - eps)))) along ``dim``."""
    return _reduce_dim(x, dim, N.RED_LOGODD, eps)


def log1p_mean(x: Tensor, dim: int) -> Tensor:
    """exp(mean(log1p x)) - 1 along ``dim`` (non-negative inputs)."""
    return _reduce_dim(x, dim, N.RED_LOG1P)
