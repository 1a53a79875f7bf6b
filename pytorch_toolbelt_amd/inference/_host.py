"""Host-tensor evaluation of the inference surface: what runs when the caller hands in CPU tensors.

The reference is device-agnostic -- its mergers default to ``device="cpu"`` (inference/tiles.py:295), its TTA functions take any
tensor and "respect gradients flow" (inference/tta.py:1-5), its own tests feed CPU tensors -- so a drop-in has to accept those calls
too: a DataLoader worker that augments on the host, a CPU-only CI box, the reference's unmodified test-suite.  This module is that
path, written as plain differentiable torch ops of any floating dtype.

The dispatch rule is the tensor's (or the merger's) DEVICE and nothing else: a CUDA tensor always takes the HIP kernels and
fails loudly when ``libptb_hip.so`` is missing or refuses the call -- it is never re-routed here.  Nothing in this module is used
by, or can stand in for, the MI355X path, and it shares no code with the test oracle (``oracle/`` is never imported by the package).

View codes (include/ptb_hip.h): bit 0 transpose, bit 1 flip source rows, bit 2 flip source columns, i.e.
``out[i][j] = src[r][c]`` with ``(r, c) = (j, i) if transpose else (i, j)``, then ``r -> rows - 1 - r`` / ``c -> cols - 1 - c``.
"""
from typing import Sequence

import numpy as np
import torch

RED_SUM, RED_MEAN, RED_GMEAN, RED_HMEAN, RED_HARMONIC1P, RED_LOGODD, RED_LOG1P = range(7)


def apply_view(x: torch.Tensor, code: int) -> torch.Tensor:
    """One element of D4 acting on dims (2, 3) of a tensor of rank >= 4 -- ``x.flip(2)`` / ``x.flip(3)`` / ``x.transpose(2, 3)`` like
    the reference's chains (inference/functional.py:47-132); dims beyond the fourth ride along, a tensor of lower rank raises what
    torch raises for a dim that does not exist.  A strided view of ``x``."""
    flips = [d for d, bit in ((2, 2), (3, 4)) if code & bit]
    y = x.flip(flips) if flips else x
    return y.transpose(2, 3) if code & 1 else y


def _needs_square(views, x):
    """Views that transpose and views that do not give different shapes on a non-square plane: they cannot share one batch
    (d4_image_augment says so, inference/tta.py:399-403).  A call whose views all transpose -- or none -- is fine on any plane."""
    n_t = sum(v & 1 for v in views)
    if n_t and n_t != len(views) and x.dim() >= 4 and x.shape[2] != x.shape[3]:
        raise ValueError(f"Input tensor must have number of rows equal to number of cols. Got input tensor of shape {x.size()}")


def view_transform(x: torch.Tensor, views: Sequence[int], in_is_batch: bool = True, scale: float = 1.0) -> torch.Tensor:
    """``in_is_batch``: out = cat_k view_k(x) (augment); else ``x`` is the chunk-major [V*B, ...] stack and chunk k gets view k."""
    _needs_square(views, x)
    V = len(views)
    if in_is_batch:
        parts = [apply_view(x, c) for c in views]
    else:
        if x.shape[0] % V:
            raise RuntimeError(f"Input batch size ({x.size(0)}) must be divisible by {V}.")
        parts = [apply_view(chunk, c) for chunk, c in zip(torch.chunk(x, V), views)]
    out = torch.cat(parts, dim=0) if V > 1 else parts[0]
    return out if scale == 1.0 else out * scale


def reduce_stack(stack: torch.Tensor, code: int, eps: float = 1e-6) -> torch.Tensor:
    """The reductions of ``_deaugment_averaging`` over dim 0 (inference/tta.py:63-96, inference/functional.py:250-333), in the
    reference's operation order."""
    if code == RED_SUM:
        return stack.sum(dim=0)
    if code == RED_MEAN:
        return stack.mean(dim=0)
    if code == RED_GMEAN:
        return stack.log().mean(dim=0).exp()
    if code == RED_HMEAN:
        inv = torch.reciprocal(stack.clamp_min(eps)).mean(dim=0)
        return torch.reciprocal(inv.clamp_min(eps))
    if code == RED_HARMONIC1P:
        return torch.reciprocal(torch.reciprocal(stack + 1).mean(dim=0)) - 1
    if code == RED_LOGODD:
        p = stack.clamp(min=eps, max=1.0 - eps)
        m = torch.log(p / (1 - p)).mean(dim=0)
        return torch.exp(m) / (1 + torch.exp(m))
    if code == RED_LOG1P:
        return torch.exp(torch.log1p(stack).mean(dim=0)) - 1
    raise KeyError(f"Unsupported reduction code {code}")


def deaug_reduce(x: torch.Tensor, views: Sequence[int], code: int) -> torch.Tensor:
    """``reduce_k view_k(chunk_k(x))``: undo every chunk's transform, stack, reduce."""
    V = len(views)
    if x.shape[0] % V:
        raise RuntimeError(f"Input batch size ({x.size(0)}) must be divisible by {V}.")
    _needs_square(views, x)
    return reduce_stack(torch.stack([apply_view(chunk, c) for chunk, c in zip(torch.chunk(x, V), views)]), code)


def resize(x: torch.Tensor, size, mode: str, align_corners) -> torch.Tensor:
    return torch.nn.functional.interpolate(x, size=(int(size[0]), int(size[1])), mode=mode, align_corners=align_corners)


def ms_reduce(maps, size, align_corners, code: int) -> torch.Tensor:
    size = (int(size[0]), int(size[1]))
    same = [m if tuple(m.shape[2:]) == size else resize(m, size, "bilinear", align_corners) for m in maps]
    return reduce_stack(torch.stack(same), code)


class HostTileMerger:
    """The accumulators of a ``TileMerger(device="cpu")`` and the arithmetic on them (reference inference/tiles.py:290-350):
    ``image`` / ``norm_mask`` / ``weight`` are plain tensors of the caller's ``dtype`` on the host, tiles are blended one after the
    other in batch order (``image[:, window] += tile * weight``), ``merge()`` is ``image / norm_mask`` without an eps clamp.
    ``inference.tiles.TileMerger`` wraps this for ``device="cpu"``; see there for the public surface."""

    def __init__(self, image_shape, channels, weight, device, dtype):
        self.image_height, self.image_width, self.channels = image_shape[0], image_shape[1], channels
        window = torch.from_numpy(np.expand_dims(weight, axis=0)) if isinstance(weight, np.ndarray) else torch.as_tensor(weight).unsqueeze(0)
        self.weight = window.to(device=device, dtype=dtype)
        self.image = torch.zeros((channels, self.image_height, self.image_width), device=device, dtype=dtype)
        self.norm_mask = torch.zeros((1, self.image_height, self.image_width), device=device, dtype=dtype)

    def blend(self, tiles: torch.Tensor, crop_coords):
        """tiles [B, C, h, w] already on the accumulators' device and of their dtype."""
        image, norm, w = self.image, self.norm_mask, self.weight
        for tile, box in zip(tiles, crop_coords):
            x, y, tw, th = (int(v) for v in box)
            rows, cols = slice(y, y + th), slice(x, x + tw)
            image[:, rows, cols] += tile * w
            norm[:, rows, cols] += w

    def reset(self):
        self.image.zero_()
        self.norm_mask.zero_()
