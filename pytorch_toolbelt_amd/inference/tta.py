"""Test-time augmentation for segmentation / classification (drop-in for ``pytorch_toolbelt.inference.tta``).

Augment = concatenate D4-group views of the input along the batch dim; de-augment = split the model output into
chunks, undo each chunk's transform, stack and reduce.  Here every ``*_image_augment`` is one HIP gather launch
and every ``*_image_deaugment`` is one fused launch (inverse transform + reduction in registers), instead of the
reference's cat/stack of strided views followed by a second pass for the mean (reference inference/tta.py:257-524).
The functions stay differentiable (linear reductions) through custom autograd functions whose backward passes are
the same kernels run with the inverse views.
"""
from functools import partial
from typing import Callable, Dict, List, Mapping, Optional, Tuple, Union

import sys

import torch
from torch import Tensor, nn

from .. import _native as N
from ..utils.support import pytorch_toolbelt_deprecated
from . import _lazy
from . import _views as V
from . import functional as F

__all__ = [
    "GeneralizedTTA",
    "MultiscaleTTA",
    "d2_image_augment",
    "d2_labels_augment",
    "d2_image_deaugment",
    "d2_labels_deaugment",
    "d4_image2label",
    "d4_image2mask",
    "d4_image_augment",
    "d4_labels_augment",
    "d4_image_deaugment",
    "d4_labels_deaugment",
    "fivecrop_image2label",
    "fivecrop_image_augment",
    "fivecrop_label_deaugment",
    "fliplr_image2label",
    "fliplr_image2mask",
    "fliplr_image_augment",
    "fliplr_labels_augment",
    "fliplr_image_deaugment",
    "fliplr_labels_deaugment",
    "flips_image_augment",
    "flips_labels_augment",
    "flips_image_deaugment",
    "flips_labels_deaugment",
    "flipud_image_augment",
    "flipud_image_deaugment",
    "flipud_labels_deaugment",
    "ms_image_augment",
    "ms_labels_augment",
    "ms_image_deaugment",
    "ms_flips_image_deaugment",
    "tencrop_image2label",
]

MaybeStrOrCallable = Optional[Union[str, Callable]]

# Forward views in concatenation order, and the inverse views applied chunk by chunk when de-augmenting.
# d4 (reference tta.py:409-422): x, rot90_cw(x), rot180(x), rot90_ccw(x), xT, rot90_cw(xT) [= fliplr],
# rot180(xT) [= anti-transpose], rot90_ccw(xT) [= flipud]; inverses per tta.py:455-466.
AUGMENT_VIEWS = {
    "fliplr": (N.IDENT, N.FLIPLR),
    "flipud": (N.IDENT, N.FLIPUD),
    "flips": (N.IDENT, N.FLIPLR, N.FLIPUD),
    "d2": (N.IDENT, N.FLIPLR, N.FLIPUD, N.ROT180),
    "d4": (N.IDENT, N.ROT90_CW, N.ROT180, N.ROT90_CCW, N.TRANSPOSE, N.FLIPLR, N.ANTITRANSPOSE, N.FLIPUD),
}
DEAUGMENT_VIEWS = {g: tuple(V.INVERSE[c] for c in views) for g, views in AUGMENT_VIEWS.items()}


def _reduction_code(reduction):
    """HIP reduction code for a string reduction; None when the caller must handle it (no reduction / callable)."""
    if isinstance(reduction, str) and reduction in V.REDUCTION_CODES:
        return V.REDUCTION_CODES[reduction]
    return None


def set_lazy_deaugment(flag: bool) -> bool:
    """Extension: switch the lazy results of ``*_image_deaugment`` on / off (default on; ``PTB_LAZY_DEAUG=0``).  Returns the
    previous setting.  See ``inference/_lazy.py`` for what "lazy" means and why nothing but the speed changes."""
    return _lazy.set_enabled(flag)


def split_into_chunks(input: Tensor, batch_size: int) -> Tuple[Tensor, ...]:
    """Split dim 0 into ``batch_size`` equal chunks (the argument is the NUMBER of chunks, as in the reference)."""
    if not torch.jit.is_scripting() and not torch.jit.is_tracing():
        if input.size(0) % batch_size != 0:
            raise RuntimeError(f"Input batch size ({input.size(0)}) must be divisible by {batch_size}.")
    return torch.chunk(input, batch_size)


def _deaugment_averaging(x: Tensor, reduction: MaybeStrOrCallable) -> Tensor:
    """Reduce the TTA dimension (dim 0) of ``x [T, B, ...]``.

    "mean" | "sum" | "gmean"/"geometric_mean" | "hmean"/"harmonic_mean" | "harmonic1p" | "logodd" | "log1p" run as one
    HIP kernel; a callable is invoked as ``reduction(x, dim=0)``; None / "None" / "none" returns ``x`` unchanged.
    """
    code = _reduction_code(reduction)
    if code is not None:
        return V.stack_reduce(x, code)
    if callable(reduction):
        return reduction(x, dim=0)
    if reduction in {None, "None", "none"}:
        return x
    raise KeyError(f"Unsupported reduction mode {reduction}")


def _image_augment(image: Tensor, group: str) -> Tensor:
    return V.view_transform(image, AUGMENT_VIEWS[group], in_is_batch=True)


def _image_deaugment(image: Tensor, group: str, reduction: MaybeStrOrCallable, lazy: bool = True, owned: bool = False) -> Tensor:
    views = DEAUGMENT_VIEWS[group]
    if image.size(0) % len(views) != 0:
        raise RuntimeError(f"Input batch size ({image.size(0)}) must be divisible by {len(views)}.")
    code = _reduction_code(reduction)
    if code is not None:
        # inference-shaped calls come back as a handle that `TileMerger.integrate_batch` fuses into its own launch and that
        # turns into the real tensor on any other use (inference/_lazy.py); everything else is evaluated here and now
        handle = _lazy.maybe_lazy(image, group, views, code, V.deaug_reduce, owned=owned) if lazy else None
        return handle if handle is not None else V.deaug_reduce(image, views, code)
    if not (callable(reduction) or reduction in {None, "None", "none"}):
        raise KeyError(f"Unsupported reduction mode {reduction}")
    stack = V.view_transform(image, views, in_is_batch=False)
    stack = stack.view(len(views), image.size(0) // len(views), *image.shape[1:])
    return reduction(stack, dim=0) if callable(reduction) else stack


def _labels_deaugment(logits: Tensor, n: int, reduction: MaybeStrOrCallable, order=None) -> Tensor:
    chunks = split_into_chunks(logits, n)
    if order is None:  # chunk-major input is already the [T, B, ...] stack
        stack = logits.reshape(n, logits.size(0) // n, *logits.shape[1:])
    else:
        stack = torch.stack([chunks[i] for i in order])
    return _deaugment_averaging(stack, reduction=reduction)


# ------------------------------------------------------------------------------------------------- five / ten crop
def _corner_and_center_crops(image: Tensor, crop_size: Tuple[int, int]):
    rows, cols = int(image.size(2)), int(image.size(3))
    ch, cw = crop_size
    if ch > rows:
        raise ValueError(f"Tensor height ({rows}) is less than requested crop size ({ch})")
    if cw > cols:
        raise ValueError(f"Tensor width ({cols}) is less than requested crop size ({cw})")
    y1, x1 = rows - ch, cols - cw
    yc, xc = (rows - ch) // 2, (cols - cw) // 2
    return [
        image[..., :ch, :cw],
        image[..., :ch, x1:],
        image[..., y1:, :cw],
        image[..., y1:, x1:],
        image[..., yc:yc + ch, xc:xc + cw],
    ]


def fivecrop_image_augment(image: Tensor, crop_size: Tuple[int, int]) -> Tensor:
    """Top-left, top-right, bottom-left, bottom-right and centre crops concatenated along the batch dim."""
    return torch.cat(_corner_and_center_crops(image, crop_size), dim=0)


def fivecrop_label_deaugment(logits: Tensor, reduction: MaybeStrOrCallable = "mean") -> Tensor:
    return _labels_deaugment(logits, 5, reduction)


def fivecrop_image2label(model: nn.Module, image: Tensor, crop_size: Tuple) -> Tensor:
    """Average the classifier's predictions over the five crops."""
    return fivecrop_label_deaugment(model(fivecrop_image_augment(image, crop_size)))


def tencrop_image2label(model: nn.Module, image: Tensor, crop_size: Tuple) -> Tensor:
    """Five crops plus their horizontal mirrors: ten sequential forward passes, arithmetic mean."""
    assert crop_size[0] <= int(image.size(2))
    assert crop_size[1] <= int(image.size(3))
    total = None
    for crop in _corner_and_center_crops(image, crop_size):
        assert crop.size(2) == crop_size[0] and crop.size(3) == crop_size[1]
        for inp in (crop, F.torch_fliplr(crop.contiguous())):
            pred = model(inp)
            total = pred if total is None else total + pred
    return total * float(1.0 / 10.0)


# ------------------------------------------------------------------------------------------------- model wrappers
def fliplr_image2label(model: nn.Module, image: Tensor) -> Tensor:
    return fliplr_labels_deaugment(model(fliplr_image_augment(image)))


def fliplr_image2mask(model: nn.Module, image: Tensor) -> Tensor:
    return fliplr_image_deaugment(model(fliplr_image_augment(image)))


def d4_image2label(model: nn.Module, image: Tensor) -> Tensor:
    return d4_labels_deaugment(model(d4_image_augment(image)))


def d4_image2mask(model: nn.Module, image: Tensor) -> Tensor:
    return d4_image_deaugment(model(d4_image_augment(image)))


# ------------------------------------------------------------------------------------------------- image augment
def fliplr_image_augment(image: Tensor) -> Tensor:
    """[B,C,H,W] -> [2B,C,H,W]: original, horizontally flipped."""
    return _image_augment(image, "fliplr")


def flipud_image_augment(image: Tensor) -> Tensor:
    """[B,C,H,W] -> [2B,C,H,W]: original, vertically flipped."""
    return _image_augment(image, "flipud")


def flips_image_augment(image: Tensor) -> Tensor:
    """[B,C,H,W] -> [3B,C,H,W]: original, horizontally flipped, vertically flipped."""
    return _image_augment(image, "flips")


def d2_image_augment(image: Tensor) -> Tensor:
    """[B,C,H,W] -> [4B,C,H,W]: original, fliplr, flipud, rot180."""
    return _image_augment(image, "d2")


def d4_image_augment(image: Tensor) -> Tensor:
    """[B,C,N,N] -> [8B,C,N,N]: the eight symmetries of the square (x and its transpose, each rotated 0/90/180/270)."""
    if not torch.jit.is_scripting() and not torch.jit.is_tracing():
        if image.size(2) != image.size(3):
            raise ValueError(
                f"Input tensor must have number of rows equal to number of cols. Got input tensor of shape {image.size()}"
            )
    return _image_augment(image, "d4")


# ------------------------------------------------------------------------------------------------- image de-augment
# A lazy result reads its argument LATER, so it is only handed out where a change of the argument in between would be noticed (a
# version counter) -- or could not happen: the argument is a temporary nobody else references, `deaugment(model(x))` written as one
# expression.  That is what `sys.getrefcount` says inside the public function itself: as many references as an argument created in
# the call expression has (calibrated once, `_TEMP_REFS`; a tensor bound to a name, kept by the model or sitting in a list shows
# more).  It matters under `torch.inference_mode()`, whose tensors carry no version counter (`_lazy.maybe_lazy` also asks the storage
# for other owners).
def _refs_of_argument(image, reduction="mean"):
    return sys.getrefcount(image)


_TEMP_REFS = _refs_of_argument(object())


def _probe_owned(image, reduction="mean"):
    # (the shape of the public de-augment functions: the count is taken in the function's own frame)
    return sys.getrefcount(image) <= _TEMP_REFS


def _refcount_probe_works():
    """the ownership test rests on CPython's reference counting and on how a call passes its arguments.  Checked once at
    import on this interpreter: a temporary of the call expression must read as owned, the same object bound to a name, kept in a list or
    passed by keyword from a name must not.  If any of that fails (another interpreter, a tracing tool that holds references) the test
    is switched off: `owned` is then never true and tensors without a version counter are always evaluated on the spot."""
    try:
        named = object()
        box = [object()]
        return (_probe_owned(object()) and not _probe_owned(named) and not _probe_owned(box[0]) and not _probe_owned(image=named)
                and _probe_owned(image=object()))
    except Exception:  # noqa: BLE001
        return False


if not _refcount_probe_works():
    _TEMP_REFS = -1      # (no argument ever counts as a temporary of the call expression)


def fliplr_image_deaugment(image: Tensor, reduction: MaybeStrOrCallable = "mean") -> Tensor:
    """[2B,C,H,W] -> [B,C,H,W] (or the [2,B,C,H,W] stack when reduction is None)."""
    owned = sys.getrefcount(image) <= _TEMP_REFS      # (its own statement: inside the call below `image` already sits on the stack once more)
    return _image_deaugment(image, "fliplr", reduction, owned=owned)


def flipud_image_deaugment(image: Tensor, reduction: MaybeStrOrCallable = "mean") -> Tensor:
    owned = sys.getrefcount(image) <= _TEMP_REFS      # (its own statement: inside the call below `image` already sits on the stack once more)
    return _image_deaugment(image, "flipud", reduction, owned=owned)


def flips_image_deaugment(image: Tensor, reduction: MaybeStrOrCallable = "mean") -> Tensor:
    owned = sys.getrefcount(image) <= _TEMP_REFS      # (its own statement: inside the call below `image` already sits on the stack once more)
    return _image_deaugment(image, "flips", reduction, owned=owned)


def d2_image_deaugment(image: Tensor, reduction: MaybeStrOrCallable = "mean") -> Tensor:
    owned = sys.getrefcount(image) <= _TEMP_REFS      # (its own statement: inside the call below `image` already sits on the stack once more)
    return _image_deaugment(image, "d2", reduction, owned=owned)


def d4_image_deaugment(image: Tensor, reduction: MaybeStrOrCallable = "mean") -> Tensor:
    """[8B,C,N,N] -> [B,C,N,N] (or the [8,B,C,N,N] stack when reduction is None)."""
    owned = sys.getrefcount(image) <= _TEMP_REFS      # (its own statement: inside the call below `image` already sits on the stack once more)
    return _image_deaugment(image, "d4", reduction, owned=owned)


# ------------------------------------------------------------------------------------------------- labels
def fliplr_labels_augment(labels: Tensor) -> Tensor:
    return torch.cat([labels] * 2, dim=0)


def flips_labels_augment(labels: Tensor) -> Tensor:
    return torch.cat([labels] * 3, dim=0)


def d2_labels_augment(labels: Tensor) -> Tensor:
    return torch.cat([labels] * 4, dim=0)


def d4_labels_augment(labels: Tensor) -> Tensor:
    return torch.cat([labels] * 8, dim=0)


def fliplr_labels_deaugment(logits: Tensor, reduction: MaybeStrOrCallable = "mean") -> Tensor:
    return _labels_deaugment(logits, 2, reduction)


def flipud_labels_deaugment(logits: Tensor, reduction: MaybeStrOrCallable = "mean") -> Tensor:
    return _labels_deaugment(logits, 2, reduction)


def flips_labels_deaugment(logits: Tensor, reduction: MaybeStrOrCallable = "mean") -> Tensor:
    if logits.size(0) % 3 != 0:
        raise RuntimeError("Batch size must be divisible by 3")
    return _labels_deaugment(logits, 3, reduction)


def d2_labels_deaugment(logits: Tensor, reduction: MaybeStrOrCallable = "mean") -> Tensor:
    return _labels_deaugment(logits, 4, reduction)


def d4_labels_deaugment(image: Tensor, reduction: MaybeStrOrCallable = "mean") -> Tensor:
    """Reference quirk kept on purpose (inference/tta.py:437): chunk 6 is dropped and chunk 7 counted twice."""
    return _labels_deaugment(image, 8, reduction, order=(0, 1, 2, 3, 4, 6, 6, 7))


@pytorch_toolbelt_deprecated("This class is deprecated. Please use GeneralizedTTA instead")
class TTAWrapper(nn.Module):
    def __init__(self, model: nn.Module, tta_function, **kwargs):
        super().__init__()
        self.model = model
        self.tta = partial(tta_function, **kwargs)

    def forward(self, *input):
        return self.tta(self.model, *input)


# ------------------------------------------------------------------------------------------------- multiscale
def _offset_pair(offset):
    if isinstance(offset, (tuple, list)):
        return offset[0], offset[1]
    return offset, offset


def ms_labels_augment(labels: Tensor, size_offsets: List[Union[int, Tuple[int, int]]]) -> List[Tensor]:
    return [labels] * len(size_offsets)


def ms_image_augment(image: Tensor, size_offsets: List[Union[int, Tuple[int, int]]], mode="bilinear", align_corners=False) -> List[Tensor]:
    """List of resized copies of ``image``: one per pixel offset (rows+dr, cols+dc); offset 0 returns the input itself."""
    rows, cols = image.size(2), image.size(3)
    out = []
    for offset in size_offsets:
        dr, dc = _offset_pair(offset)
        if dr == 0 and dc == 0:
            out.append(image)
        else:
            out.append(_resize(image, (rows + dr, cols + dc), mode, align_corners))
    return out


def ms_labels_deaugment(logits: List[Tensor], size_offsets: List[Union[int, Tuple[int, int]]], reduction: MaybeStrOrCallable = "mean"):
    if len(logits) != len(size_offsets):
        raise ValueError("Number of images must be equal to number of size offsets")
    return _deaugment_averaging(torch.stack(logits), reduction=reduction)


def ms_image_deaugment(images: List[Tensor], size_offsets: List[Union[int, Tuple[int, int]]], reduction: MaybeStrOrCallable = "mean",
                       mode: str = "bilinear", align_corners: bool = True, stride: int = 1) -> Tensor:
    """Resize every scale's prediction back to ``rows - offset // stride`` (floor division, like the reference) and reduce."""
    if len(images) != len(size_offsets):
        raise ValueError("Number of images must be equal to number of size offsets")
    fused = _ms_fuse_lazy(images, size_offsets, reduction, mode, align_corners, stride)
    if fused is not None:
        return fused
    return _ms_image_deaugment(images, size_offsets, reduction, mode, align_corners, stride)


def _ms_fuse_lazy(images, size_offsets, reduction, mode, align_corners, stride):
    """The reference's composition for multiscale + flip TTA, ``ms_image_deaugment([<group>_image_deaugment(y_s) for s ...])``
    (tta.py:287-316 feeding :645-689; what ``MultiscaleTTA`` around a flip-TTA model computes): when every scale arrives as a lazy
    de-augmentation handle of the same flip group and reduction, nothing has been computed yet and the whole thing is ONE pass over
    every view of every scale (``ptb_ms_flip_deaug_reduce``) instead of a launch per scale that writes a map and one more that
    reads them back.  None: not that shape (the caller evaluates the handles and composes)."""
    if mode != "bilinear" or _reduction_code(reduction) is None or not images or not all(type(t) is _lazy.LazyDeaugment for t in images):
        return None
    first = images[0]
    if first._value is not None or any(t._value is not None or t._group != first._group or t._code != first._code for t in images):
        return None
    if any(v & 1 for v in first._views):
        return None                       # transposing groups are not combined with multiscale in the one-pass kernel
    if any(t.dtype != torch.float32 for t in images):
        return None                       # half-precision handles stand for half tensors (rounded per scale): evaluated, then composed
    taken = [t._take_source() for t in images]
    if any(x is None for x in taken):
        return None
    _lazy.fused += len(images)
    return ms_flips_image_deaugment([x[0] for x in taken], size_offsets, group=first._group, inner_reduction=V.REDUCTION_NAMES[first._code],
                                    reduction=reduction, mode=mode, align_corners=align_corners, stride=stride)


def _ms_image_deaugment(images, size_offsets, reduction, mode, align_corners, stride):
    code = _reduction_code(reduction)
    if code is not None and mode == "bilinear" and 1 <= len(images) <= 8:
        # fused path: every scale is sampled at the target grid and reduced in registers (one HIP launch)
        sizes = set()
        for fmap, offset in zip(images, size_offsets):
            dr, dc = _offset_pair(offset)
            sizes.add((fmap.size(2) - dr // stride, fmap.size(3) - dc // stride) if (dr != 0 or dc != 0) else (fmap.size(2), fmap.size(3)))
        if len(sizes) == 1:
            from . import _resample

            return _resample.ms_reduce(list(images), sizes.pop(), align_corners, code)
    restored = []
    for fmap, offset in zip(images, size_offsets):
        dr, dc = _offset_pair(offset)
        if dr == 0 and dc == 0:
            restored.append(fmap)
        else:
            size = fmap.size(2) - dr // stride, fmap.size(3) - dc // stride
            restored.append(_resize(fmap, size, mode, align_corners))
    return _deaugment_averaging(torch.stack(restored), reduction=reduction)


def ms_flips_image_deaugment(images: List[Tensor], size_offsets: List[Union[int, Tuple[int, int]]], group: str = "fliplr",
                             inner_reduction: MaybeStrOrCallable = "mean", reduction: MaybeStrOrCallable = "mean", mode: str = "bilinear",
                             align_corners: bool = True, stride: int = 1) -> Tensor:
    """Extension (BASELINE configs[4]): multiscale TTA whose every scale is itself flip-augmented, merged in ONE pass.

    ``images[s]`` is the model output for the ``<group>_image_augment``-ed input of scale ``s`` (``[V*B, C, h_s, w_s]``,
    chunk-major).  Equals ``ms_image_deaugment([<group>_image_deaugment(y, inner_reduction) for y in images], size_offsets,
    reduction, mode, align_corners, stride)`` -- which is also what runs whenever the fused kernel does not take the
    configuration (transposing groups, callables, non-bilinear modes, autograd, widths not divisible by 4)."""
    if group not in DEAUGMENT_VIEWS:
        raise KeyError(group)
    if len(images) != len(size_offsets):
        raise ValueError("Number of images must be equal to number of size offsets")
    views = DEAUGMENT_VIEWS[group]
    inner, outer = _reduction_code(inner_reduction), _reduction_code(reduction)
    if (inner is not None and outer is not None and mode == "bilinear" and 1 <= len(images) <= 8
            and not (torch.is_grad_enabled() and any(t.requires_grad for t in images))):
        sizes = set()
        for fmap, offset in zip(images, size_offsets):
            dr, dc = _offset_pair(offset)
            sizes.add((fmap.size(2) - dr // stride, fmap.size(3) - dc // stride) if (dr != 0 or dc != 0) else (fmap.size(2), fmap.size(3)))
        if len(sizes) == 1:
            from . import _resample

            out = _resample.ms_flip_reduce(list(images), views, sizes.pop(), align_corners, inner, outer)
            if out is not None:
                return out
    # (the composed path wants the maps themselves: no handles that would come straight back here)
    per_scale = [_image_deaugment(y, group, inner_reduction, lazy=False) for y in images]
    return _ms_image_deaugment(per_scale, size_offsets, reduction, mode, align_corners, stride)


def _resize(x: Tensor, size, mode, align_corners):
    from . import _resample

    return _resample.resize(x, size, mode, align_corners)


# ------------------------------------------------------------------------------------------------- nn.Module wrappers
class GeneralizedTTA(nn.Module):
    """``deaugment_fn(model(augment_fn(x)))`` where the functions may be single callables, lists (positional
    inputs / outputs) or dicts (keyword inputs / dict outputs)."""

    __slots__ = ["augment_fn", "deaugment_fn"]

    def __init__(self, model: Union[nn.Module, nn.DataParallel], augment_fn: Union[Callable, Dict[str, Callable], List[Callable]],
                 deaugment_fn: Union[Callable, Dict[str, Callable], List[Callable]]):
        super().__init__()
        self.model = model
        self.augment_fn = augment_fn
        self.deaugment_fn = deaugment_fn

    def forward(self, *input, **kwargs):
        aug = self.augment_fn
        if isinstance(aug, dict):
            if len(input) != 0:
                raise ValueError("Input for GeneralizedTTA must not have positional arguments when augment_fn is dictionary")
            outputs = self.model(**{key: fn(kwargs[key]) for key, fn in aug.items()})
        elif isinstance(aug, (list, tuple)):
            if len(kwargs) != 0:
                raise ValueError("Input for GeneralizedTTA must be exactly one tensor")
            outputs = self.model(*[fn(x) for x, fn in zip(input, aug)])
        else:
            if len(input) != 1 or len(kwargs) != 0:
                raise ValueError("Input for GeneralizedTTA must be exactly one tensor")
            outputs = self.model(aug(input[0]))

        deaug = self.deaugment_fn
        if isinstance(deaug, dict):
            if not isinstance(outputs, dict):
                raise ValueError("Output of the model must be a dict")
            return {key: deaug[key](outputs[key]) for key in deaug.keys()}
        if isinstance(deaug, (list, tuple)):
            if not isinstance(outputs, (dict, tuple)):
                raise ValueError("Output of the model must be a dict")
            return [fn(value) for value, fn in zip(outputs, deaug)]
        return deaug(outputs)


class MultiscaleTTA(nn.Module):
    """Run the model at several input sizes (sequentially) and merge the predictions at the original size."""

    def __init__(self, model: nn.Module, size_offsets: List[int], mode: str = "bilinear", align_corners: bool = False,
                 augment_fn: Callable = ms_image_augment, deaugment_fn: Union[Callable, Dict[str, Callable]] = ms_image_deaugment):
        self.keys = set(deaugment_fn.keys()) if isinstance(deaugment_fn, Mapping) else None
        super().__init__()
        self.model = model
        self.size_offsets = size_offsets
        self.mode = mode
        self.align_corners = align_corners
        self.augment_fn = augment_fn
        self.deaugment_fn = deaugment_fn

    def forward(self, x):
        ms_inputs = self.augment_fn(x, size_offsets=self.size_offsets, mode=self.mode, align_corners=self.align_corners)
        ms_outputs = [self.model(inp) for inp in ms_inputs]
        if self.keys is None:
            # reference quirk (inference/tta.py:790): mode / align_corners are NOT forwarded on this path, so the
            # de-augmentation runs with its own defaults (bilinear, align_corners=True)
            return self.deaugment_fn(ms_outputs, self.size_offsets)
        outputs = {}
        for key in self.keys:
            values = [out[key] for out in ms_outputs]
            outputs[key] = self.deaugment_fn[key](values, size_offsets=self.size_offsets, mode=self.mode, align_corners=self.align_corners)
        return outputs
