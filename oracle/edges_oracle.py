"""Oracle (test infrastructure only) for the two ends of the tiled-inference loop.

The reference has no single function for either end; these are the compositions its README loop performs
(``README.md:196-227``), restated in numpy:

* tiles -> model input:  ``ImageSlicer.split`` (``inference/tiles.py:177-204``) -> ``tensor_from_rgb_image`` /
  ``image_to_tensor`` (``utils/torch_utils.py:204-237``: HWC -> CHW) -> ``.float()`` [-> per-channel affine]
  [-> ``*_image_augment``, ``inference/tta.py:257-284,319-341,385-422,470-484``];
* accumulators -> result: ``TileMerger.merge`` (``tiles.py:345-346``) -> ``np.moveaxis(to_numpy(..), 0, -1)``
  [-> ``.astype(np.uint8)`` | ``argmax``] -> ``ImageSlicer.crop_to_orignal_size`` (``tiles.py:271-280``).
"""
import numpy as np

from . import tiles_oracle as TO
from . import tta_oracle as AO


def tiles_to_batch(image, geom, indices=None, scale=None, bias=None, value=0, augment=None):
    """fp32 ``[V*n, C, th, tw]`` model input for the tiles ``indices`` (default: all) of ``image`` (uint8 HWC or HW).

    tiles.py:182-198 pads the whole image with a constant and slices the crops; torch_utils.py:223-230 moves the channel
    axis first (a 2-D tile gets a leading channel axis); ``.float()`` is exact for uint8; the optional affine is the two
    float32 roundings of ``x * scale + bias`` with per-channel ``[1, C, 1, 1]`` operands."""
    tiles = TO.split(image, geom, value)
    if indices is None:
        indices = range(len(tiles))
    chw = []
    for i in indices:
        t = tiles[i]
        chw.append(t[None] if t.ndim == 2 else np.moveaxis(t, -1, 0))
    batch = np.stack(chw).astype(np.float32)
    if scale is not None:
        s = np.asarray(scale, dtype=np.float32).reshape(1, -1, 1, 1)
        b = np.asarray(bias, dtype=np.float32).reshape(1, -1, 1, 1)
        batch = (batch * s).astype(np.float32) + b
    if augment is not None:
        batch = AO.image_augment(batch, augment)
    return np.ascontiguousarray(batch)


def cast_u8(x):
    """``ndarray.astype(np.uint8)`` of float32 as x86-64 numpy performs it: truncate toward zero to int32, keep the
    low byte; NaN, infinities and anything outside the int32 range give 0."""
    x = np.asarray(x, dtype=np.float32)
    ok = np.abs(x) < 2147483648.0  # False for NaN
    t = np.where(ok, x, 0).astype(np.int64)
    return (t & 255).astype(np.uint8)


def merge_crop(state, geom, image_shape, layout="hwc", kind="float32"):
    """Cropped result of a merger ``state`` (``tiles_oracle.merger_new`` dict).

    kind: "float32" | "uint8" (truncating cast, README.md:225) | "argmax_u8" | "argmax_i64" (over channels, first maximum,
    NaN counts as the maximum like numpy / torch).  layout "hwc" = the reference's moveaxis result, "chw" = no moveaxis."""
    merged = TO.merger_merge(state)                                   # [C, H', W']
    left, _r, top, _b = geom["margins"]
    H, W = int(image_shape[0]), int(image_shape[1])
    win = merged[:, top:top + H, left:left + W]
    if kind.startswith("argmax"):
        out = np.argmax(win, axis=0)                                  # numpy: first max, NaN is max
        return out.astype(np.uint8 if kind == "argmax_u8" else np.int64)
    if layout == "hwc":
        win = np.moveaxis(win, 0, -1)
    win = np.ascontiguousarray(win)
    return cast_u8(win) if kind == "uint8" else win.astype(np.float32)
