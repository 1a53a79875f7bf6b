"""Oracle (test infrastructure only) for inference/tiles.py of the reference.

numpy restatement; citations are ``pytorch_toolbelt/inference/tiles.py:LINE``.
"""
import math

import numpy as np


# --------------------------------------------------------------------------- window
def pyramid_window(width, height):
    """tiles.py:16-50.  Returns (W, Dc, De) -- the reference returns the 3-tuple (quirk Q4).

    Same float64 operation order as the reference so the window is bit-identical:
    Dc = distance to the tile centre, De = distance to the nearest edge (per axis, then min),
    W = alpha * De / (Dc + De) with alpha chosen so that mean(W) == 1.
    """
    ax = np.arange(width)
    ay = np.arange(height)
    half_w = width * 0.5
    half_h = height * 0.5

    dc_x = np.square(ax - half_w + 0.5)                      # :35
    dc_y = np.square(ay - half_h + 0.5)                      # :36
    Dc = np.sqrt(dc_x[:, None] + dc_y[None, :])              # :37

    quarter = np.square(0.5)
    e_left = np.square(ax - 0 + 0.5) + quarter               # :39
    e_right = np.square(ax - width + 0.5) + quarter          # :40
    e_bottom = quarter + np.square(ay - 0 + 0.5)             # :41
    e_top = quarter + np.square(ay - height + 0.5)           # :42
    de_x = np.sqrt(np.minimum(e_left, e_right))              # :44
    de_y = np.sqrt(np.minimum(e_bottom, e_top))              # :45
    De = np.minimum(de_x[:, None], de_y[None, :])            # :46

    ratio = np.divide(De, np.add(Dc, De))
    alpha = (width * height) / np.sum(ratio)                 # :48
    return alpha * ratio, Dc, De


def mean_window(tile_h, tile_w):
    """tiles.py:282-283 -- float32 ones."""
    return np.ones((tile_h, tile_w), dtype=np.float32)


# --------------------------------------------------------------------------- geometry
def _pair(v):
    if isinstance(v, (np.ndarray, list, tuple)):
        if len(v) != 2:
            raise ValueError("need exactly 2 elements")
        return int(v[0]), int(v[1])
    return int(v), int(v)


def slicer_geometry(image_shape, tile_size, tile_step, image_margin=0):
    """tiles.py:62-142.  Pure integer math; must be bit-exact.

    Returns dict(tile_size, tile_step, margins=(left,right,top,bottom), crops[N,4], bbox_crops[N,4], target_shape)
    with crops in (x, y, w, h) order, row-major (y outer, x inner).
    """
    img_h, img_w = int(image_shape[0]), int(image_shape[1])
    th, tw = _pair(tile_size)                                  # :74-79
    sh, sw = _pair(tile_step)                                  # :81-86
    if sh < 1 or sh > th or sw < 1 or sw > tw:                # :92-95
        raise ValueError()
    ov_h, ov_w = th - sh, tw - sw                              # :97

    if isinstance(image_margin, (list, tuple)) or image_margin != 0:
        if isinstance(image_margin, (list, tuple)):            # :118-122
            left, right, top, bottom = image_margin
        else:
            left = right = top = bottom = image_margin
    else:                                                      # :101-116 automatic margins
        n_w = max(1, math.ceil((img_w - ov_w) / sw))
        n_h = max(1, math.ceil((img_h - ov_h) / sh))
        extra_w = sw * n_w - (img_w - ov_w)
        extra_h = sh * n_h - (img_h - ov_h)
        left = extra_w // 2
        right = extra_w - left
        top = extra_h // 2
        bottom = extra_h - top

    crops, bbox = [], []
    for y in range(0, img_h + top + bottom - th + 1, sh):      # :132-139
        for x in range(0, img_w + left + right - tw + 1, sw):
            crops.append((x, y, tw, th))
            bbox.append((x - left, y - top, tw, th))
    return dict(
        tile_size=(th, tw),
        tile_step=(sh, sw),
        margins=(left, right, top, bottom),
        crops=np.array(crops),
        bbox_crops=np.array(bbox),
        target_shape=(img_h + top + bottom, img_w + left + right),  # :236-242
    )


# --------------------------------------------------------------------------- split
def _zero_pad(img, top, bottom, left, right, value=0):
    """cv2.copyMakeBorder(BORDER_CONSTANT) as used at tiles.py:161,182,220 == constant padding."""
    pad = [(top, bottom), (left, right)] + [(0, 0)] * (img.ndim - 2)
    return np.pad(img, pad, mode="constant", constant_values=value)


def split(image, geom, value=0):
    """tiles.py:177-204: pad the whole image by the margins, then cut every crop."""
    left, right, top, bottom = geom["margins"]
    padded = _zero_pad(image, top, bottom, left, right, value)
    return [padded[y:y + h, x:x + w] for (x, y, w, h) in geom["crops"]]


def cut_patch(image, geom, index, value=0):
    """tiles.py:206-234 (and iter_split :144-175): cut one tile lazily, padding only what hangs over."""
    x, y, w, h = (int(v) for v in geom["bbox_crops"][index])
    ih, iw = image.shape[0], image.shape[1]
    inside = image[max(y, 0):min(ih, y + h), max(x, 0):min(iw, x + w)]
    return _zero_pad(inside, max(0, -y), max(0, y + h - ih), max(0, -x), max(0, x + w - iw), value)


def crop_to_original(arr, geom, image_shape):
    """tiles.py:271-280 -- slices the first two axes (HWC layout)."""
    left, _right, top, _bottom = geom["margins"]
    return arr[top:top + image_shape[0], left:left + image_shape[1]]


# --------------------------------------------------------------------------- merging
def slicer_merge(tiles, geom, weight, image_shape, dtype=np.float32):
    """ImageSlicer.merge, tiles.py:244-269: float64 HWC accumulation, eps clamp, truncating cast, crop."""
    crops = geom["crops"]
    if len(tiles) != len(crops):
        raise ValueError
    ch = 1 if tiles[0].ndim == 2 else tiles[0].shape[2]
    H, W = geom["target_shape"]
    acc = np.zeros((H, W, ch), dtype=np.float64)
    nrm = np.zeros((H, W, ch), dtype=np.float64)
    w3 = np.repeat(np.asarray(weight)[:, :, None], ch, axis=2)          # dstack([w]*C), :258
    for t, (x, y, tw, th) in zip(tiles, crops):
        t3 = t if t.ndim == 3 else t[:, :, None]
        acc[y:y + th, x:x + tw] += t3 * w3                              # :262
        nrm[y:y + th, x:x + tw] += w3                                   # :263
    nrm = np.clip(nrm, np.finfo(np.float64).eps, None)                  # :266
    out = (acc / nrm).astype(dtype)                                     # :267 (truncates for ints, quirk Q6)
    return crop_to_original(out, geom, image_shape)


def merger_new(target_shape, channels, weight, dtype=np.float32):
    """TileMerger.__init__, tiles.py:295-308: weight -> [1,h,w] in accumulator dtype; zero accumulators."""
    H, W = target_shape
    return dict(
        weight=np.asarray(weight)[None].astype(dtype),
        image=np.zeros((channels, H, W), dtype=dtype),
        norm_mask=np.zeros((1, H, W), dtype=dtype),
    )


def merger_integrate(state, batch, coords):
    """TileMerger.integrate_batch, tiles.py:321-339: sequential  image[:,y:y+h,x:x+w] += tile*weight ; norm += weight
    in the accumulator dtype (fp32: one rounding for the product, one for the add)."""
    if len(batch) != len(coords):
        raise ValueError("Number of images in batch does not correspond to number of coordinates")
    img, nrm, w = state["image"], state["norm_mask"], state["weight"]
    batch = np.asarray(batch).astype(img.dtype, copy=False)                # :334-335
    for tile, (x, y, tw, th) in zip(batch, coords):
        x, y, tw, th = int(x), int(y), int(tw), int(th)
        img[:, y:y + th, x:x + tw] += tile * w                             # :338
        nrm[:, y:y + th, x:x + tw] += w                                    # :339
    return state


def merger_merge(state):
    """TileMerger.merge, tiles.py:345-346: plain division, NO eps clamp (uncovered -> NaN, quirk Q5)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        return state["image"] / state["norm_mask"]
