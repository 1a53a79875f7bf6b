"""Tiled inference on huge images: host-side slicing + MI355X-side weighted blending.

Drop-in for ``pytorch_toolbelt.inference.tiles`` (reference inference/tiles.py): same class / method / attribute
names and error behaviour.  ``ImageSlicer`` is host geometry (numpy, integer-exact); ``TileMerger(device="cuda")`` keeps its
accumulators in HBM and blends with the hand-written HIP kernels of ``libptb_hip.so`` -- there is no torch-op or
CPU fallback on that path.  ``TileMerger(device="cpu")`` -- the reference's default -- is what it says: accumulators on the
host in the caller's dtype, blended with torch ops (``inference/_host.py``); the device the caller names decides, nothing else.
"""
import math
import warnings
from typing import Iterable, List, Sequence, Tuple

import numpy as np
import torch

from .. import _native as N
from . import _lazy

__all__ = ["ImageSlicer", "TileMerger", "CudaTileMerger", "HostBackedTileMerger", "compute_pyramid_patch_weight_loss"]

# OpenCV border codes accepted by split/cut_patch (the reference forwards them to cv2.copyMakeBorder,
# inference/tiles.py:161,182,220).  Only constant padding is pinned by the oracle; the others map to the numpy
# mode with OpenCV's documented semantics.
BORDER_CONSTANT = 0
_FRESH_ROWS = 32  # rows of a first-touch block = default chunk rows of the view kernels (64 columns wide)
_NP_PAD_MODE = {1: "edge", 2: "symmetric", 3: "wrap", 4: "reflect"}


def compute_pyramid_patch_weight_loss(width: int, height: int):
    """Blending window: large in the tile centre, small at its border (reference inference/tiles.py:16-50).

    W = alpha * De / (Dc + De), Dc = distance to the centre, De = distance to the closest edge, alpha normalises the
    mean to 1.  Returns the triple ``(W, Dc, De)`` of float64 ``[width, height]`` arrays exactly like the reference
    (same operation order, so the window is bit-identical).
    """
    u = np.arange(width)
    v = np.arange(height)
    cu, cv = width * 0.5, height * 0.5
    Dc = np.sqrt(np.square(u - cu + 0.5)[:, None] + np.square(v - cv + 0.5)[None, :])

    q = np.square(0.5)
    eu = np.sqrt(np.minimum(np.square(u - 0 + 0.5) + q, np.square(u - width + 0.5) + q))
    ev = np.sqrt(np.minimum(q + np.square(v - 0 + 0.5), q + np.square(v - height + 0.5)))
    De = np.minimum(eu[:, None], ev[None, :])

    ratio = np.divide(De, np.add(Dc, De))
    alpha = (width * height) / np.sum(ratio)
    return alpha * ratio, Dc, De


def _two(value, what) -> Tuple[int, int]:
    if isinstance(value, (np.ndarray, Sequence)):
        if len(value) != 2:
            raise ValueError(f"{what} must have exactly 2 elements. Got: {what}={value}")
        return int(value[0]), int(value[1])
    return int(value), int(value)


def _pad2d(arr: np.ndarray, top, bottom, left, right, border_type, value) -> np.ndarray:
    widths = [(int(top), int(bottom)), (int(left), int(right))] + [(0, 0)] * (arr.ndim - 2)
    if border_type == BORDER_CONSTANT:
        return np.pad(arr, widths, mode="constant", constant_values=value)
    if border_type not in _NP_PAD_MODE:
        raise NotImplementedError(f"border_type={border_type} is not supported")
    return np.pad(arr, widths, mode=_NP_PAD_MODE[border_type])


class ImageSlicer:
    """Cut an image into overlapping tiles and (on the host, in float64) blend tiles back.

    Host-only and picklable (plain ints + ndarrays), so it can live in DataLoader workers.
    Attributes follow the reference (inference/tiles.py:62-142): ``image_height/width``, ``tile_size``,
    ``tile_step``, ``weight``, ``margin_left/right/top/bottom``, ``crops`` and ``bbox_crops`` as int64 ``[N, 4]``
    arrays of ``(x, y, width, height)``, row-major over the tile grid.
    """

    tile_size: Tuple[int, int]
    tile_step: Tuple[int, int]

    def __init__(self, image_shape: Tuple[int, int], tile_size, tile_step=0, image_margin=0, weight="mean"):
        self.image_height = image_shape[0]
        self.image_width = image_shape[1]
        self.tile_size = _two(tile_size, "tile_size")
        self.tile_step = _two(tile_step, "tile_step")

        if isinstance(weight, np.ndarray):
            self.weight = weight
        else:
            self.weight = {"mean": self._mean, "pyramid": self._pyramid}[weight](self.tile_size)  # KeyError if unknown

        (th, tw), (sh, sw) = self.tile_size, self.tile_step
        if not (1 <= sh <= th):
            raise ValueError()
        if not (1 <= sw <= tw):
            raise ValueError()

        if isinstance(image_margin, Sequence):
            ml, mr, mt, mb = image_margin
        elif image_margin == 0:
            # automatic margins: just enough padding for a whole number of steps, split evenly (extra pixel right/bottom)
            lap_h, lap_w = th - sh, tw - sw
            n_cols = max(1, math.ceil((self.image_width - lap_w) / sw))
            n_rows = max(1, math.ceil((self.image_height - lap_h) / sh))
            pad_w = sw * n_cols - (self.image_width - lap_w)
            pad_h = sh * n_rows - (self.image_height - lap_h)
            ml, mt = pad_w // 2, pad_h // 2
            mr, mb = pad_w - ml, pad_h - mt
        else:
            ml = mr = mt = mb = image_margin
        self.margin_left, self.margin_right, self.margin_top, self.margin_bottom = ml, mr, mt, mb

        padded_h = self.image_height + mt + mb
        padded_w = self.image_width + ml + mr
        ys = range(0, padded_h - th + 1, sh)
        xs = range(0, padded_w - tw + 1, sw)
        self.crops = np.array([(x, y, tw, th) for y in ys for x in xs])
        self.bbox_crops = np.array([(x - ml, y - mt, tw, th) for y in ys for x in xs])

    # ------------------------------------------------------------------ slicing
    def _check_shape(self, image, exc):
        if image.shape[0] != self.image_height or image.shape[1] != self.image_width:
            raise exc

    def _lazy_tile(self, image, box, border_type, value):
        x, y, w, h = (int(v) for v in box)
        H, W = image.shape[0], image.shape[1]
        tile = image[max(y, 0):min(H, y + h), max(x, 0):min(W, x + w)]
        if x < 0 or y < 0 or x + w > W or y + h > H:
            tile = _pad2d(tile, max(0, -y), max(0, y + h - H), max(0, -x), max(0, x + w - W), border_type, value)
        return tile

    def iter_split(self, image: np.ndarray, border_type=BORDER_CONSTANT, value=0) -> Iterable[Tuple[np.ndarray, np.ndarray]]:
        """Yield ``(tile, crops[i])`` lazily; only tiles hanging over the image border are padded (copied)."""
        self._check_shape(image, ValueError())
        for coords, box in zip(self.crops, self.bbox_crops):
            yield self._lazy_tile(image, box, border_type, value), coords

    def split(self, image, border_type=BORDER_CONSTANT, value=0) -> List[np.ndarray]:
        """Pad the whole image by the margins once, return the N tiles as views of the padded copy.

        ``border_type`` takes OpenCV's codes like the reference (which forwards them to ``cv2.copyMakeBorder``,
        inference/tiles.py:161-182).  ``BORDER_CONSTANT`` (0, the default) is pinned against the reference's outputs; the other
        codes are mapped to the numpy padding mode with the same extension rule (1 replicate -> "edge", 2 reflect -> "symmetric",
        3 wrap, 4 reflect_101 -> "reflect") and pinned to the tables of OpenCV's documentation (``aaaaaa|abcdefgh|hhhhhhh`` ...:
        tests/test_slicer_cpu.py) -- OpenCV itself is not available where the goldens are generated."""
        assert image.shape[0] == self.image_height
        assert image.shape[1] == self.image_width
        padded = _pad2d(image, self.margin_top, self.margin_bottom, self.margin_left, self.margin_right, border_type, value)
        tiles = []
        for x, y, w, h in self.crops:
            tile = padded[y:y + h, x:x + w]
            assert tile.shape[0] == self.tile_size[0]
            assert tile.shape[1] == self.tile_size[1]
            tiles.append(tile)
        return tiles

    def cut_patch(self, image: np.ndarray, slice_index, border_type=BORDER_CONSTANT, value=0):
        assert image.shape[0] == self.image_height
        assert image.shape[1] == self.image_width
        return self._lazy_tile(image, self.bbox_crops[slice_index], border_type, value)

    @property
    def target_shape(self):
        return (
            self.image_height + self.margin_bottom + self.margin_top,
            self.image_width + self.margin_right + self.margin_left,
        )

    # ------------------------------------------------------------------ host merge (float64)
    def merge(self, tiles: List[np.ndarray], dtype=np.float32):
        """Host blend in float64 (HWC), eps-clamped normalisation, ``astype(dtype)`` (truncating), crop to the image."""
        if len(tiles) != len(self.crops):
            raise ValueError
        channels = 1 if tiles[0].ndim == 2 else tiles[0].shape[2]
        full = self.target_shape + (channels,)
        total = np.zeros(full, dtype=np.float64)
        mass = np.zeros(full, dtype=np.float64)
        w = np.dstack([self.weight] * channels)
        for tile, (x, y, tw, th) in zip(tiles, self.crops):
            total[y:y + th, x:x + tw] += tile * w
            mass[y:y + th, x:x + tw] += w
        mass = np.clip(mass, a_min=np.finfo(mass.dtype).eps, a_max=None)
        return self.crop_to_orignal_size(np.divide(total, mass).astype(dtype))

    def crop_to_orignal_size(self, image):
        """Remove the margins from an ``[H', W', ...]`` array (sic: the misspelt name is the reference's API)."""
        assert image.shape[0] == self.target_shape[0]
        assert image.shape[1] == self.target_shape[1]
        crop = image[self.margin_top:self.image_height + self.margin_top, self.margin_left:self.image_width + self.margin_left]
        assert crop.shape[0] == self.image_height
        assert crop.shape[1] == self.image_width
        return crop

    # ------------------------------------------------------------------ device-side split
    def split_device(self, image: torch.Tensor, indices=None, augment=None, scale=None, bias=None, value: int = 0) -> torch.Tensor:
        """Model input for the tiles ``indices`` straight from a uint8 image that already lives in HBM.

        Equals ``torch.stack([image_to_tensor(t) for t in self.split(image)][indices]).float()`` (optionally
        ``* scale[c] + bias[c]`` and then ``tta.<augment>_image_augment``) -- the front of the reference's loop
        (README.md:209-216; tiles.py:177-204; utils/torch_utils.py:204-231) -- but as ONE HIP launch: no padded copy,
        no per-tile HWC->CHW copies, no fp32 upload.  ``image``: CUDA uint8 ``[H, W, C]`` or ``[H, W]``;
        ``indices``: None (all tiles), a slice, or a sequence of tile indices; ``augment``: None | "fliplr" | "flipud" |
        "flips" | "d2" | "d4"; ``scale`` / ``bias``: per-channel sequences (both or neither); ``value``: constant border.
        Returns fp32 ``[V*n, C, tile_h, tile_w]``, chunk-major like the augment functions.
        """
        from .tta import AUGMENT_VIEWS

        N.require_device(image, "ImageSlicer.split_device")
        if image.dtype != torch.uint8:
            raise NotImplementedError(f"split_device takes a uint8 image, got {image.dtype}")
        if image.dim() not in (2, 3) or image.shape[0] != self.image_height or image.shape[1] != self.image_width:
            raise ValueError(f"image of shape {tuple(image.shape)} does not match the slicer ({self.image_height}, {self.image_width})")
        if augment is not None and augment not in AUGMENT_VIEWS:
            raise KeyError(augment)
        views = list(AUGMENT_VIEWS[augment]) if augment is not None else [N.IDENT]
        th, tw = int(self.tile_size[0]), int(self.tile_size[1])
        if any(v & 1 for v in views) and th != tw:
            raise ValueError(f"Input tensor must have number of rows equal to number of cols. Got tiles of {th}x{tw}")
        channels = 1 if image.dim() == 2 else int(image.shape[2])
        if (scale is None) != (bias is None):
            raise ValueError("scale and bias go together")
        if indices is None:
            boxes = self.bbox_crops
        elif isinstance(indices, slice):
            boxes = self.bbox_crops[indices]
        else:
            boxes = self.bbox_crops[np.asarray(indices, dtype=np.int64).reshape(-1)]
        n = len(boxes)
        image = image.contiguous()
        out = torch.empty((len(views) * n, channels, th, tw), device=image.device, dtype=torch.float32)
        if n == 0:
            return out
        xy = np.ascontiguousarray(np.asarray(boxes, dtype=np.int64)[:, :2].T)
        fa = None
        if scale is not None:
            sc = np.ascontiguousarray(np.broadcast_to(np.asarray(scale, dtype=np.float32).reshape(-1), (channels,)))
            bi = np.ascontiguousarray(np.broadcast_to(np.asarray(bias, dtype=np.float32).reshape(-1), (channels,)))
            fa = (sc.ctypes.data_as(N._fp), bi.ctypes.data_as(N._fp))
        lib = N.load()
        with N.on_device(image.device):
            rc = lib.ptb_split_tiles_u8(image.data_ptr(), self.image_height, self.image_width, channels,
                                        xy[0].ctypes.data_as(N._i64p), xy[1].ctypes.data_as(N._i64p), n, th, tw,
                                        len(views), N.int_array(views), fa[0] if fa else None, fa[1] if fa else None,
                                        int(value), out.data_ptr(), N.stream_ptr(image.device))
        N.bump()
        N.check(rc, "ImageSlicer.split_device")
        return out

    def _mean(self, tile_size):
        return np.ones((tile_size[0], tile_size[1]), dtype=np.float32)

    def _pyramid(self, tile_size):
        return compute_pyramid_patch_weight_loss(tile_size[0], tile_size[1])[0]


def _coords_xy(crop_coords, n_expected=None):
    """crop_coords: ndarray [B,4], list of 4-sequences, or the CPU int64 tensor default_collate builds."""
    if torch.is_tensor(crop_coords):
        arr = crop_coords.detach().cpu().numpy()
    else:
        arr = np.asarray([[int(v) for v in c] for c in crop_coords] if not isinstance(crop_coords, np.ndarray) else crop_coords)
    arr = np.ascontiguousarray(arr, dtype=np.int64).reshape(-1, 4)
    return arr


_warned = set()


def _warn_once(key, message):
    if key not in _warned:
        _warned.add(key)
        warnings.warn(message, RuntimeWarning, stacklevel=3)


def _resolve_device(device, what):
    """The device of a HIP-backed merger: a CUDA device, or an error.  (``TileMerger(device="cpu")`` / ``VolumeMerger(device="cpu")``
    never get here: they are the torch-op mergers, like the reference's.)"""
    device = torch.device(device)
    if device.type == "cuda":
        return device
    raise RuntimeError(
        f"{what}(device='{device}'): this merger keeps its accumulators in MI355X HBM and has no CPU "
        "path; construct it with device='cuda' on a machine with a GPU."
    )


# ------------------------------------------------------------------------------------------------ self-planning mergers
# The reference's loop builds `TileMerger(tiler.target_shape, C, tiler.weight)` -- no crop list -- for every image and feeds it the
# same crops in the same order (README.md:201-226).  A merger without `crops=` therefore records the crop sequence it saw
# (at `merge()`), and the NEXT merger of the same geometry and window (or the same one after `reset()`) plans itself from it:
# normaliser precomputed, every block divided in the launch that brings its last tile, no separate merge pass.  Any deviation
# from the remembered sequence drops back to the ordinary path for the rest of that image (bit-identical results either way).
# Opt-in since round 4 (PTB_AUTO_PLAN=1 / set_auto_plan(True)): with the lazy de-augmentation handle already fusing the two reference
# calls into one launch, and self-planned mergers keeping their accumulators exact (one more store of the image), planning from the
# previous image buys ~0.5 % over the ordinary fused path at the headline geometry -- not worth module-level caches by default.
_AUTO_PLAN = __import__("os").environ.get("PTB_AUTO_PLAN", "0") == "1"
_AUTO_MAX = 8            # geometries remembered (each keeps a [1, H', W'] normaliser in HBM once planned)
_auto = __import__("collections").OrderedDict()   # key -> _AutoEntry
_auto_lock = __import__("threading").RLock()   # mergers of several threads (one inference loop each) share the cache


def set_auto_plan(flag: bool) -> bool:
    """Switch self-planning of ``TileMerger`` without ``crops=`` on / off (default off; ``PTB_AUTO_PLAN=1``); returns the previous setting."""
    global _AUTO_PLAN
    prev, _AUTO_PLAN = _AUTO_PLAN, bool(flag)
    return prev


def _weight_signature(weight: np.ndarray):
    """Content signature of a blending window: shape, dtype and a hash of ALL its bytes, computed on every call (a
    cache keyed by the array's identity plus a sample of its values returned a stale signature -- and with it a stale device copy
    of the window -- after an in-place edit at positions the sample missed).  xxh3-128 runs at ~10 GB/s: 0.2 ms for the 2 MB of a
    512 x 512 float64 window; blake2b (~1.5 ms) when the xxhash module is missing."""
    data = np.ascontiguousarray(weight)
    try:
        import xxhash

        digest = xxhash.xxh3_128_digest(data)
    except ImportError:
        import hashlib

        digest = hashlib.blake2b(data.tobytes(), digest_size=16).digest()
    return (weight.shape, weight.dtype.str, digest)


_device_windows = __import__("collections").OrderedDict()   # (device index, window signature) -> float32 [1, h, w] device tensor


def _device_window(weight: np.ndarray, device):
    """The blending window as a float32 ``[1, h, w]`` device tensor.  A merger per image (the README loop) would otherwise upload
    the same window from pageable host memory every time -- a synchronous copy that also waits for the GPU to drain; uploaded
    windows are kept per device (a handful of MB) and every merger gets its own device-side copy of the cached one."""
    key = (device.index if device.index is not None else torch.cuda.current_device(),) + _weight_signature(weight)
    with _auto_lock:
        ent = _device_windows.get(key)
        if ent is None:
            cached = torch.from_numpy(np.expand_dims(weight, axis=0)).to(device=device, dtype=torch.float32).contiguous()
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(device))
            while len(_device_windows) >= 16:
                _device_windows.popitem(last=False)
            ent = _device_windows[key] = (cached, ready)
        else:
            _device_windows.move_to_end(key)
    cached, ready = ent
    torch.cuda.current_stream(device).wait_event(ready)      # (the upload may have been issued on another thread's stream)
    return cached.clone()


class _AutoEntry:
    __slots__ = ("log", "seen", "need", "parts", "disabled")

    def __init__(self):
        self.log = None        # bytes of the [n, 4] int64 crop sequence of the last merged image
        self.seen = 0          # consecutive merged images that ended with exactly this sequence
        self.need = 1          # repeats required before planning (grows when a planned image deviated)
        self.parts = None      # (xy, remaining0, norm_full, crops4) shared by the mergers planned from `log`
        self.disabled = False  # this geometry cannot be planned / its user reads accumulators or merges partially


def _auto_entry(key, create=False):
    with _auto_lock:
        ent = _auto.get(key)
        if ent is None and create:
            while len(_auto) >= _AUTO_MAX:
                _auto.popitem(last=False)
            ent = _auto[key] = _AutoEntry()
        elif ent is not None:
            _auto.move_to_end(key)
        return ent


class _Plan:
    """State of a *planned* TileMerger (constructed with the complete ``crops`` of the image).

    ``remaining[b]`` = planned tiles that have not touched accumulator block ``b`` (64 columns x 32 rows) yet,
    ``done[b]`` = the block has been written to the merge result.  Finalisation is only performed while the integrate
    calls follow the planned sequence exactly (then every partial sum, including the normaliser's, has the reference's
    order of additions); the first deviating batch switches it off for the rest of the image and everything not yet
    finalised goes through the ordinary accumulate + merge.  Restrictions while planned blocks are finalised: a tile
    that touches a finished block raises, ``merger.image`` is not readable (the accumulators of finished blocks are
    never stored) and ``merge_()`` is unavailable."""

    def __init__(self, xy, remaining0, norm_full):
        self.xy = xy                    # [2, N] int64 origins in integration order
        self.remaining0 = remaining0
        self.norm_full = norm_full      # [1, H, W] complete normaliser (device)
        self.remaining = remaining0.copy()
        self.done = np.zeros_like(remaining0)
        self.pos = 0
        self.active = True

    @staticmethod
    def build(merger, crops):
        crops = _coords_xy(crops)
        th, tw = int(merger.weight.shape[1]), int(merger.weight.shape[2])
        H, W = merger.image_height, merger.image_width
        if len(crops) == 0 or np.any(crops[:, 2] != tw) or np.any(crops[:, 3] != th):
            return None
        aligned = (tw % 64 == 0 and th % _FRESH_ROWS == 0 and not np.any(crops[:, 0] % 64) and not np.any(crops[:, 1] % _FRESH_ROWS)
                   and np.all(crops[:, 0] >= 0) and np.all(crops[:, 1] >= 0) and np.all(crops[:, 0] + tw <= W) and np.all(crops[:, 1] + th <= H))
        if not aligned:
            return None   # geometry off the block grid: the ordinary path is used
        remaining = np.zeros(((H + _FRESH_ROWS - 1) // _FRESH_ROWS, (W + 63) // 64), dtype=np.int32)
        for x, y in crops[:, :2]:
            remaining[y // _FRESH_ROWS:(y + th) // _FRESH_ROWS, x // 64:(x + tw) // 64] += 1
        if remaining.max() > 255:
            return None
        xy = np.ascontiguousarray(crops[:, :2].T)
        norm_full = torch.zeros((1, H, W), device=merger.weight.device, dtype=torch.float32)
        lib = N.load()
        dev = norm_full.device
        with N.on_device(dev):
            rc = lib.ptb_norm_accumulate(norm_full.data_ptr(), merger.weight.data_ptr(), xy[0].ctypes.data_as(N._i64p),
                                         xy[1].ctypes.data_as(N._i64p), xy.shape[1], th, tw, H, W, None, 0, N.stream_ptr(dev))
        N.bump()
        N.check(rc, "TileMerger(crops=...)")
        plan = _Plan(xy, remaining.astype(np.uint8), norm_full)
        plan.crops4 = np.ascontiguousarray(crops, dtype=np.int64)     # the planned (x, y, w, h) rows, for the deferred fast path
        return plan

    def restart(self):
        self.remaining = self.remaining0.copy()
        self.done[:] = 0
        self.pos = 0
        self.active = True

    def touches_done(self, xy, th, tw):
        for x, y in xy.T:
            if self.done[y // _FRESH_ROWS:(y + th + _FRESH_ROWS - 1) // _FRESH_ROWS, x // 64:(x + tw + 63) // 64].any():
                return True
        return False


def _tensor_version(t):
    try:
        return t._version
    except RuntimeError:      # inference tensors carry no version counter: in-place edits of them cannot be seen
        return None


def _held_entry(batch):
    """(first byte, one past the last byte, version counter) of a batch a deferred merger is about to keep a reference to."""
    p0 = batch.data_ptr()
    return p0, p0 + batch.numel() * batch.element_size(), _tensor_version(batch)


def _check_held(held, batch, span, launches, what):
    """The contract of deferred merging, enforced: a held batch is read by a LATER launch, so (1) a new batch must not live in
    the memory of one that is still held -- a model writing into a static output buffer (HIP graphs, ``out=``, preallocated
    outputs) has then already overwritten data the merger has not read, which no fallback can bring back -- and (2) a held batch
    must not have been modified in place since it was handed in (checked when its launch is due).  ``held`` rows end with
    (p0, p1, version); ``span`` = ``_held_entry(batch)``."""
    p0, p1, _v = span
    for h in held:
        if h[-3] < p1 and p0 < h[-2]:
            raise RuntimeError(f"{what}: this batch occupies memory of an earlier batch that is still held for a later launch (bytes "
                               f"{max(p0, h[-3]):#x}..{min(p1, h[-2]):#x}) -- the model writes its outputs into a reused buffer, so the earlier "
                               "predictions are already gone.  Deferred merging needs every batch to stay alive and unmodified until its rows "
                               "are merged: hand over fresh tensors (or clones), or construct the merger without defer=True.")
    if launches:
        for i, h in enumerate(held):
            if h[-1] is not None and _tensor_version(h[0]) != h[-1]:
                raise RuntimeError(f"{what}: held batch {i} of the rows about to be merged was modified in place after it was handed to the "
                                   "merger (its version counter moved).  Deferred merging reads the batches later: keep them unmodified, "
                                   "or construct the merger without defer=True.")


def _defer_rows_default():
    import os

    return int(os.environ.get("PTB_DEFER_ROWS", "1024"))


class _Bands:
    """Deferred planned merging (``TileMerger(..., crops=tiler.crops, defer=True)``), planned once and driven from C
    (``ptb_band_plan_*``, csrc/ptb_bandplan.hip).

    A *band* is the rows between two consecutive tile edges; every tile that touches a band covers all of its rows.  Consecutive
    bands form a *launch group* of about ``rows`` rows (default 1024; ``defer_rows=`` / ``PTB_DEFER_ROWS``).  The merger only keeps
    references to the model outputs it is handed, and when the last tile of a group has arrived ONE launch reads all covering
    tiles of its rows, de-augments, reduces, blends in integration order and writes ``sum / norm`` to the result -- the
    accumulator image never travels through HBM (the incremental path re-reads and re-writes every pixel once per overlapping
    tile row).  The fp32 operation order per pixel is the incremental path's, so the result is bit-identical.  Cost: the batches
    of the last ``rows / step + 1`` tile rows stay alive until their group is done, and they must not be modified in place in the
    meantime -- which is why this is opt-in.

    Until the first group is launched any deviation from the plan simply replays the held batches through the incremental path;
    afterwards ``merger.image``, a partial ``merge()`` or an unplanned tile raise."""

    def __init__(self, handle, table, bands, n_bands, last_group, monotone):
        self.handle = handle            # ptb_band_plan*
        self.table = table              # uint8 device tensor holding the work-item table (owned here)
        self.bands = bands              # [(y0, y1, last tile)] per launch group, top to bottom
        self.n_bands = n_bands
        self.last_group = last_group    # plan index of a tile -> the last launch group that reads it
        self.monotone = monotone        # groups complete in index order (row-major crops): batches can be released early

    def __del__(self):
        try:
            if self.handle:
                N.load().ptb_band_plan_destroy(self.handle)
                self.handle = None
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    @staticmethod
    def build(plan, channels, th, tw, H, W, device, rows):
        import ctypes

        lib = N.load()
        handle = ctypes.c_void_p()
        n = plan.xy.shape[1]
        nbytes = lib.ptb_band_plan_create(plan.xy[0].ctypes.data_as(N._i64p), plan.xy[1].ctypes.data_as(N._i64p), n, channels, th, tw, H, W,
                                          int(rows), 0, H, None, 0, ctypes.byref(handle))
        if nbytes < 0:
            return None
        table = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)
        with N.on_device(device):
            rc = lib.ptb_band_plan_upload(handle, table.data_ptr(), N.stream_ptr(device))
        N.bump()
        N.check(rc, "TileMerger(defer=True)")
        ng, nb, ni = ctypes.c_int(), ctypes.c_int(), ctypes.c_int64()
        lib.ptb_band_plan_info(handle, ctypes.byref(ng), ctypes.byref(nb), ctypes.byref(ni), None, None)
        last_group = np.zeros(n, dtype=np.int64)
        rows_arr = np.zeros(3 * ng.value, dtype=np.int64)
        lib.ptb_band_plan_info(handle, None, None, None, last_group.ctypes.data_as(N._i64p), rows_arr.ctypes.data_as(N._i64p))
        groups = [tuple(int(v) for v in rows_arr[3 * g:3 * g + 3]) for g in range(ng.value)]
        lasts = [g[2] for g in groups]
        return _Bands(handle, table, groups, nb.value, last_group, all(a <= b for a, b in zip(lasts, lasts[1:])))


class TileMerger:
    """Blend tile predictions into a full-size map that lives in HBM (reference inference/tiles.py:290-350).

    ``image`` ``[C, H', W']``, ``norm_mask`` ``[1, H', W']`` and ``weight`` ``[1, h, w]`` are public fp32 tensors on the
    GPU (``image`` / ``norm_mask`` are properties: blocks no kernel has written yet are zero-filled on first read).  ``integrate_batch`` is one HIP launch per batch: overlapping tiles are accumulated race-free in batch order,
    bit-identical to the reference's sequential ``+=`` loop.  ``integrate_batch_deaugment`` additionally fuses the TTA
    de-augmentation (``tta.*_image_deaugment``) so the reduced tile never travels through HBM.
    """

    def __new__(cls, image_shape=None, channels=None, weight=None, device="cpu", *args, **kwargs):
        # the device the caller names decides the implementation: "cpu" (the reference's default) -> accumulators on the host,
        # torch ops (HostBackedTileMerger below); "cuda" -> HBM + HIP kernels (this class).  Never the other way round.
        if cls is TileMerger:
            dtype = kwargs.get("dtype", args[0] if args else torch.float32)
            # float64 accumulators (reference tiles.py:306-308 accumulates in the caller's dtype): the HIP kernels sum in float32, so a
            # float64 merger -- on any device -- is the torch-op one, whose sums ARE float64 (said once for CUDA devices).
            if torch.device(device).type != "cuda" or dtype == torch.float64:
                return object.__new__(HostBackedTileMerger)
        return object.__new__(cls)

    def __init__(self, image_shape, channels, weight, device="cpu", dtype=torch.float32, crops=None, defer=False, defer_rows=None,
                 auto_plan=None):
        """``crops`` (extension, optional): the complete crop list the image will receive (``tiler.crops``), in the
        order it will be integrated.  With it the merger runs *planned*: the normaliser is known up front and every
        block of the image is divided by it in the very launch that brings its last tile, so ``merge()`` has nothing
        left to do.  Results are bit-identical; see ``_Plan`` for what a planned merger restricts.

        ``defer=True`` (with ``crops``): *deferred* planned merging -- the merger holds on to the batches and merges a
        horizontal band of the image in one launch as soon as all its tiles are in, without an accumulator in HBM; see
        ``_Bands`` (the batches must stay unmodified until then).  ``defer_rows``: rows merged per launch (default 1024).

        Without ``crops`` the merger can plan itself (``auto_plan``, opt-in: ``set_auto_plan(True)`` / ``PTB_AUTO_PLAN=1``): the crop
        sequence an image ended with at ``merge()`` is remembered per geometry + window, and the next merger of that geometry
        (or this one after ``reset()``) runs planned from it -- the reference's per-image ``TileMerger(shape, C, weight)``
        gets the planned kernels from the second image on.  A deviating batch, a read of ``image`` / ``norm_mask`` or
        ``merge_()`` drop back to the ordinary path, bit-exactly (a self-planned merger keeps its accumulators complete: ``_unfinalise``)."""
        device = _resolve_device(device, "TileMerger")
        # The reference keeps image / norm_mask / weight in `dtype` (tiles.py:295-308) and so accumulates in it.  Here the accumulators
        # are always float32 (what the kernels read-modify-write); any other floating dtype is honoured at the boundary: tile batches
        # of that dtype are read as they are, and merge() / image / norm_mask hand out tensors of that dtype.  For float16 / bfloat16
        # that is a strictly more accurate sum than the reference's.  (float64 never gets here: __new__ hands it to the torch-op merger.)
        if dtype not in (torch.float32, torch.float16, torch.bfloat16, torch.float64):
            raise TypeError(f"TileMerger: dtype must be a floating point type, got {dtype}")
        self.dtype = dtype
        dtype = torch.float32
        N.load()
        self.image_height = image_shape[0]
        self.image_width = image_shape[1]
        self.channels = channels
        if isinstance(weight, np.ndarray):
            self.weight = _device_window(weight, device)
        else:
            self.weight = torch.from_numpy(np.expand_dims(weight, axis=0)).to(device=device, dtype=dtype).contiguous()
        # First-touch accumulators: allocated uninitialised; `_fresh` (host, one byte per 64x32 block) records which
        # blocks were never written.  The kernels STORE into fresh blocks instead of read-modify-write, so no memset and
        # no read of zeros is ever paid; reading `image` / `norm_mask` zero-fills whatever is still fresh first.
        self._image = torch.empty((channels, self.image_height, self.image_width), device=device, dtype=dtype)
        self._norm = torch.empty((1, self.image_height, self.image_width), device=device, dtype=dtype)
        self._fresh = np.ones(((self.image_height + _FRESH_ROWS - 1) // _FRESH_ROWS, (self.image_width + 63) // 64), dtype=np.uint8)
        # Lazy normaliser: `norm_mask` depends only on the crop list and the window, never on the predictions.  The
        # accumulate kernels therefore skip it (norm = NULL); the crops are logged on the host and the normaliser is
        # materialised when somebody needs it (`merge`, or a read of `norm_mask`) -- and when the log of an image equals
        # the log the buffer was last built from (every image of one slicer), it is simply reused.
        self._log = []            # [2, B] int64 origin arrays of the integrate calls since the last reset
        self._applied = 0         # log entries already summed into _norm
        self._norm_zero = True    # _norm is logically zero (its buffer may hold the cached normaliser of _norm_key)
        self._norm_key = None     # crop log the buffer content was built from, start to end, by this object alone
        self._norm_pure = False   # this cycle's _norm was built from zero by _norm_ready only
        self._eager_norm = False  # norm_mask was handed out: keep it up to date inside the accumulate kernels
        self._merged = None       # planned mode: the merge result the accumulate launches fill in
        self._plan = _Plan.build(self, crops) if crops is not None else None
        self._auto_key = None     # self-planning: key of this geometry + window in the module cache
        self._auto_planned = False
        self._auto_noted = None   # log length at the last merge() of this image
        self._weight_version0, self._weight_ptr0 = _tensor_version(self.weight), self.weight.data_ptr()
        if crops is None and (auto_plan if auto_plan is not None else _AUTO_PLAN) and isinstance(weight, np.ndarray):
            self._auto_key = (device.index if device.index is not None else torch.cuda.current_device(), int(self.image_height),
                              int(self.image_width)) + _weight_signature(weight)
            self._auto_attach()
        self._bands = None
        self._fast_cache = {}
        self.fast_submits = 0     # deferred batches that took the cached host path (diagnostic)
        if crops is not None and self._plan is None:
            _warn_once(("plan", tuple(self.weight.shape), self.image_height, self.image_width),
                       "TileMerger(crops=...): this geometry is off the 64 x 32 block grid of the planned kernels (tile size / origins); "
                       "the ordinary accumulate + merge path is used (same results, one more pass).")
        if defer and self._plan is not None:
            self._bands = _Bands.build(self._plan, channels, int(self.weight.shape[1]), int(self.weight.shape[2]), self.image_height,
                                       self.image_width, device, defer_rows if defer_rows is not None else _defer_rows_default())
            if self._bands is None:
                _warn_once(("defer", tuple(self.weight.shape), self.image_height, self.image_width),
                           "TileMerger(defer=True): the deferred band kernel does not take this geometry (tile origins, tile size or image width "
                           "off the 4-pixel grid, more than 224 tiles per launch group or more than 4 tiles over a pixel); the planned incremental "
                           "path is used (same results, slower).")
        elif defer:
            _warn_once(("defer-noplan",), "TileMerger(defer=True) needs the complete crop list (crops=tiler.crops) on the planned block "
                                          "grid; the ordinary path is used.")
        self._defer_reset()

    # ------------------------------------------------------------------ deferred bands
    def _defer_reset(self):
        self._defer_active = self._bands is not None
        self._held = []          # [batch tensor, coords, views, reduction, last launch group that reads it, p0, p1, version], integration order
        self._bands_done = 0     # launch groups issued for this image
        if self._bands is not None:
            N.load().ptb_band_plan_reset(self._bands.handle)

    def _defer_flush(self, what, keep_plan=False):
        """Leave deferred mode: replay the held batches through the incremental path (only before the first band) --
        the planned one when ``keep_plan`` (nothing deviated from the plan), else the ordinary one."""
        if not self._defer_active:
            return
        if self._bands_done:
            raise RuntimeError(f"TileMerger(defer=True): {what} is not available after bands of the image were merged; "
                               "integrate the planned tiles and call merge(), or construct the merger without defer=True")
        held, self._held = self._held, []
        self._defer_active = False
        self._plan.restart()
        self._plan.active = keep_plan
        self._log, self._applied = [], 0
        for batch, coords, views, reduction, *_rest in held:
            self._accumulate(batch, coords, views, reduction)

    def _launch_due(self, end):
        """Will a submit that brings the planned tiles up to index ``end`` (exclusive) launch a group?  (Only then are the held
        batches' version counters compared; groups of a row-major crop list complete in index order.)"""
        bands = self._bands
        if not bands.monotone:
            return True
        return self._bands_done < len(bands.bands) and end > bands.bands[self._bands_done][2]

    def _window_edited(self):
        w = self.weight
        return _tensor_version(w) != self._weight_version0 or w.data_ptr() != self._weight_ptr0

    def _defer_step(self, batch, coords, xy, views, reduction, dcode):
        """Take one planned batch into custody and merge the launch groups it completes (``ptb_band_plan_submit``: the pointer
        bookkeeping and the launches happen in C).  False: not deferrable (the caller goes on with the incremental path, after
        the held batches were replayed)."""
        plan, bands = self._plan, self._bands
        B = xy.shape[1]
        pos = plan.pos
        ok = (plan.active and not self._eager_norm and pos + B <= plan.xy.shape[1]
              and xy[0].data == plan.xy[0, pos:pos + B].data and xy[1].data == plan.xy[1, pos:pos + B].data)
        rc = N.PTB_EUNSUPPORTED
        if ok:
            span = _held_entry(batch)
            _check_held(self._held, batch, span, self._launch_due(pos + B), "TileMerger(defer=True)")
            if self._merged is None:
                self._merged = torch.empty_like(self._image)
            th, tw = int(self.weight.shape[1]), int(self.weight.shape[2])
            per_tile = self.channels * th * tw
            varr = N.int_array(views) if views is not None else N.int_array([N.IDENT])
            dev = self._image.device
            with N.on_device(dev):
                rc = N.load().ptb_band_plan_submit(bands.handle, pos, B, batch.data_ptr(), per_tile, B * per_tile, dcode,
                                                   len(views) if views is not None else 1, varr, reduction, self._merged.data_ptr(),
                                                   plan.norm_full.data_ptr(), self.weight.data_ptr(), N.stream_ptr(dev))
            N.bump()
        if rc == N.PTB_EUNSUPPORTED:
            _warn_once(("defer-deviation",), "TileMerger(defer=True): a batch deviates from the planned crop sequence / configuration (or "
                                             "norm_mask was read); leaving deferred mode for this image, the held batches are replayed incrementally.")
            self._defer_flush("an unplanned tile batch")
            return False
        if rc < 0:
            N.check(rc, "TileMerger.integrate_batch (deferred bands)")
        self._held.append((batch, coords, views, reduction, int(bands.last_group[pos:pos + B].max())) + span)
        plan.pos += B
        self._log.append(xy)
        if rc:
            self._bands_done += rc
            if self._bands_done == len(bands.bands):
                self._held.clear()
            elif bands.monotone:   # groups 0 .. done-1 are out: batches no later group reads can go
                done = self._bands_done
                while self._held and self._held[0][4] < done:
                    self._held.pop(0)
        return True

    # ------------------------------------------------------------------ first-touch state
    def _plan_off(self, what):
        """Leave planned mode; impossible once blocks were finalised (their accumulators were never stored)."""
        self._defer_flush(what)
        if self._plan is not None:
            if self._plan.done.any():
                if not self._auto_planned:
                    raise RuntimeError(f"TileMerger(crops=...): {what} is not available after planned blocks were finalised; "
                                       "call merge(), or construct the merger without crops=")
                self._unfinalise(what)
            self._plan.active = False
        self._auto_opt_out()

    # ------------------------------------------------------------------ self-planning (no crops= given)
    def _auto_attach(self):
        """(Re)plan this merger from the crop sequence its geometry ended the last image(s) with, when there is a stable one."""
        ent = _auto_entry(self._auto_key)
        usable = (ent is not None and not ent.disabled and ent.log is not None and ent.seen >= ent.need and not self._window_edited())
        if not usable:
            if self._auto_planned:
                self._plan, self._auto_planned = None, False
            return
        if self._auto_planned and ent.parts is not None and ent.parts[0] is self._plan.xy:
            self._plan.restart()
            return
        if ent.parts is None:
            plan = _Plan.build(self, np.frombuffer(ent.log, dtype=np.int64).reshape(-1, 4))
            if plan is None:          # off the block grid: this geometry never plans
                ent.disabled = True
                self._plan, self._auto_planned = None, False
                return
            built = torch.cuda.Event()
            built.record(torch.cuda.current_stream(plan.norm_full.device))
            ent.parts = (plan.xy, plan.remaining0, plan.norm_full, plan.crops4, built)
        xy, remaining0, norm_full, crops4, built = ent.parts
        torch.cuda.current_stream(norm_full.device).wait_event(built)      # (the normaliser may have been built on another stream)
        plan = _Plan(xy, remaining0, norm_full)
        plan.crops4 = crops4
        self._plan, self._auto_planned = plan, True

    def _auto_opt_out(self):
        """The caller touched the accumulators themselves: this geometry stays on the ordinary (exact, unplanned) path from now on."""
        if self._auto_key is not None:
            _auto_entry(self._auto_key, create=True).disabled = True

    def _auto_note(self):
        """At merge(): remember the crop sequence this image was made of (what the next image of this geometry is planned from)."""
        if self._auto_key is None:
            return
        n = len(self._log)
        if self._auto_noted == n:
            return
        ent = _auto_entry(self._auto_key, create=True)
        if self._auto_noted is not None or self._eager_norm or self._window_edited():
            ent.disabled = True       # tiles after a merge() / a caller-visible norm_mask / an edited window: not the README loop
            return
        self._auto_noted = n
        if n == 0:
            return
        plan = self._plan
        if self._auto_planned and plan.active and plan.pos == plan.xy.shape[1] and ent.parts is not None and ent.parts[0] is plan.xy:
            ent.seen += 1             # the planned sequence, start to end
            return
        xy = np.concatenate(self._log, axis=1)
        crops4 = np.empty((xy.shape[1], 4), dtype=np.int64)
        crops4[:, 0], crops4[:, 1] = xy[0], xy[1]
        crops4[:, 2], crops4[:, 3] = int(self.weight.shape[2]), int(self.weight.shape[1])
        log = crops4.tobytes()
        if self._auto_planned:        # a planned image that went another way: ask for more evidence before planning again
            ent.need = min(ent.need + 1, 4)
        if ent.log == log:
            ent.seen += 1
        else:
            ent.log, ent.seen, ent.parts = log, 1, None

    def _unfinalise(self, what):
        """Self-planned merger only: somebody needs the accumulators of blocks the planned kernels have already turned into results.
        A merger that planned ITSELF stores the weighted sum of a block next to its merged value (PTB_PLANNED_KEEP_SUMS: one more
        store of the image per image), so the accumulators are complete and exact -- the same bits the unplanned kernels would have
        left; the merger simply goes back to the ordinary path and its geometry stops planning itself."""
        plan = self._plan
        _warn_once(("unfinalise", self._auto_key), f"TileMerger: {what} after the self-planned kernels had finalised part of the image; the "
                                                   "accumulators are complete (self-planned mergers keep them), mergers of this geometry use the "
                                                   "ordinary accumulate + merge path from now on (TileMerger(..., auto_plan=False) avoids the switch).")
        plan.done[:] = 0
        plan.active = False
        self._auto_opt_out()

    @property
    def image(self) -> torch.Tensor:
        """``[C, H', W']`` accumulator (zeros where nothing was integrated yet)."""
        self._plan_off("reading .image")
        self._materialize()
        return self._image if self.dtype == torch.float32 else self._image.to(self.dtype)   # (a copy: write through integrate_* only)

    @image.setter
    def image(self, value: torch.Tensor):
        self._plan_off("assigning .image")
        self._materialize()
        self._image = value.to(device=self._image.device, dtype=torch.float32).contiguous()

    @property
    def norm_mask(self) -> torch.Tensor:
        """``[1, H', W']`` sum of the blending windows (materialised on first access; from then on the accumulate
        kernels keep this tensor up to date, exactly like the reference's attribute)."""
        if self._plan is not None:
            self._plan.active = False   # (finalised blocks keep their results; the rest accumulates with this norm)
        self._auto_opt_out()
        self._norm_ready()
        self._materialize()
        self._eager_norm, self._norm_pure, self._norm_key = True, False, None
        return self._norm if self.dtype == torch.float32 else self._norm.to(self.dtype)

    @norm_mask.setter
    def norm_mask(self, value: torch.Tensor):
        if self._plan is not None:
            self._plan.active = False
        self._auto_opt_out()
        self._norm_ready()
        self._materialize()
        self._eager_norm, self._norm_pure, self._norm_key = True, False, None
        self._norm = value

    def reset(self):
        """Start a new image: the accumulators become logically zero again (no memset: O(1) on the host)."""
        self._fresh[:] = 1
        self._log, self._applied = [], 0
        self._norm_zero, self._norm_pure, self._eager_norm = True, False, False
        self._merged = None
        self._auto_noted = None
        if self._auto_key is not None:
            self._auto_attach()       # plan from what the last image(s) looked like / restart / drop a plan that no longer holds
        elif self._plan is not None:
            self._plan.restart()
        self._defer_reset()

    def _log_key(self):
        return (self.weight.data_ptr(), _tensor_version(self.weight), b"".join(a.tobytes() for a in self._log))

    def _norm_ready(self):
        """Bring ``_norm`` up to date with the crop log (a no-op in eager mode, where the kernels maintain it)."""
        if self._eager_norm:
            return
        if self._norm_zero:
            if self._applied == 0 and self._norm_key is not None and self._norm_key == self._log_key():
                self._applied, self._norm_zero, self._norm_pure = len(self._log), False, True   # same crops as last image
                return
            self._norm.zero_()
            self._norm_zero, self._norm_pure, self._norm_key = False, True, None
        pending = self._log[self._applied:]
        if pending:
            xy = np.ascontiguousarray(np.concatenate(pending, axis=1))
            th, tw = int(self.weight.shape[1]), int(self.weight.shape[2])
            lib = N.load()
            dev = self._norm.device
            with N.on_device(dev):
                rc = lib.ptb_norm_accumulate(self._norm.data_ptr(), self.weight.data_ptr(), xy[0].ctypes.data_as(N._i64p),
                                             xy[1].ctypes.data_as(N._i64p), xy.shape[1], th, tw, self.image_height,
                                             self.image_width, None, 0, N.stream_ptr(dev))
            N.bump()
            N.check(rc, "TileMerger.norm_mask")
            self._applied = len(self._log)
        if self._norm_pure:
            self._norm_key = self._log_key()

    def _materialize(self):
        """Zero-fill the blocks no kernel has written yet, so the tensors read as plain zero-initialised accumulators."""
        fresh = self._fresh
        if not fresh.any():
            return
        if fresh.all():
            self._image.zero_()
            if self._eager_norm:   # lazy mode: the normaliser is not tied to the image's first-touch bitmap
                self._norm.zero_()
            fresh[:] = 0
        else:
            self._zero_fresh(0, self.image_height, 0, self.image_width)

    def _zero_fresh(self, y0, y1, x0, x1):
        """Zero-fill the never-written blocks that intersect rows y0:y1, columns x0:x1 (and mark them written)."""
        fresh = self._fresh[y0 // _FRESH_ROWS:(y1 + _FRESH_ROWS - 1) // _FRESH_ROWS, x0 // 64:(x1 + 63) // 64]
        if not fresh.any():
            return
        by, bx = y0 // _FRESH_ROWS, x0 // 64
        eager = self._eager_norm
        rows = np.nonzero(fresh.any(axis=1))[0]
        i = 0
        while i < len(rows):  # group consecutive block rows with identical column patterns into rectangles
            j = i
            while j + 1 < len(rows) and rows[j + 1] == rows[j] + 1 and np.array_equal(fresh[rows[j + 1]], fresh[rows[i]]):
                j += 1
            r0, r1 = (int(rows[i]) + by) * _FRESH_ROWS, min(self.image_height, (int(rows[j]) + by + 1) * _FRESH_ROWS)
            cols = np.nonzero(fresh[rows[i]])[0]
            k = 0
            while k < len(cols):
                m = k
                while m + 1 < len(cols) and cols[m + 1] == cols[m] + 1:
                    m += 1
                c0, c1 = (int(cols[k]) + bx) * 64, min(self.image_width, (int(cols[m]) + bx + 1) * 64)
                self._image[:, r0:r1, c0:c1] = 0
                if eager:
                    self._norm[:, r0:r1, c0:c1] = 0
                k = m + 1
            i = j + 1
        fresh[:] = 0

    # ------------------------------------------------------------------ helpers
    def _prep(self, batch):
        if batch.device != self._image.device:
            batch = batch.to(device=self._image.device)
        if batch.dtype not in N.DTYPE_CODES:   # fp16 / bf16 model outputs are widened inside the kernel, not copied
            batch = batch.type_as(self._image)
        return batch.detach().contiguous()

    def _check_state(self):
        for t in (self._image, self._norm, self.weight):
            N.require_device(t, "TileMerger")
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise RuntimeError("TileMerger accumulators must be contiguous float32 tensors")

    def _accumulate(self, batch, coords, views, reduction):
        self._check_state()
        if self._plan is not None and self._plan.active and self._window_edited():
            self._plan_off("integrating with an edited blending window")   # (the planned normaliser was built from the original one)
        th, tw = int(self.weight.shape[1]), int(self.weight.shape[2])
        n_views = len(views) if views is not None else 1
        B = len(coords)
        if batch.shape[0] != B * n_views or batch.shape[1] != self.channels or tuple(batch.shape[2:]) != (th, tw):
            raise RuntimeError(
                f"tile batch of shape {tuple(batch.shape)} does not match {B} tiles x {n_views} views of "
                f"[{self.channels}, {th}, {tw}]"
            )
        if B and not (coords[:, 2:] == (tw, th)).all():
            raise RuntimeError("crop size in crop_coords does not match the tile / weight size")
        xy = np.ascontiguousarray(coords[:, :2].T)          # [2, B] int64: xs row, ys row (host arrays for the C ABI)
        xs = xy[0].ctypes.data_as(N._i64p)
        ys = xy[1].ctypes.data_as(N._i64p)
        lib = N.load()
        dev = self._image.device
        varr = N.int_array(views) if views is not None else N.int_array([N.IDENT])
        norm_ptr = self._norm.data_ptr() if self._eager_norm else None
        dcode = N.DTYPE_CODES[batch.dtype]
        if self._defer_active and B and self._defer_step(batch, coords, xy, views, reduction, dcode):
            return

        def launch(fresh_ptr):
            return lib.ptb_deaug_accumulate_t(
                self._image.data_ptr(), norm_ptr, self.weight.data_ptr(), batch.data_ptr(), dcode,
                n_views, varr, reduction, xs, ys, B, self.channels, th, tw,
                self.image_height, self.image_width, fresh_ptr, _FRESH_ROWS, N.stream_ptr(dev))

        plan = self._plan
        if plan is not None and B:
            planned = (plan.active and not self._eager_norm and plan.pos + B <= plan.xy.shape[1]
                       and xy[0].data == plan.xy[0, plan.pos:plan.pos + B].data and xy[1].data == plan.xy[1, plan.pos:plan.pos + B].data)
            if planned:
                if self._merged is None:
                    self._merged = torch.empty_like(self._image)
                def launch_planned(fresh_ptr):
                    # (a self-planned merger keeps the weighted sums of finalised blocks as well: PTB_PLANNED_KEEP_SUMS)
                    return lib.ptb_accumulate_planned2(
                        self._image.data_ptr(), plan.norm_full.data_ptr(), self._merged.data_ptr(), self.weight.data_ptr(),
                        batch.data_ptr(), dcode, n_views, varr, reduction, xs, ys, B, self.channels, th, tw,
                        self.image_height, self.image_width, fresh_ptr, _FRESH_ROWS,
                        plan.remaining.ctypes.data, plan.done.ctypes.data, 1 if self._auto_planned else 0, N.stream_ptr(dev))

                with N.on_device(dev):
                    rc = launch_planned(self._fresh.ctypes.data if self._fresh.any() else None)
                    if rc == N.EFRESH:  # a cell straddles written and never-written blocks: zero-fill once, then plain RMW
                        N.fresh_fallbacks += 1
                        self._materialize()
                        rc = launch_planned(None)
                N.bump()
                if rc == 0:
                    plan.pos += B
                    self._log.append(xy)
                    return
                if rc != -2:
                    N.check(rc, "TileMerger.integrate_batch")
                # nothing was launched: this batch (and the rest of the image) takes the ordinary path
            plan.active = False
            if plan.done.any() and plan.touches_done(xy, th, tw) and self._auto_planned:
                self._unfinalise("a tile over pixels that were already merged")
            if plan.done.any() and plan.touches_done(xy, th, tw):
                raise RuntimeError("TileMerger(crops=...): a tile touches pixels that were already finalised -- every planned "
                                   "tile may be integrated once; construct the merger without crops= for free-form accumulation")
        with N.on_device(dev):
            rc = launch(self._fresh.ctypes.data if self._fresh.any() else None)
            if rc == N.EFRESH:  # geometry not block aligned (or a non-default chunk size): zero-fill once, then plain RMW
                N.fresh_fallbacks += 1
                self._materialize()
                rc = launch(None)
        N.bump()
        if rc == -2 and dcode != N.F32:   # shape needs the scalar kernels: take the reference's route (cast, then accumulate)
            return self._accumulate(batch.float(), coords, views, reduction)
        N.check(rc, "TileMerger.integrate_batch")
        if not self._eager_norm and B:
            self._log.append(xy)

    # ------------------------------------------------------------------ reference API
    def accumulate_single(self, tile: torch.Tensor, coords):
        """Accumulate one ``[C, h, w]`` prediction at ``coords = (x, y, w, h)``."""
        self._accumulate(self._prep(tile.unsqueeze(0)), _coords_xy([coords]), None, N.RED_SUM)

    def integrate_batch(self, batch: torch.Tensor, crop_coords):
        """Accumulate ``[B, C, h, w]`` predictions at ``crop_coords[b] = (x, y, w, h)``."""
        if len(batch) != len(crop_coords):
            raise ValueError("Number of images in batch does not correspond to number of coordinates")
        if type(batch) is _lazy.LazyDeaugment:
            # the reference's literal `integrate_batch(tta.d4_image_deaugment(y), crops)`: the de-augmentation has not run yet, so
            # it is fused into this launch (bit-identical: same reduction, then the same multiply and add per pixel)
            taken = batch._take_source()
            if taken is not None:
                source, _group, views, code = taken
                _lazy.fused += 1
                if self._defer_active:
                    if self._defer_fast(source, crop_coords, (_group, code), views, code):
                        return
                elif self._plan is not None and self._planned_fast(source, crop_coords, (_group, code), views, code):
                    return
                return self._accumulate(self._prep(source), _coords_xy(crop_coords), list(views), code)
        if self._defer_active:
            if self._defer_fast(batch, crop_coords, None, None, N.RED_SUM):
                return
        elif self._plan is not None and self._planned_fast(batch, crop_coords, None, None, N.RED_SUM):
            return
        self._accumulate(self._prep(batch), _coords_xy(crop_coords), None, N.RED_SUM)

    def _defer_fast(self, batch, crop_coords, key, views, code):
        """Deferred mode, the common call: a contiguous model output on this device for exactly the next planned crops (a numpy
        slice of ``tiler.crops``).  Everything constant per merger / per (group, reduction) is cached, the rest is one C call
        (``ptb_band_plan_submit``): ~10 us of host time instead of ~20.  False: the general path decides (and reports)."""
        plan = self._plan
        if not (self._defer_active and plan.active and not self._eager_norm and not self._window_edited() and type(crop_coords) is np.ndarray and crop_coords.ndim == 2 and crop_coords.dtype == np.int64
                and batch.is_cuda and batch.is_contiguous() and not batch.requires_grad):
            return False
        dcode = N.DTYPE_CODES.get(batch.dtype)
        B, pos = crop_coords.shape[0], plan.pos
        if dcode is None or B == 0 or batch.device != self._image.device or not np.array_equal(crop_coords, plan.crops4[pos:pos + B]):
            return False
        cache = self._fast_cache
        ent = cache.get(key)
        if ent is None:
            ent = cache[key] = (N.int_array(list(views)) if views is not None else N.int_array([N.IDENT]), len(views) if views is not None else 1)
        varr, n_views = ent
        th, tw = self.weight.shape[1], self.weight.shape[2]
        if batch.shape != (B * n_views, self.channels, th, tw):
            return False
        bands = self._bands
        span = _held_entry(batch)
        _check_held(self._held, batch, span, self._launch_due(pos + B), "TileMerger(defer=True)")
        if self._merged is None:
            self._merged = torch.empty_like(self._image)
        per_tile = self.channels * th * tw
        dev = self._image.device
        with N.on_device(dev):
            rc = N.load().ptb_band_plan_submit(bands.handle, pos, B, batch.data_ptr(), per_tile, B * per_tile, dcode, n_views, varr, code,
                                               self._merged.data_ptr(), plan.norm_full.data_ptr(), self.weight.data_ptr(), N.stream_ptr(dev))
        N.bump()
        if rc < 0:
            if rc == N.PTB_EUNSUPPORTED:
                return False       # (nothing was launched: the general path warns and replays)
            N.check(rc, "TileMerger.integrate_batch (deferred bands)")
        self._held.append((batch, crop_coords, views, code, int(bands.last_group[pos:pos + B].max())) + span)
        plan.pos = pos + B
        self.fast_submits += 1
        self._log.append(np.ascontiguousarray(crop_coords[:, :2].T))
        if rc:
            done = self._bands_done = self._bands_done + rc
            if done == len(bands.bands):
                self._held.clear()
            elif bands.monotone:
                held = self._held
                while held and held[0][4] < done:
                    held.pop(0)
        return True

    def _planned_fast(self, batch, crop_coords, key, views, code):
        """Planned (not deferred) mode, the common call -- a contiguous model output on this device for exactly the next planned
        crops: everything constant per merger / per (group, reduction) is cached and the rest is one C call (``ptb_accumulate_planned``):
        ~12 us of host time instead of ~40.  False: the general path decides (and reports)."""
        plan = self._plan
        if (plan is None or not plan.active or self._defer_active or self._eager_norm or type(crop_coords) is not np.ndarray or crop_coords.ndim != 2
                or crop_coords.dtype != np.int64 or not batch.is_cuda or not batch.is_contiguous() or batch.requires_grad or self._window_edited()):
            return False
        dcode = N.DTYPE_CODES.get(batch.dtype)
        B, pos = crop_coords.shape[0], plan.pos
        if (dcode is None or B == 0 or batch.device != self._image.device or pos + B > plan.xy.shape[1]
                or not np.array_equal(crop_coords, plan.crops4[pos:pos + B])):
            return False
        cache = self._fast_cache
        ent = cache.get(key)
        if ent is None:
            ent = cache[key] = (N.int_array(list(views)) if views is not None else N.int_array([N.IDENT]), len(views) if views is not None else 1)
        varr, n_views = ent
        th, tw = self.weight.shape[1], self.weight.shape[2]
        if batch.shape != (B * n_views, self.channels, th, tw) or self._image.dtype != torch.float32:
            return False
        if self._merged is None:
            self._merged = torch.empty_like(self._image)
        import ctypes

        base = plan.xy.ctypes.data            # [2, N] int64, C order: row 0 = xs, row 1 = ys
        xs = ctypes.cast(base + 8 * pos, N._i64p)
        ys = ctypes.cast(base + 8 * (plan.xy.shape[1] + pos), N._i64p)
        lib = N.load()
        dev = self._image.device
        fresh = self._fresh

        keep = 1 if self._auto_planned else 0      # (a self-planned merger keeps the sums of finalised blocks: `.image` stays exact)

        def launch(fresh_ptr):
            return lib.ptb_accumulate_planned2(self._image.data_ptr(), plan.norm_full.data_ptr(), self._merged.data_ptr(), self.weight.data_ptr(),
                                               batch.data_ptr(), dcode, n_views, varr, code, xs, ys, B, self.channels, th, tw, self.image_height,
                                               self.image_width, fresh_ptr, _FRESH_ROWS, plan.remaining.ctypes.data, plan.done.ctypes.data,
                                               keep, N.stream_ptr(dev))

        with N.on_device(dev):
            rc = launch(fresh.ctypes.data if fresh.any() else None)
            if rc == N.EFRESH:
                N.fresh_fallbacks += 1
                self._materialize()
                rc = launch(None)
        N.bump()
        if rc == 0:
            plan.pos = pos + B
            self._log.append(plan.xy[:, pos:pos + B])
            return True
        if rc == N.PTB_EUNSUPPORTED:
            return False                      # (nothing was launched: the general path takes this batch the ordinary way)
        N.check(rc, "TileMerger.integrate_batch")
        return False

    def integrate_batch_deaugment(self, batch: torch.Tensor, crop_coords, group: str = "d4", reduction="mean"):
        """Fused ``integrate_batch(tta.<group>_image_deaugment(batch, reduction), crop_coords)``.

        ``batch`` is the model output for the ``<group>_image_augment``-ed tiles, ``[V*B, C, h, w]`` chunk-major.
        One HIP launch reads the V views, applies the inverse transforms on the fly, reduces and blends.
        """
        from .tta import DEAUGMENT_VIEWS, _reduction_code

        views = DEAUGMENT_VIEWS[group]
        if (self._defer_active or self._plan is not None) and type(reduction) is str:
            code = _reduction_code(reduction)
            if code is not None:
                if self._defer_active:
                    if self._defer_fast(batch, crop_coords, (group, code), views, code):
                        return
                elif self._planned_fast(batch, crop_coords, (group, code), views, code):
                    return
        if len(batch) != len(crop_coords) * len(views):
            raise ValueError("Number of images in batch does not correspond to number of coordinates x views")
        code = _reduction_code(reduction)
        if code is None:
            raise ValueError(f"reduction={reduction!r} cannot be fused into the tile merge")
        self._accumulate(self._prep(batch), _coords_xy(crop_coords), list(views), code)

    @property
    def device(self) -> torch.device:
        return self._image.device

    @property
    def mode(self) -> str:
        """Which path this merger is on right now: "deferred bands" | "planned" | "incremental" (diagnostics; a geometry the
        faster kernels do not take, or a deviation from the planned crop sequence, degrades it -- with a one-time warning)."""
        if self._bands is not None and self._defer_active:
            return "deferred bands"
        if self._plan is not None and self._plan.active:
            return "planned"
        return "incremental"

    def _finish_planned(self):
        """Planned mode: divide whatever the accumulate launches have not finalised themselves; returns the result."""
        if self._defer_active:
            if self._bands_done == len(self._bands.bands):
                return self._merged          # every band was merged by the launch that completed it
            self._defer_flush("merge() before all planned tiles were integrated", keep_plan=True)
            if self._merged is None:
                return self._merge_into(torch.empty_like(self._image))
        plan, out = self._plan, self._merged
        pending = plan.done == 0
        if pending.any():
            self._norm_ready()
            self._materialize()
            self._check_state()
            mask = torch.from_numpy(pending.astype(np.uint8)).to(self._image.device)
            lib = N.load()
            dev = self._image.device
            with N.on_device(dev):
                rc = lib.ptb_merge_div_masked(self._image.data_ptr(), self._norm.data_ptr(), out.data_ptr(), self.channels,
                                              self.image_height, self.image_width, mask.data_ptr(), _FRESH_ROWS, N.stream_ptr(dev))
            N.bump()
            N.check(rc, "TileMerger.merge")
        return out

    def _merge_into(self, out):
        self._norm_ready()
        self._materialize()
        self._check_state()
        lib = N.load()
        dev = self._image.device
        with N.on_device(dev):
            rc = lib.ptb_merge_div(self._image.data_ptr(), self._norm.data_ptr(), out.data_ptr(), self.channels,
                                   self.image_height * self.image_width, N.stream_ptr(dev))
        N.bump()
        N.check(rc, "TileMerger.merge")
        return out

    def merge(self) -> torch.Tensor:
        """``image / norm_mask`` as a new tensor (no eps clamp: never-covered pixels are NaN, like the reference)."""
        out = self._merge_f32()
        return out if self.dtype == torch.float32 else out.to(self.dtype)

    def _merge_f32(self) -> torch.Tensor:
        self._auto_note()
        if self._merged is not None:
            # planned: the accumulate launches have been writing the result block by block.  The buffer belongs to this
            # image (reset() lets go of it); a second merge() of the same image returns the same, updated tensor.
            return self._finish_planned()
        return self._merge_into(torch.empty_like(self._image))

    def merge_(self) -> torch.Tensor:
        """In-place ``image /= norm_mask``; returns ``image`` (a converted copy of it when ``dtype`` is not float32)."""
        self._plan_off("merge_()")
        self._materialize()
        out = self._merge_into(self._image)
        return out if self.dtype == torch.float32 else out.to(self.dtype)

    _CROP_KINDS = {"float32": (0, torch.float32), "uint8": (1, torch.uint8), "argmax_u8": (2, torch.uint8), "argmax_i64": (3, torch.int64)}

    def merge_crop(self, crop, layout: str = "hwc", dtype=torch.float32, argmax: bool = False) -> torch.Tensor:
        """``merge()`` + channel-last + cast + ``crop_to_orignal_size`` in one pass over the cropped window only.

        Equals ``tiler.crop_to_orignal_size(np.moveaxis(to_numpy(merger.merge()), 0, -1).astype(dtype))`` -- the tail
        of the reference's loop (README.md:225-226; tiles.py:345-346, 271-280) -- as a device tensor, so 25-100 MB
        travel to the host instead of the 419 MB padded fp32 map.  ``crop``: the ``ImageSlicer`` (its margins and
        image size are used) or ``(top, left, height, width)``; ``layout``: "hwc" | "chw"; ``dtype``: torch.float32 |
        torch.uint8 (truncating cast, like ``ndarray.astype``); ``argmax=True`` returns ``[H, W]`` class indices
        (``dtype`` uint8 or int64) instead.
        """
        if isinstance(crop, ImageSlicer):
            top, left, oh, ow = crop.margin_top, crop.margin_left, crop.image_height, crop.image_width
        else:
            top, left, oh, ow = (int(v) for v in crop)
        if layout not in ("hwc", "chw"):
            raise ValueError(f"layout must be 'hwc' or 'chw', got {layout!r}")
        if argmax:
            key = {torch.uint8: "argmax_u8", torch.int64: "argmax_i64", torch.float32: "argmax_i64"}.get(dtype)
        else:
            key = {torch.float32: "float32", torch.uint8: "uint8"}.get(dtype)
        if key is None:
            raise NotImplementedError(f"merge_crop: dtype {dtype} is not supported")
        kind, out_dtype = self._CROP_KINDS[key]
        if top < 0 or left < 0 or oh < 0 or ow < 0 or top + oh > self.image_height or left + ow > self.image_width:
            raise ValueError("crop window is outside the accumulator")
        self._auto_note()
        planned = self._merged is not None
        if planned:
            src, norm_ptr = self._finish_planned(), None     # already normalised
        else:
            self._norm_ready()
            self._materialize()
            self._check_state()
            src, norm_ptr = self._image, self._norm.data_ptr()
        shape = (oh, ow) if argmax else ((oh, ow, self.channels) if layout == "hwc" else (self.channels, oh, ow))
        out = torch.empty(shape, device=self._image.device, dtype=out_dtype)
        if out.numel() == 0:
            return out
        lib = N.load()
        dev = self._image.device
        with N.on_device(dev):
            rc = lib.ptb_merge_crop(src.data_ptr(), norm_ptr, self.channels, self.image_height,
                                    self.image_width, top, left, oh, ow, 1 if layout == "hwc" else 0, kind, out.data_ptr(),
                                    N.stream_ptr(dev))
        N.bump()
        N.check(rc, "TileMerger.merge_crop")
        return out


class HostBackedTileMerger(TileMerger):
    """``TileMerger(device="cpu")`` (and ``dtype=torch.float64`` on any device): the reference's torch-op merger (inference/tiles.py:290-350) -- ``image`` / ``norm_mask`` /
    ``weight`` are public tensors of the caller's ``dtype`` on the CPU, accumulated in that dtype (fp64 accumulators accumulate in
    fp64), ``integrate_batch`` moves / casts what it is given and blends tile by tile in batch order.  The arithmetic lives in
    ``inference/_host.py``; the extensions of the HIP merger are accepted so that code written for either runs on both:
    ``integrate_batch_deaugment`` (= ``integrate_batch(tta.<group>_image_deaugment(...))``), ``reset()``, ``merge_crop``;
    ``crops=`` / ``defer=`` / ``auto_plan=`` change nothing here."""

    def __init__(self, image_shape, channels, weight, device="cpu", dtype=torch.float32, crops=None, defer=False, defer_rows=None,
                 auto_plan=None):
        from ._host import HostTileMerger

        device = torch.device(device)
        if device.type == "cuda":
            if dtype != torch.float64:
                raise RuntimeError("the torch-op merger serves CUDA devices for float64 accumulators only; TileMerger(device='cuda') is the HIP merger")
            _warn_once(("fp64-cuda",), "TileMerger(device='cuda', dtype=torch.float64): the HIP kernels accumulate in float32; float64 accumulators "
                                       "are kept with torch ops on the device (exact float64 sums like the reference's, not the fused kernels).")
        self.dtype = dtype
        self._host = HostTileMerger(image_shape, channels, weight, device, dtype)
        self.image_height, self.image_width, self.channels = self._host.image_height, self._host.image_width, channels

    image = property(lambda self: self._host.image, lambda self, v: setattr(self._host, "image", v))
    norm_mask = property(lambda self: self._host.norm_mask, lambda self, v: setattr(self._host, "norm_mask", v))
    weight = property(lambda self: self._host.weight, lambda self, v: setattr(self._host, "weight", v))

    @property
    def device(self) -> torch.device:
        return self._host.image.device

    @property
    def mode(self) -> str:
        return "host"

    def reset(self):
        self._host.reset()

    def accumulate_single(self, tile: torch.Tensor, coords):
        self._host.blend(tile.to(device=self.image.device).unsqueeze(0), [coords])

    def integrate_batch(self, batch: torch.Tensor, crop_coords):
        if len(batch) != len(crop_coords):
            raise ValueError("Number of images in batch does not correspond to number of coordinates")
        image = self._host.image
        if batch.device != image.device:
            batch = batch.to(device=image.device)
        if batch.dtype != image.dtype:
            batch = batch.type_as(image)
        self._host.blend(batch, crop_coords)

    def integrate_batch_deaugment(self, batch: torch.Tensor, crop_coords, group: str = "d4", reduction="mean"):
        from .tta import DEAUGMENT_VIEWS, _image_deaugment

        if len(batch) != len(crop_coords) * len(DEAUGMENT_VIEWS[group]):
            raise ValueError("Number of images in batch does not correspond to number of coordinates x views")
        self.integrate_batch(_image_deaugment(batch.to(device=self.image.device), group, reduction, lazy=False), crop_coords)

    def merge(self) -> torch.Tensor:
        return self._host.image / self._host.norm_mask

    def merge_(self) -> torch.Tensor:
        self._host.image /= self._host.norm_mask
        return self._host.image

    def merge_crop(self, crop, layout: str = "hwc", dtype=torch.float32, argmax: bool = False) -> torch.Tensor:
        if isinstance(crop, ImageSlicer):
            top, left, oh, ow = crop.margin_top, crop.margin_left, crop.image_height, crop.image_width
        else:
            top, left, oh, ow = (int(v) for v in crop)
        if layout not in ("hwc", "chw"):
            raise ValueError(f"layout must be 'hwc' or 'chw', got {layout!r}")
        if top < 0 or left < 0 or oh < 0 or ow < 0 or top + oh > self.image_height or left + ow > self.image_width:
            raise ValueError("crop window is outside the accumulator")
        window = self.merge()[:, top:top + oh, left:left + ow]
        if argmax:
            if dtype not in (torch.uint8, torch.int64, torch.float32):
                raise NotImplementedError(f"merge_crop: dtype {dtype} is not supported")
            return window.argmax(dim=0).to(torch.uint8 if dtype == torch.uint8 else torch.int64)
        if dtype not in (torch.float32, torch.uint8):
            raise NotImplementedError(f"merge_crop: dtype {dtype} is not supported")
        out = window.permute(1, 2, 0) if layout == "hwc" else window
        return out.to(dtype).contiguous()


class CudaTileMerger(TileMerger):
    """The name the reference README uses (README.md:201,215): a TileMerger that defaults to the GPU."""

    def __init__(self, image_shape, channels, weight, device="cuda", dtype=torch.float32, crops=None, defer=False, defer_rows=None, auto_plan=None):
        super().__init__(image_shape, channels, weight, device=device, dtype=dtype, crops=crops, defer=defer, defer_rows=defer_rows,
                         auto_plan=auto_plan)
