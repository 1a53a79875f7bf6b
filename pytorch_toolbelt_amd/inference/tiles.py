"""Tiled inference on huge images: host-side slicing + MI355X-side weighted blending.

Drop-in for ``pytorch_toolbelt.inference.tiles`` (reference inference/tiles.py): same class / method / attribute
names and error behaviour.  ``ImageSlicer`` is host geometry (numpy, integer-exact); ``TileMerger(device="cuda")`` keeps its
accumulators in HBM and blends with the hand-written HIP kernels of ``libptb_hip.so`` -- there is no torch-op or
CPU fallback on that path.  ``TileMerger(device="cpu")`` -- the reference's default -- is what it says: accumulators on the
host in the caller's dtype, blended with torch ops (``inference/_host.py``); the device the caller names decides, nothing else.
"""
import math
from typing import Iterable, List, Sequence, Tuple

import numpy as np
import torch

from .. import _native as N
from . import _lazy
from ._merge_modes import FRESH_ROWS as _FRESH_ROWS
from ._merge_modes import LAZY_SRC as _LAZY_SRC
from ._merge_modes import Bands as _Bands
from ._merge_modes import DeferredBands, Incremental, PlannedBlocks, SelfPlanning
from ._merge_modes import Plan as _Plan
from ._merge_modes import auto_cache as _auto
from ._merge_modes import auto_lock as _auto_lock
from ._merge_modes import check_held as _check_held
from ._merge_modes import coords_xy as _coords_xy
from ._merge_modes import default_defer_rows as _defer_rows_default
from ._merge_modes import held_entry as _held_entry
from ._merge_modes import tensor_version as _tensor_version
from ._merge_modes import _warned  # noqa: F401  (tests reset the once-only warnings)
from ._merge_modes import warn_once as _warn_once

__all__ = ["ImageSlicer", "TileMerger", "CudaTileMerger", "HostBackedTileMerger", "compute_pyramid_patch_weight_loss", "set_auto_plan", "clear_auto_plans",
           "set_reference_accumulators"]

# OpenCV border codes accepted by split/cut_patch (the reference forwards them to cv2.copyMakeBorder,
# inference/tiles.py:161,182,220).  Only constant padding is pinned by the oracle; the others map to the numpy
# mode with OpenCV's documented semantics.
BORDER_CONSTANT = 0
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
        3 wrap, 4 reflect_101 -> "reflect").  **Parity of those four is UNPINNED**: OpenCV is not installed where the goldens are
        generated and the reference's own tests never pass a non-constant border, so they are checked only against the extension tables
        of OpenCV's documentation as typed into tests/test_slicer_cpu.py (``aaaaaa|abcdefgh|hhhhhhh`` ...), not against ``cv2``."""
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


# Self-planning of mergers constructed without `crops=` (see _merge_modes.SelfPlanning): on unless PTB_AUTO_PLAN=0 / set_auto_plan(False) /
# pytorch_toolbelt_amd.set_strict_dropin().
_AUTO_PLAN = __import__("os").environ.get("PTB_AUTO_PLAN", "1") != "0"


def set_auto_plan(flag: bool) -> bool:
    """Switch self-planning of ``TileMerger`` without ``crops=`` on / off (default on; ``PTB_AUTO_PLAN=0``); returns the previous setting."""
    global _AUTO_PLAN
    prev, _AUTO_PLAN = _AUTO_PLAN, bool(flag)
    return prev


_REFERENCE_ACCUMULATORS = __import__("os").environ.get("PTB_REFERENCE_ACCUMULATORS", "0") == "1"


def set_reference_accumulators(flag: bool) -> bool:
    """``TileMerger(device="cuda", dtype=torch.float16 | torch.bfloat16)``: the reference accumulates in the caller's dtype
    (inference/tiles.py:306-308, 334-339: every ``+=`` rounds to half precision); the HIP merger sums in float32 and rounds once when
    ``merge()`` / ``image`` hand out the caller's dtype -- closer to the exact sum, not the reference's bits.  ``True`` (also set by
    ``pytorch_toolbelt_amd.set_strict_dropin()``, ``PTB_REFERENCE_ACCUMULATORS=1``) builds such mergers as the torch-op merger on the
    device, like ``dtype=torch.float64``: the reference's own op sequence, its bits, none of the fused kernels.  Default off; float32
    mergers are not affected.  Returns the previous setting."""
    global _REFERENCE_ACCUMULATORS
    prev, _REFERENCE_ACCUMULATORS = _REFERENCE_ACCUMULATORS, bool(flag)
    return prev


def _torch_op_accumulators(dtype) -> bool:
    """Accumulator dtypes of a CUDA merger that the torch-op merger keeps (``HostBackedTileMerger``) instead of the HIP kernels."""
    return dtype == torch.float64 or (_REFERENCE_ACCUMULATORS and dtype in (torch.float16, torch.bfloat16))


def clear_auto_plans() -> int:
    """Forget what self-planning mergers have learnt (per geometry: the crop sequence, a ``[1, H', W']`` normaliser and up to four band
    plan tables in HBM, at most ``_merge_modes.AUTO_MAX`` = 8 geometries, least recently used first out).  Returns the number of
    geometries dropped; mergers that are alive keep the plans they hold."""
    with _auto_lock:
        n = len(_auto)
        _auto.clear()
    return n


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


def _device_window(weight: np.ndarray, device, signature=None):
    """The blending window as a float32 ``[1, h, w]`` device tensor.  A merger per image (the README loop) would otherwise upload
    the same window from pageable host memory every time -- a synchronous copy that also waits for the GPU to drain; uploaded
    windows are kept per device (a handful of MB) and every merger gets its own device-side copy of the cached one."""
    key = (device.index if device.index is not None else torch.cuda.current_device(),) + (signature or _weight_signature(weight))
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
    cur = torch.cuda.current_stream(device)
    cur.wait_event(ready)            # (the upload may have been issued on another thread's stream)
    cached.record_stream(cur)        # ... and the LRU may drop the cached tensor while this stream's clone() is still reading it
    return cached.clone()




class TileMerger:
    """Blend tile predictions into a full-size map that lives in HBM (reference inference/tiles.py:290-350).

    ``image`` ``[C, H', W']``, ``norm_mask`` ``[1, H', W']`` and ``weight`` ``[1, h, w]`` are public fp32 tensors on the
    GPU (``image`` / ``norm_mask`` are properties: blocks no kernel has written yet are zero-filled on first read).  ``integrate_batch`` is one HIP launch per batch: overlapping tiles are accumulated race-free in batch order,
    bit-identical to the reference's sequential ``+=`` loop.  ``integrate_batch_deaugment`` additionally fuses the TTA
    de-augmentation (``tta.*_image_deaugment``) so the reduced tile never travels through HBM.

    This class owns the state every execution strategy shares (accumulators, first-touch bitmap, crop log / lazy normaliser) and
    the reference's API; HOW a batch is merged is one of three strategy objects in ``_merge_modes`` -- ``DeferredBands``,
    ``PlannedBlocks``, ``Incremental`` -- asked in that order, plus the ``SelfPlanning`` policy (``mode`` says which one is live).
    """

    def __new__(cls, image_shape=None, channels=None, weight=None, device="cpu", *args, **kwargs):
        # the device the caller names decides the implementation: "cpu" (the reference's default) -> accumulators on the host,
        # torch ops (HostBackedTileMerger below); "cuda" -> HBM + HIP kernels (this class).  Never the other way round.
        if cls is TileMerger:
            dtype = kwargs.get("dtype", args[0] if args else torch.float32)
            # float64 accumulators (reference tiles.py:306-308 accumulates in the caller's dtype): the HIP kernels sum in float32, so a
            # float64 merger -- on any device -- is the torch-op one, whose sums ARE float64 (said once for CUDA devices).
            # float16 / bfloat16 accumulators do the same when the reference's bits are asked for (set_reference_accumulators).
            if torch.device(device).type != "cuda" or _torch_op_accumulators(dtype):
                return object.__new__(HostBackedTileMerger)
        return object.__new__(cls)

    def __init__(self, image_shape, channels, weight, device="cpu", dtype=torch.float32, crops=None, defer=False, defer_rows=None,
                 auto_plan=None):
        """``crops`` (extension, optional): the complete crop list the image will receive (``tiler.crops``), in the
        order it will be integrated.  With it the merger runs *planned*: the normaliser is known up front and every
        block of the image is divided by it in the very launch that brings its last tile, so ``merge()`` has nothing
        left to do.  Results are bit-identical; see ``_merge_modes.Plan`` for what a planned merger restricts.

        ``defer=True`` (with ``crops``): *deferred* planned merging -- the merger holds on to the batches and merges a
        horizontal band of the image in one launch as soon as all its tiles are in, without an accumulator in HBM; see
        ``_merge_modes.DeferredBands`` (the batches must stay unmodified until then).  ``defer_rows``: rows merged per launch (default 1024).

        Without ``crops`` the merger plans itself (``auto_plan``; on by default, ``set_auto_plan(False)`` / ``PTB_AUTO_PLAN=0`` /
        ``pytorch_toolbelt_amd.set_strict_dropin()`` switch it off): the crop sequence an image ended with at ``merge()`` is remembered
        per geometry + window, and the next merger of that geometry (or this one after ``reset()``) runs from it -- as deferred bands
        where the geometry, the byte budget (``PTB_DEFER_BYTES``) and what the first image showed of the model's outputs allow (see
        ``_merge_modes.SelfPlanning``), else as planned blocks -- so the reference's per-image ``TileMerger(shape, C, weight)`` gets
        the headline kernel from the second image on.  A deviating batch, a read of ``image`` / ``norm_mask`` or ``merge_()`` drop back
        to the ordinary path without an exception (bit-exactly before any band was merged, within one float32 rounding after)."""
        device = _resolve_device(device, "TileMerger")
        # The reference keeps image / norm_mask / weight in `dtype` (tiles.py:295-308) and so accumulates in it.  Here the accumulators
        # are always float32 (what the kernels read-modify-write); any other floating dtype is honoured at the boundary: tile batches
        # of that dtype are read as they are, and merge() / image / norm_mask hand out tensors of that dtype.  For float16 / bfloat16
        # that is a strictly more accurate sum than the reference's; set_reference_accumulators(True) / set_strict_dropin() build such
        # mergers as the torch-op merger instead, for the reference's bits.  (float64: __new__ hands it to the torch-op merger; a subclass that gets here with it is refused below.)
        if dtype not in (torch.float32, torch.float16, torch.bfloat16, torch.float64):
            raise TypeError(f"TileMerger: dtype must be a floating point type, got {dtype}")
        if _torch_op_accumulators(dtype):
            # (TileMerger / CudaTileMerger route these dtypes to the torch-op merger in __new__; any other subclass of the HIP merger would
            # silently sum in float32 and cast -- ADVICE round 4; float16 / bfloat16 only once set_reference_accumulators(True) asked for
            # the reference's bits)
            raise TypeError(f"{type(self).__name__}: {str(dtype).replace('torch.', '')} accumulators are kept by the torch-op merger (TileMerger(..., dtype={dtype}) / "
                            "HostBackedTileMerger); the HIP kernels of this class accumulate in float32")
        self.dtype = dtype
        dtype = torch.float32
        N.load()
        self.image_height = image_shape[0]
        self.image_width = image_shape[1]
        self.channels = channels
        if isinstance(weight, np.ndarray):
            signature = _weight_signature(weight)      # (hashed once per merger: 50 us for a 512 x 512 float64 window)
            self.weight = _device_window(weight, device, signature)
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
        self._merged = None       # planned / deferred: the merge result the launches fill in
        self._weight_version0, self._weight_ptr0 = _tensor_version(self.weight), self.weight.data_ptr()
        self._view_arrays = {}    # (group, reduction) -> (ctypes view-code array, number of views)
        self.fast_submits = 0     # deferred batches that took the cached host path (diagnostic)
        # ---- the strategies
        self._plan = _Plan.build(self, crops) if crops is not None else None      # shared by PlannedBlocks and DeferredBands
        self._incremental = Incremental(self)
        self._planned = PlannedBlocks(self)
        auto_key = None
        if crops is None and (auto_plan if auto_plan is not None else _AUTO_PLAN) and isinstance(weight, np.ndarray):
            auto_key = (device.index if device.index is not None else torch.cuda.current_device(), int(self.image_height),
                        int(self.image_width)) + signature
        self._selfplan = SelfPlanning(self, auto_key)
        self._selfplan.attach()
        bands = None
        band_only = _Plan.build(self, crops, blocks=False) if (defer and crops is not None and self._plan is None) else None
        if defer and (self._plan is not None or band_only is not None):
            bands = _Bands.build(self._plan or band_only, channels, int(self.weight.shape[1]), int(self.weight.shape[2]), self.image_height,
                                 self.image_width, device, defer_rows if defer_rows is not None else _defer_rows_default())
            if bands is not None and self._plan is None:
                self._plan = band_only       # off the 64 x 32 block grid but on the band kernel's 4-pixel grid: deferred bands, no block strategy behind them
            if bands is None and self._plan is not None:
                _warn_once(("defer", tuple(self.weight.shape), self.image_height, self.image_width),
                           "TileMerger(defer=True): the deferred band kernel does not take this geometry (tile origins, tile size or image width "
                           "off the 4-pixel grid, more than 224 tiles per launch group or more than 4 tiles over a pixel); the planned incremental "
                           "path is used (same results, slower).")
        elif defer:
            _warn_once(("defer-noplan",), "TileMerger(defer=True) needs the complete crop list (crops=tiler.crops) on the band kernel's 4-pixel "
                                          "grid; the ordinary path is used.")
        if crops is not None and self._plan is None:
            _warn_once(("plan", tuple(self.weight.shape), self.image_height, self.image_width),
                       "TileMerger(crops=...): this geometry is off the 64 x 32 block grid of the planned kernels (tile size / origins)"
                       + (" and off the 4-pixel grid of the deferred band kernel" if defer else "") + "; "
                       "the ordinary accumulate + merge path is used (same results, one more pass).")
        soft = False
        if bands is None and self._selfplan.bands is not None:      # planned from the previous image of this geometry, band plan included
            bands, soft = self._selfplan.bands, True
        self._deferred = DeferredBands(self, bands, soft)

    # ------------------------------------------------------------------ strategy state under its former names (tests, bench, tools)
    _bands = property(lambda self: self._deferred.bands)
    _bands_done = property(lambda self: self._deferred.done)
    _defer_active = property(lambda self: self._deferred.active)
    _held = property(lambda self: self._deferred.held)
    _auto_planned = property(lambda self: self._selfplan.planned)

    def _window_edited(self):
        w = self.weight
        return _tensor_version(w) != self._weight_version0 or w.data_ptr() != self._weight_ptr0

    def _view_array(self, key, views):
        ent = self._view_arrays.get(key)
        if ent is None:
            ent = self._view_arrays[key] = (N.int_array(list(views)) if views is not None else N.int_array([N.IDENT]),
                                            len(views) if views is not None else 1)
        return ent

    # ------------------------------------------------------------------ accumulators as the reference's public attributes
    @property
    def image(self) -> torch.Tensor:
        """``[C, H', W']`` accumulator (zeros where nothing was integrated yet)."""
        self._planned.off("reading .image")
        self._materialize()
        return self._image if self.dtype == torch.float32 else self._image.to(self.dtype)   # (a copy: write through integrate_* only)

    @image.setter
    def image(self, value: torch.Tensor):
        self._planned.off("assigning .image")
        self._materialize()
        self._image = value.to(device=self._image.device, dtype=torch.float32).contiguous()

    def _norm_handed_out(self):
        if self._plan is not None:
            self._plan.active = False   # (finalised blocks keep their results; the rest accumulates with this norm)
        self._selfplan.opt_out()
        self._norm_ready()
        self._materialize()
        self._eager_norm, self._norm_pure, self._norm_key = True, False, None

    @property
    def norm_mask(self) -> torch.Tensor:
        """``[1, H', W']`` sum of the blending windows (materialised on first access; from then on the accumulate
        kernels keep this tensor up to date, exactly like the reference's attribute)."""
        self._norm_handed_out()
        return self._norm if self.dtype == torch.float32 else self._norm.to(self.dtype)

    @norm_mask.setter
    def norm_mask(self, value: torch.Tensor):
        self._norm_handed_out()
        self._norm = value

    def reset(self):
        """Start a new image: the accumulators become logically zero again (no memset: O(1) on the host)."""
        self._fresh[:] = 1
        self._log, self._applied = [], 0
        self._norm_zero, self._norm_pure, self._eager_norm = True, False, False
        self._merged = None
        self._selfplan.noted = None
        if self._selfplan.key is not None:
            self._selfplan.attach()   # plan from what the last image(s) looked like / restart / drop a plan that no longer holds
            self._deferred.rebind(self._selfplan.bands, self._selfplan.bands is not None)
        elif self._plan is not None:
            self._plan.restart()
        self._deferred.reset()

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

    def _accumulate(self, batch, coords, views, reduction, rnd=0):
        """Any batch, validated once, offered to the strategies in order: deferred bands, planned blocks, incremental.  ``rnd`` (flags):
        ``N.ROUND_SRC`` -- the batch is the half-precision source of a lazy de-augmentation handle (the reduced value is rounded to the
        source dtype before it is blended, as the reference's two calls do); ``_LAZY_SRC`` -- it is the source of such a handle at all
        (what a self-deferring merger may keep without a version counter: ``_merge_modes.DeferredBands._submit``)."""
        self._check_state()
        if self._plan is not None and self._plan.active and self._window_edited():
            self._planned.off("integrating with an edited blending window")   # (the planned normaliser was built from the original one)
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
        dcode = N.DTYPE_CODES[batch.dtype]
        if rnd & N.ROUND_SRC and dcode != N.F32:
            dcode |= N.ROUND_SRC
        if B:
            self._selfplan.observe(batch, n_views, rnd & _LAZY_SRC)
        if B and self._deferred.active and self._deferred.take(batch, coords, xy, views, reduction, dcode | (rnd & _LAZY_SRC)):
            return
        xs = xy[0].ctypes.data_as(N._i64p)
        ys = xy[1].ctypes.data_as(N._i64p)
        varr = N.int_array(views) if views is not None else N.int_array([N.IDENT])
        if B and self._plan is not None and self._planned.take(batch, xy, xs, ys, n_views, varr, reduction, dcode):
            return
        return self._incremental.take(batch, coords, xy, xs, ys, views, n_views, varr, reduction, dcode)

    def _offer_fast(self, batch, crop_coords, key, views, code, rnd=0):
        """The live strategy's cheap host path for the common call; False: ``_accumulate`` validates and decides."""
        if self._deferred.active:
            return self._deferred.take_fast(batch, crop_coords, key, views, code, rnd)
        return self._plan is not None and self._planned.take_fast(batch, crop_coords, key, views, code, rnd)

    # ------------------------------------------------------------------ reference API
    def accumulate_single(self, tile: torch.Tensor, coords):
        """Accumulate one ``[C, h, w]`` prediction at ``coords = (x, y, w, h)``."""
        self._accumulate(self._prep(tile.unsqueeze(0)), _coords_xy([coords]), None, N.RED_SUM)

    def integrate_batch(self, batch: torch.Tensor, crop_coords):
        """Accumulate ``[B, C, h, w]`` predictions at ``crop_coords[b] = (x, y, w, h)``."""
        kind = type(batch)
        # (Tensor.__len__ is a Python function: 0.9 us; on a lazy handle it travels through __torch_function__: 3 us -- the handle knows its length)
        if (batch.shape[0] if kind is torch.Tensor else (batch._len if kind is _lazy.LazyDeaugment else len(batch))) != len(crop_coords):
            raise ValueError("Number of images in batch does not correspond to number of coordinates")
        if kind is _lazy.LazyDeaugment:
            # the reference's literal `integrate_batch(tta.d4_image_deaugment(y), crops)`: the de-augmentation has not run yet, so
            # it is fused into this launch (bit-identical: same reduction, then the same multiply and add per pixel)
            taken = batch._take_source()
            if taken is not None:
                source, _group, views, code = taken
                _lazy.fused += 1
                # a half-precision source (torch.autocast): the handle stands for a HALF tensor, i.e. the reduced value rounded to
                # the source dtype (tta.py:442-467) before tiles.py:334-335 widens it again -- the fused launch rounds in registers
                rnd = _LAZY_SRC | (0 if source.dtype == torch.float32 else N.ROUND_SRC)
                if self._offer_fast(source, crop_coords, (_group, code), views, code, rnd):
                    return
                return self._accumulate(self._prep(source), _coords_xy(crop_coords), list(views), code, rnd)
        if self._offer_fast(batch, crop_coords, None, None, N.RED_SUM):
            return
        self._accumulate(self._prep(batch), _coords_xy(crop_coords), None, N.RED_SUM)

    def integrate_batch_deaugment(self, batch: torch.Tensor, crop_coords, group: str = "d4", reduction="mean"):
        """Fused ``integrate_batch(tta.<group>_image_deaugment(batch, reduction), crop_coords)``.

        ``batch`` is the model output for the ``<group>_image_augment``-ed tiles, ``[V*B, C, h, w]`` chunk-major.
        One HIP launch reads the V views, applies the inverse transforms on the fly, reduces and blends.
        """
        from .tta import DEAUGMENT_VIEWS, _reduction_code

        views = DEAUGMENT_VIEWS[group]
        if type(reduction) is str:
            code = _reduction_code(reduction)
            if code is not None and self._offer_fast(batch, crop_coords, (group, code), views, code):
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
        """Which strategy this merger is on right now: "deferred bands" | "planned" | "incremental" (diagnostics; a geometry the
        faster kernels do not take, or a deviation from the planned crop sequence, degrades it -- with a one-time warning)."""
        if self._deferred.active:
            return "deferred bands"
        if self._plan is not None and self._plan.active:
            return "planned"
        return "incremental"

    def _finish_planned(self):
        """``_merged`` exists (planned or deferred launches have been writing the result): complete it and hand it out."""
        if self._deferred.active:
            out = self._deferred.finish()
            if out is not None:
                return out
            if self._merged is None:      # (nothing had been submitted: the replay went the ordinary way)
                return self._incremental.merge_into(torch.empty_like(self._image))
        return self._planned.finish()

    def merge(self) -> torch.Tensor:
        """``image / norm_mask`` as a new tensor (no eps clamp: never-covered pixels are NaN, like the reference)."""
        out = self._merge_f32()
        return out if self.dtype == torch.float32 else out.to(self.dtype)

    def _merge_f32(self) -> torch.Tensor:
        self._selfplan.note()
        if self._merged is not None:
            # planned: the accumulate launches have been writing the result block by block.  The buffer belongs to this
            # image (reset() lets go of it); a second merge() of the same image returns the same, updated tensor.
            return self._finish_planned()
        return self._incremental.merge_into(torch.empty_like(self._image))

    def merge_(self) -> torch.Tensor:
        """In-place ``image /= norm_mask``; returns ``image`` (a converted copy of it when ``dtype`` is not float32)."""
        self._planned.off("merge_()")
        self._materialize()
        out = self._incremental.merge_into(self._image)
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
        self._selfplan.note()
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
            if not _torch_op_accumulators(dtype):
                raise RuntimeError("the torch-op merger serves CUDA devices for float64 accumulators (and, under set_reference_accumulators(True), "
                                   "float16 / bfloat16 ones) only; TileMerger(device='cuda') is the HIP merger")
            if dtype == torch.float64:
                _warn_once(("fp64-cuda",), "TileMerger(device='cuda', dtype=torch.float64): the HIP kernels accumulate in float32; float64 accumulators "
                                           "are kept with torch ops on the device (exact float64 sums like the reference's, not the fused kernels).")
            else:
                _warn_once(("ref-acc-cuda", str(dtype)), f"TileMerger(device='cuda', dtype={dtype}) under set_reference_accumulators(True): accumulated in "
                                                         f"{dtype} with torch ops on the device (the reference's bits, not the fused kernels).")
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

    def __new__(cls, image_shape=None, channels=None, weight=None, device="cuda", *args, **kwargs):
        # float64 accumulators get the torch-op merger on the device, exactly like TileMerger(..., device="cuda", dtype=torch.float64) does
        # (device="cpu" stays an error: this is the name of the GPU merger)
        dtype = kwargs.get("dtype", args[0] if args else torch.float32)
        if cls is CudaTileMerger and torch.device(device).type == "cuda" and _torch_op_accumulators(dtype):
            return HostBackedTileMerger(image_shape, channels, weight, device, *args, **kwargs)      # (not a CudaTileMerger: __init__ is not run again)
        return object.__new__(cls)

    def __init__(self, image_shape, channels, weight, device="cuda", dtype=torch.float32, crops=None, defer=False, defer_rows=None, auto_plan=None):
        super().__init__(image_shape, channels, weight, device=device, dtype=dtype, crops=crops, defer=defer, defer_rows=defer_rows,
                         auto_plan=auto_plan)
