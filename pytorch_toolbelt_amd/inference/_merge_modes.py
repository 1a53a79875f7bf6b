"""The execution strategies of the HIP ``TileMerger`` (inference/tiles.py), one class each.

``TileMerger`` owns what every strategy shares -- the accumulators in HBM, their first-touch bitmap, the crop log the lazy
normaliser is built from, the reference's public API (reference inference/tiles.py:290-350) -- and asks its strategies, in this
order, whether they take a batch:

    DeferredBands   ``TileMerger(crops=, defer=True)``: holds the model outputs and merges a band of rows in ONE launch when its last
                    tile is in; no accumulator traffic at all (csrc/ptb_bandplan.hip)
    PlannedBlocks   ``TileMerger(crops=)`` (or a self-planned merger): accumulates, and divides every 64 x 32 block by the
                    precomputed normaliser in the launch that brings its last tile -- ``merge()`` has nothing left to do
    Incremental     the reference's semantics literally: accumulate now, divide in ``merge()``

A strategy that cannot take a batch (a deviation from the planned crop sequence, a caller that reads ``norm_mask``, a geometry
off the kernels' grid) says so and the next one does, bit-identically; ``SelfPlanning`` is the policy (on by default) that gives a merger
constructed WITHOUT ``crops=`` a plan from the crop sequence of the previous image.  ``HeldBatches`` -- the custody contract of
model outputs that are read by a later launch -- is shared with ``parallel.ShardedTileMerger``'s deferred bands.
"""
import collections
import ctypes
import threading
import warnings
import weakref

import numpy as np
import torch

from .. import _native as N

_current_device = torch._C._cuda_getDevice if hasattr(torch._C, "_cuda_getDevice") else torch.cuda.current_device
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or (lambda idx: torch.cuda.current_stream(idx).cuda_stream)

FRESH_ROWS = 32  # rows of a first-touch block = default chunk rows of the view kernels (64 columns wide)
LAZY_SRC = 0x10000  # host-only bit of a batch's `how` flags (next to N.ROUND_SRC): the batch is the source of a lazy de-augmentation handle --
                    # handed out only where a change of it would be noticed (version counter) or nobody else can reach it (inference/_lazy.py)


def default_defer_rows():
    import os

    return int(os.environ.get("PTB_DEFER_ROWS", "1024"))

_warned = set()


def warn_once(key, message):
    if key not in _warned:
        _warned.add(key)
        warnings.warn(message, RuntimeWarning, stacklevel=3)


def coords_xy(crop_coords):
    """crop_coords: ndarray [B,4], list of 4-sequences, or the CPU int64 tensor default_collate builds."""
    if torch.is_tensor(crop_coords):
        arr = crop_coords.detach().cpu().numpy()
    else:
        arr = np.asarray([[int(v) for v in c] for c in crop_coords] if not isinstance(crop_coords, np.ndarray) else crop_coords)
    return np.ascontiguousarray(arr, dtype=np.int64).reshape(-1, 4)


# ------------------------------------------------------------------------------------------------ custody of held model outputs
def tensor_version(t):
    try:
        return t._version
    except RuntimeError:      # inference tensors carry no version counter: in-place edits of them cannot be seen
        return None


def held_entry(batch):
    """(first byte, one past the last byte, version counter) of a batch a deferred merger is about to keep a reference to."""
    p0 = batch.data_ptr()
    return p0, p0 + batch.numel() * batch.element_size(), tensor_version(batch)


def check_held(held, batch, span, launches, what, hint="construct the merger without defer=True"):
    """The contract of deferred merging, enforced: a held batch is read by a LATER launch, so (1) a new batch must not live in
    the memory of one that is still held -- a model writing into a static output buffer (HIP graphs, ``out=``, preallocated
    outputs) has then already overwritten data the merger has not read, which no fallback can bring back -- and (2) a held batch
    must not have been modified in place since it was handed in (checked when its launch is due).  ``held`` rows start with the
    tensor and end with (p0, p1, version); ``span`` = ``held_entry(batch)``."""
    p0, p1, _v = span
    for h in held:
        if h[-3] < p1 and p0 < h[-2]:
            raise RuntimeError(f"{what}: this batch occupies memory of an earlier batch that is still held for a later launch (bytes "
                               f"{max(p0, h[-3]):#x}..{min(p1, h[-2]):#x}) -- the model writes its outputs into a reused buffer, so the earlier "
                               "predictions are already gone.  Deferred merging needs every batch to stay alive and unmodified until its rows "
                               f"are merged: hand over fresh tensors (or clones), or {hint}.")
    if launches:
        for i, h in enumerate(held):
            if h[-1] is not None and tensor_version(h[0]) != h[-1]:
                raise RuntimeError(f"{what}: held batch {i} of the rows about to be merged was modified in place after it was handed to the "
                                   "merger (its version counter moved).  Deferred merging reads the batches later: keep them unmodified, "
                                   f"or {hint}.")


class HeldBatches:
    """The model outputs a deferred merger has taken into custody, in integration order: rows ``(tensor, replay, last launch group
    that reads it, p0, p1, version)``.  One object per merger / per rank band; ``admit`` is the contract check above."""

    __slots__ = ("rows", "what", "hint")

    def __init__(self, what):
        self.rows, self.what, self.hint = [], what, "construct the merger without defer=True"

    def admit(self, batch, launch_due):
        """Check a batch against everything still held; returns its span for ``keep``."""
        span = held_entry(batch)
        check_held(self.rows, batch, span, launch_due, self.what, self.hint)
        return span

    def keep(self, batch, span, replay=None, last_group=0):
        self.rows.append((batch, replay, last_group) + span)

    def release_before(self, groups_done):
        """Launch groups 0 .. groups_done-1 are out (and complete in index order): let go of what no later group reads."""
        rows = self.rows
        while rows and rows[0][2] < groups_done:
            rows.pop(0)

    def take_all(self):
        rows, self.rows = self.rows, []
        return rows

    def clear(self):
        self.rows = []

    def __len__(self):
        return len(self.rows)

    def __iter__(self):
        return iter(self.rows)


# ------------------------------------------------------------------------------------------------ plans
class Plan:
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
        self.blocks = remaining0 is not None      # False: a plan for deferred bands only (a geometry off the 64 x 32 block grid of PlannedBlocks)
        if remaining0 is None:
            remaining0 = np.zeros((0, 0), dtype=np.uint8)
        self.remaining0 = remaining0
        self.norm_full = norm_full      # [1, H, W] complete normaliser (device)
        self.remaining = remaining0.copy()
        self.done = np.zeros_like(remaining0)
        self.pos = 0
        self.active = True
        self._crops4 = None             # the planned (x, y, w, h) rows (what callers' crop arrays are compared with) ...
        self.crops_bytes = b""          # ... and their bytes: `next(crops)` compares 32 B per tile instead of calling numpy

    @property
    def crops4(self):
        return self._crops4

    @crops4.setter
    def crops4(self, value):
        self._crops4 = value
        self.crops_bytes = value.tobytes()

    def next_are(self, crop_coords, B):
        """Are these ``B`` crops (an int64 ``[B, 4]`` array) exactly the next planned ones?  (0.4 us: one ``tobytes`` and a bytes compare.)"""
        pos = self.pos
        return crop_coords.tobytes() == self.crops_bytes[32 * pos:32 * (pos + B)]

    @staticmethod
    def build(merger, crops, blocks=True):
        """``blocks=False``: a plan for the deferred band merger alone -- the crop sequence and the normaliser, no block bookkeeping: any
        geometry inside the image will do here (whether the band kernel takes it -- the 4-pixel grid -- is ``Bands.build``'s answer)."""
        crops = coords_xy(crops)
        th, tw = int(merger.weight.shape[1]), int(merger.weight.shape[2])
        H, W = merger.image_height, merger.image_width
        if len(crops) == 0 or np.any(crops[:, 2] != tw) or np.any(crops[:, 3] != th):
            return None
        inside = np.all(crops[:, 0] >= 0) and np.all(crops[:, 1] >= 0) and np.all(crops[:, 0] + tw <= W) and np.all(crops[:, 1] + th <= H)
        aligned = inside and tw % 64 == 0 and th % FRESH_ROWS == 0 and not np.any(crops[:, 0] % 64) and not np.any(crops[:, 1] % FRESH_ROWS)
        if not (aligned if blocks else inside):
            return None   # geometry off the block grid: the ordinary path is used
        remaining = None
        if blocks:
            remaining = np.zeros(((H + FRESH_ROWS - 1) // FRESH_ROWS, (W + 63) // 64), dtype=np.int32)
            for x, y in crops[:, :2]:
                remaining[y // FRESH_ROWS:(y + th) // FRESH_ROWS, x // 64:(x + tw) // 64] += 1
            if remaining.max() > 255:
                return None
            remaining = remaining.astype(np.uint8)
        xy = np.ascontiguousarray(crops[:, :2].T)
        norm_full = torch.zeros((1, H, W), device=merger.weight.device, dtype=torch.float32)
        lib = N.load()
        dev = norm_full.device
        with N.on_device(dev):
            rc = lib.ptb_norm_accumulate(norm_full.data_ptr(), merger.weight.data_ptr(), xy[0].ctypes.data_as(N._i64p),
                                         xy[1].ctypes.data_as(N._i64p), xy.shape[1], th, tw, H, W, None, 0, N.stream_ptr(dev))
        N.bump()
        N.check(rc, "TileMerger(crops=...)")
        plan = Plan(xy, remaining, norm_full)
        plan.crops4 = np.ascontiguousarray(crops, dtype=np.int64)
        return plan

    def restart(self):
        self.remaining = self.remaining0.copy()
        self.done[:] = 0
        self.pos = 0
        self.active = True

    def follows(self, xy, B):
        """Are these ``B`` origins ([2, B] int64, C order) exactly the next planned ones?"""
        pos = self.pos
        return (pos + B <= self.xy.shape[1] and xy[0].data == self.xy[0, pos:pos + B].data and xy[1].data == self.xy[1, pos:pos + B].data)

    def touches_done(self, xy, th, tw):
        for x, y in xy.T:
            if self.done[y // FRESH_ROWS:(y + th + FRESH_ROWS - 1) // FRESH_ROWS, x // 64:(x + tw + 63) // 64].any():
                return True
        return False


class Bands:
    """The C-side band plan of a deferred merger (``ptb_band_plan_*``, csrc/ptb_bandplan.hip), planned once.

    A *band* is the rows between two consecutive tile edges; every tile that touches a band covers all of its rows.  Consecutive
    bands form a *launch group* of about ``rows`` rows (default 1024; ``defer_rows=`` / ``PTB_DEFER_ROWS``)."""

    def __init__(self, handle, table, bands, n_bands, last_group, monotone):
        self.handle = handle            # ptb_band_plan*
        self.table = table              # uint8 device tensor holding the work-item table (owned here)
        self.bands = bands              # [(y0, y1, last tile)] per launch group, top to bottom
        self.n_bands = n_bands
        self.last_group = last_group    # plan index of a tile -> the last launch group that reads it
        self.monotone = monotone        # groups complete in index order (row-major crops): batches can be released early
        self.ready = None               # event behind the table upload (a plan shared through the self-planning cache is used on other streams)
        self.rows = None                # rows per launch group it was built with
        self._peak = None               # peak_tiles(), computed once
        self.sorted_tiles = bool(np.all(np.diff(last_group) >= 0)) if len(last_group) > 1 else True      # (row-major crops: a later tile is never read by an earlier group)

    def peak_tiles(self):
        """The most tiles in custody at any launch: when group g goes out, every tile handed in so far that g or a later group reads
        is still held (row-major crops: a contiguous index range); a crop order whose groups do not complete in index order holds
        everything to the end."""
        if self._peak is not None:
            return self._peak
        n = len(self.last_group)
        if not self.monotone:
            self._peak = n
            return n
        peak, first = 0, 0
        for g, (_y0, _y1, last) in enumerate(self.bands):
            while first < n and self.last_group[first] < g:
                first += 1
            peak = max(peak, last + 1 - first)
        self._peak = peak
        return peak

    def __del__(self):
        try:
            if self.handle:
                N.load().ptb_band_plan_destroy(self.handle)
                self.handle = None
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    @staticmethod
    def build(plan, channels, th, tw, H, W, device, rows):
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
        out = Bands(handle, table, groups, nb.value, last_group, all(a <= b for a, b in zip(lasts, lasts[1:])))
        out.rows = int(rows)
        out.ready = torch.cuda.Event()
        out.ready.record(torch.cuda.current_stream(device))
        return out


# ------------------------------------------------------------------------------------------------ strategies
def _fast_call(merger, batch, crop_coords):
    """The common call -- a contiguous model output on the merger's device that needs no autograd detach, crops as an int64 [B, 4]
    array (a numpy slice of ``tiler.crops``): everything per call can then be validated with a handful of comparisons."""
    return (type(crop_coords) is np.ndarray and crop_coords.ndim == 2 and crop_coords.dtype == np.int64 and batch.is_cuda
            and batch.is_contiguous() and not batch.requires_grad and not merger._eager_norm and not merger._window_edited())


class _OfMerger:
    """A strategy object's way to its merger: a weak reference (one call per method; attribute access on the merger is then direct --
    through a ``weakref.proxy`` every access paid the indirection, ~2 us per integrate call)."""

    __slots__ = ()

    @property
    def m(self):
        return self._m()


class DeferredBands(_OfMerger):
    """Strategy "deferred bands" (``TileMerger(..., crops=tiler.crops, defer=True)``).

    The merger only keeps references to the model outputs it is handed, and when the last tile of a launch group has arrived ONE
    launch reads all covering tiles of its rows, de-augments, reduces, blends in integration order and writes ``sum / norm`` to the
    result -- the accumulator image never travels through HBM (the incremental path re-reads and re-writes every pixel once per
    overlapping tile row).  The fp32 operation order per pixel is the incremental path's, so the result is bit-identical.  Cost:
    the batches of the last ``rows / step + 1`` tile rows stay alive until their group is done, and they must not be modified in
    place in the meantime -- which is why, as an argument of the constructor, this is opt-in (``soft`` mergers -- the ones that planned
    THEMSELVES into this mode -- come with the safeguards of ``SelfPlanning`` instead and never raise for what the caller did not ask for).

    Until the first group is launched any deviation from the plan simply replays the held batches through the incremental path;
    afterwards ``merger.image``, a partial ``merge()`` or an unplanned tile raise."""

    def __init__(self, merger, bands, soft=False):
        self._m = weakref.ref(merger)    # (strategies never own their merger: no reference cycle, a dropped merger frees its HBM at once)
        self.bands = bands               # Bands, or None: this merger does not defer (the strategy is never active)
        self.soft = soft                 # the merger planned ITSELF into this mode: nothing it was never asked for may raise (see soften)
        self.held = HeldBatches("TileMerger(defer=True)")
        self.done = 0                    # launch groups issued for this image
        self.active = False
        self.budget_checked = False
        self.slim = None                 # the image's configuration once its first batch went through take_fast (see take_slim)
        self.rebind(bands, soft)
        self.reset()

    def rebind(self, bands, soft):
        """Another band plan (or none) for the next image: a self-planned merger gets its plan from the geometry's cache at reset()."""
        self.bands, self.soft = bands, soft
        self.held.what = "TileMerger (self-planned deferred bands)" if soft else "TileMerger(defer=True)"
        self.held.hint = ("construct it with TileMerger(..., auto_plan=False) (or call pytorch_toolbelt_amd.set_strict_dropin()): every batch "
                          "is then read inside integrate_batch") if soft else "construct the merger without defer=True"

    def reset(self):
        self.active = self.bands is not None
        self.held.clear()
        self.done = 0
        self.budget_checked = False
        self.slim = None
        if self.bands is not None:
            N.load().ptb_band_plan_reset(self.bands.handle)

    @property
    def complete(self):
        return self.bands is not None and self.done == len(self.bands.bands)

    def flush(self, what, keep_plan=False):
        """Leave deferred mode: replay the held batches through the incremental path (only before the first band) --
        the planned one when ``keep_plan`` (nothing deviated from the plan), else the ordinary one."""
        if not self.active:
            return
        if self.done:
            if self.soft:
                return self.soften(what)
            raise RuntimeError(f"TileMerger(defer=True): {what} is not available after bands of the image were merged; "
                               "integrate the planned tiles and call merge(), or construct the merger without defer=True")
        m = self.m
        held = self.held.take_all()
        self.active = False
        self.slim = None
        m._plan.restart()
        m._plan.active = keep_plan and m._plan.blocks      # (a band-only plan has no block strategy to hand the image to)
        m._merged = None                                   # (nothing was launched into it; the planned strategy makes its own)
        m._log, m._applied = [], 0
        for batch, (coords, views, reduction, rnd), *_rest in held:
            m._accumulate(batch, coords, views, reduction, rnd)

    def soften(self, what):
        """A merger that deferred ON ITS OWN ACCOUNT (self-planned) is asked for something deferred merging cannot serve after bands
        of the image went out -- a tile off the remembered sequence, ``merge()`` of an image that ends early, a read of ``image``.
        The caller never opted into that restriction, so nothing raises: the merger turns back into the ordinary accumulating one.
        Rows already merged get ``merged * norm`` as their weighted sums (every tile over them has been blended: the one place where
        this library's result can differ from the sequential sums, by the rounding of one multiply -- far inside the 1e-5 contract,
        said once), the batches still held are replayed onto the rest, and the image carries on incrementally."""
        m, bands = self.m, self.bands
        warn_once(("soften", m._selfplan.key), f"TileMerger: {what} after the self-planned merger had already merged rows of the image from held "
                                               "model outputs; continuing on the ordinary accumulate + merge path (rows merged so far are carried over "
                                               "as merged * norm_mask: equal to the sequential sums within one float32 rounding).  "
                                               "TileMerger(..., auto_plan=False) or pytorch_toolbelt_amd.set_strict_dropin() keep every image on that path.")
        lib = N.load()
        launched = [(y0, y1) for (y0, y1, _last) in bands.bands if y1 > y0 and lib.ptb_band_plan_rows_launched(bands.handle, y0, y1) == 1]
        held = self.held.take_all()
        self.active = False
        self.slim = None
        merged, m._merged = m._merged, None
        plan = m._plan
        plan.restart()
        plan.active = False
        log = list(m._log)                 # every crop of the image so far (the normaliser is built from it); the replay must not log twice
        m._materialize()
        for batch, (coords, views, reduction, rnd), *_rest in held:
            m._accumulate(batch, coords, views, reduction, rnd)
        m._log = log
        m._norm_ready()
        for y0, y1 in launched:
            norm = m._norm[:, y0:y1]
            m._image[:, y0:y1] = torch.where(norm == 0, torch.zeros_like(norm), merged[:, y0:y1] * norm)

    def over_budget(self, per_tile_bytes, B):
        """Would this image keep more model outputs alive than the byte budget of self-planned deferral (``PTB_DEFER_BYTES``)?"""
        return (self.bands.peak_tiles() + B) * per_tile_bytes > defer_budget()

    def launch_due(self, end):
        """Will a submit that brings the planned tiles up to index ``end`` (exclusive) launch a group?  (Only then are the held
        batches' version counters compared; groups of a row-major crop list complete in index order.)"""
        bands = self.bands
        if not bands.monotone:
            return True
        return self.done < len(bands.bands) and end > bands.bands[self.done][2]

    def _submit(self, batch, coords, views, pos, B, dcode, n_views, varr, code):
        """Take the planned tiles ``pos .. pos+B-1`` into custody and merge the launch groups they complete
        (``ptb_band_plan_submit``: the pointer bookkeeping and the launches happen in C).  Returns its code: < 0 nothing was taken."""
        m, bands = self.m, self.bands
        plan = m._plan
        if self.soft and not (dcode & LAZY_SRC) and tensor_version(batch) is None:
            # A merger that defers ON ITS OWN ACCOUNT reads this batch in a later launch, and nothing could tell it that the caller
            # changed it in between: tensors made under torch.inference_mode() carry no version counter (`m1.integrate_batch(y, c);
            # y.sigmoid_(); m2.integrate_batch(y, c)` silently corrupted m1).  Such batches are read inside integrate_batch -- planned blocks.
            m._selfplan.unprovable()
            return N.PTB_EUNSUPPORTED
        dcode &= ~LAZY_SRC
        if self.soft and not self.budget_checked:
            self.budget_checked = True
            per_tile = n_views * m.channels * int(m.weight.shape[1]) * int(m.weight.shape[2]) * batch.element_size()
            if self.over_budget(per_tile, B):
                m._selfplan.too_big(per_tile, B)
                return N.PTB_EUNSUPPORTED
        span = self.held.admit(batch, self.launch_due(pos + B))
        if m._merged is None:
            m._merged = torch.empty_like(m._image)
        per_tile = m.channels * int(m.weight.shape[1]) * int(m.weight.shape[2])
        dev = m._image.device
        with N.on_device(dev):
            rc = N.load().ptb_band_plan_submit(bands.handle, pos, B, batch.data_ptr(), per_tile, B * per_tile, dcode, n_views, varr, code,
                                               m._merged.data_ptr(), plan.norm_full.data_ptr(), m.weight.data_ptr(), N.stream_ptr(dev))
        N.bump()
        if rc < 0:
            return rc
        lg = bands.last_group
        self.held.keep(batch, span, (coords, views, code, dcode & N.ROUND_SRC), int(lg[pos + B - 1]) if bands.sorted_tiles else int(lg[pos:pos + B].max()))
        plan.pos = pos + B
        if rc:
            done = self.done = self.done + rc
            if done == len(bands.bands):
                self.held.clear()
            elif bands.monotone:      # groups 0 .. done-1 are out: batches no later group reads can go
                self.held.release_before(done)
        return rc

    def take_slim(self, batch, crop_coords, key, rnd):
        """Every batch of an image after its first one, in ~4 us of host time (round 6; the 14-argument ``ptb_band_plan_submit`` path costs
        ~7 us a call -- 0.33 ms per 5000 x 5000 image, more than the kernels of the TTA-free loop take): the image's configuration
        (dtype, views, reduction, buffers) sits in the C plan since its first submit, custody overlap is checked there
        (``ptb_band_plan_submit_next``), and what is left here is the contract of the call itself -- the next planned crops, a contiguous
        batch of the image's dtype / shape on the merger's device, held batches unmodified when a launch is due.  None: not that call
        (the ordinary fast path decides)."""
        s = self.slim
        if (key != s[0] or rnd != s[1] or type(crop_coords) is not np.ndarray or crop_coords.dtype.char != "l" or batch.dtype is not s[2]
                or not batch.is_contiguous() or batch.requires_grad or not self.active):
            return None
        m = self._m()
        plan = m._plan
        B, pos = crop_coords.shape[0], plan.pos
        shape = batch.shape
        if (B == 0 or shape[0] != B * s[4] or shape[1:] != s[3] or m._eager_norm or not plan.active or batch.get_device() != s[5]
                or _current_device() != s[5] or not plan.next_are(crop_coords, B)):
            return None
        bands, held = self.bands, self.held
        try:
            version = batch._version
        except RuntimeError:
            version = None
            if self.soft and not (rnd & LAZY_SRC):
                return None                      # (a version-less batch nobody vouches for: _submit leaves deferred mode)
        end = pos + B
        p0 = batch.data_ptr()
        if not bands.monotone or (self.done < len(bands.bands) and end > bands.bands[self.done][2]):
            # a launch is due: it reads the held batches (and the window) -- they must be what they were when they were handed in
            if m._window_edited():
                return None
            for h in held.rows:
                if h[-1] is not None and tensor_version(h[0]) != h[-1]:
                    # (the full contract check words the refusal: a batch in a held batch's memory first -- a static output buffer
                    # moves the counter of every slice of it too --, then the in-place edit)
                    check_held(held.rows, batch, (p0, p0 + B * s[7], version), True, held.what, held.hint)
        rc = s[6](bands.handle, p0, B, _raw_stream(s[5]))
        N.calls += 1
        if rc < 0:
            if rc == N.PTB_EHELD:
                check_held(held.rows, batch, (p0, p0 + B * s[7], version), False, held.what, held.hint)      # (raises, naming the overlap)
            if rc == N.PTB_EUNSUPPORTED:
                return None
            N.check(rc, "TileMerger.integrate_batch (deferred bands)")
        lg = bands.last_group
        held.rows.append((batch, (crop_coords, s[8], s[9], rnd & N.ROUND_SRC), int(lg[end - 1]) if bands.sorted_tiles else int(lg[pos:end].max()),
                          p0, p0 + B * s[7], version))
        plan.pos = end
        if rc:
            done = self.done = self.done + rc
            if done == len(bands.bands):
                held.rows = []
            elif bands.monotone:
                held.release_before(done)
        m.fast_submits += 1
        m._log.append(plan.xy[:, pos:end])
        return True

    def take_fast(self, batch, crop_coords, key, views, code, rnd=0):
        """The common call (see ``_fast_call``) for exactly the next planned crops: everything constant per merger / per
        (group, reduction) is cached, the rest is one C call: ~7 us of host time instead of ~20 -- and ~4 us from the image's second batch
        on (``take_slim``).  False: ``take`` decides (and reports)."""
        if self.slim is not None:
            taken = self.take_slim(batch, crop_coords, key, rnd)
            if taken is not None:
                return taken
        m = self.m
        plan = m._plan
        if not (self.active and plan.active and _fast_call(m, batch, crop_coords)):
            return False
        dcode = N.DTYPE_CODES.get(batch.dtype)
        B, pos = crop_coords.shape[0], plan.pos
        if dcode is None or B == 0 or batch.device != m._image.device or crop_coords.shape[1] != 4 or not plan.next_are(crop_coords, B):
            return False
        if rnd:
            dcode |= (rnd & LAZY_SRC) | (rnd & N.ROUND_SRC if dcode else 0)
        varr, n_views = m._view_array(key, views)
        if batch.shape != (B * n_views, m.channels, m.weight.shape[1], m.weight.shape[2]):
            return False
        rc = self._submit(batch, crop_coords, views, pos, B, dcode, n_views, varr, code)
        if rc < 0:
            if rc == N.PTB_EUNSUPPORTED:
                return False       # (nothing was launched: the general path warns and replays)
            N.check(rc, "TileMerger.integrate_batch (deferred bands)")
        m.fast_submits += 1
        m._log.append(plan.xy[:, pos:pos + B])      # (the planned origins themselves: a view, no copy -- they were just compared equal)
        if self.active and self.slim is None and self.bands is not None:
            dev = m._image.device
            idx = dev.index if dev.index is not None else torch.cuda.current_device()
            th, tw = int(m.weight.shape[1]), int(m.weight.shape[2])
            self.slim = (key, rnd, batch.dtype, torch.Size((m.channels, th, tw)), n_views, idx, N.load().ptb_band_plan_submit_next,
                         n_views * m.channels * th * tw * batch.element_size(), views, code)
        return True

    def take(self, batch, coords, xy, views, reduction, dcode):
        """Any validated batch.  False: not deferrable -- the held batches were replayed and the caller goes on with the next
        strategy."""
        m = self.m
        plan = m._plan
        B = xy.shape[1]
        rc = N.PTB_EUNSUPPORTED
        soft = self.soft        # (read before _submit: its budget / provability exits rebind the strategy, and what follows is about the merger the caller made)
        if plan.active and not m._eager_norm and plan.follows(xy, B):
            varr = N.int_array(views) if views is not None else N.int_array([N.IDENT])
            rc = self._submit(batch, coords, views, plan.pos, B, dcode, len(views) if views is not None else 1, varr, reduction)
        if rc == N.PTB_EUNSUPPORTED:
            if not soft:
                warn_once(("defer-deviation",), "TileMerger(defer=True): a batch deviates from the planned crop sequence / configuration (or "
                                                "norm_mask was read); leaving deferred mode for this image, the held batches are replayed incrementally.")
            self.flush("a tile batch off the remembered crop sequence" if soft else "an unplanned tile batch", keep_plan=soft and self.done == 0
                       and plan.active and not m._eager_norm and plan.follows(xy, B))
            return False
        if rc < 0:
            N.check(rc, "TileMerger.integrate_batch (deferred bands)")
        m._log.append(xy)
        return True

    def finish(self):
        """At ``merge()``: the finished result, or None after handing an incomplete image back to the planned strategy."""
        if self.complete:
            return self.m._merged          # every band was merged by the launch that completed it
        self.flush("merge() before all planned tiles were integrated", keep_plan=True)
        return None


class PlannedBlocks(_OfMerger):
    """Strategy "planned" (``TileMerger(..., crops=tiler.crops)`` and self-planned mergers): ``ptb_accumulate_planned2`` accumulates
    a batch and writes ``sum / norm`` of every block whose last planned tile this batch brings -- see ``Plan`` for what that
    restricts.  A self-planned merger asks the kernel to keep the weighted sums of finalised blocks as well
    (PTB_PLANNED_KEEP_SUMS), so its accumulators stay exact."""

    def __init__(self, merger):
        self._m = weakref.ref(merger)

    def _launch(self, batch, dcode, n_views, varr, code, xs, ys, B):
        m = self.m
        plan = m._plan
        if m._merged is None:
            m._merged = torch.empty_like(m._image)
        lib = N.load()
        dev = m._image.device
        th, tw = int(m.weight.shape[1]), int(m.weight.shape[2])
        keep = 1 if m._selfplan.planned else 0
        fresh = m._fresh

        def launch(fresh_ptr):
            return lib.ptb_accumulate_planned2(m._image.data_ptr(), plan.norm_full.data_ptr(), m._merged.data_ptr(), m.weight.data_ptr(),
                                               batch.data_ptr(), dcode, n_views, varr, code, xs, ys, B, m.channels, th, tw, m.image_height,
                                               m.image_width, fresh_ptr, FRESH_ROWS, plan.remaining.ctypes.data, plan.done.ctypes.data,
                                               keep, N.stream_ptr(dev))

        with N.on_device(dev):
            rc = launch(fresh.ctypes.data if fresh.any() else None)
            if rc == N.EFRESH:  # a cell straddles written and never-written blocks: zero-fill once, then plain RMW
                N.fresh_fallbacks += 1
                m._materialize()
                rc = launch(None)
        N.bump()
        return rc

    def take_fast(self, batch, crop_coords, key, views, code, rnd=0):
        """The common call (see ``_fast_call``) for exactly the next planned crops: ~12 us of host time instead of ~40.  False:
        ``take`` decides (and reports)."""
        m = self.m
        plan = m._plan
        if plan is None or not plan.active or not plan.blocks or m._deferred.active or not _fast_call(m, batch, crop_coords):
            return False
        dcode = N.DTYPE_CODES.get(batch.dtype)
        if dcode and rnd & N.ROUND_SRC:
            dcode |= N.ROUND_SRC
        B, pos = crop_coords.shape[0], plan.pos
        if (dcode is None or B == 0 or batch.device != m._image.device or pos + B > plan.xy.shape[1] or crop_coords.shape[1] != 4
                or not plan.next_are(crop_coords, B)):
            return False
        varr, n_views = m._view_array(key, views)
        if batch.shape != (B * n_views, m.channels, m.weight.shape[1], m.weight.shape[2]) or m._image.dtype != torch.float32:
            return False
        base = plan.xy.ctypes.data            # [2, N] int64, C order: row 0 = xs, row 1 = ys
        xs = ctypes.cast(base + 8 * pos, N._i64p)
        ys = ctypes.cast(base + 8 * (plan.xy.shape[1] + pos), N._i64p)
        rc = self._launch(batch, dcode, n_views, varr, code, xs, ys, B)
        if rc == 0:
            plan.pos = pos + B
            m._log.append(plan.xy[:, pos:pos + B])
            return True
        if rc == N.PTB_EUNSUPPORTED:
            return False                      # (nothing was launched: the general path takes this batch the ordinary way)
        N.check(rc, "TileMerger.integrate_batch")
        return False

    def take(self, batch, xy, xs, ys, n_views, varr, reduction, dcode):
        """Any validated batch.  False: the ordinary path takes it (and the rest of the image)."""
        m = self.m
        plan = m._plan
        B = xy.shape[1]
        if not plan.blocks:         # a band-only plan that is no longer deferring: the ordinary path from here on
            plan.active = False
            return False
        if plan.active and not m._eager_norm and plan.follows(xy, B):
            rc = self._launch(batch, dcode, n_views, varr, reduction, xs, ys, B)
            if rc == 0:
                plan.pos += B
                m._log.append(xy)
                return True
            if rc != -2:
                N.check(rc, "TileMerger.integrate_batch")
            # nothing was launched: this batch (and the rest of the image) takes the ordinary path
        plan.active = False
        if plan.done.any():
            th, tw = int(m.weight.shape[1]), int(m.weight.shape[2])
            if plan.touches_done(xy, th, tw):
                if not m._selfplan.planned:
                    raise RuntimeError("TileMerger(crops=...): a tile touches pixels that were already finalised -- every planned "
                                       "tile may be integrated once; construct the merger without crops= for free-form accumulation")
                m._selfplan.unfinalise("a tile over pixels that were already merged")
        return False

    def off(self, what):
        """Leave planned mode for this image; impossible once blocks were finalised (their accumulators were never stored) --
        unless the merger planned itself and kept them."""
        m = self.m
        m._deferred.flush(what)
        plan = m._plan
        if plan is not None:
            if plan.done.any():
                if not m._selfplan.planned:
                    raise RuntimeError(f"TileMerger(crops=...): {what} is not available after planned blocks were finalised; "
                                       "call merge(), or construct the merger without crops=")
                m._selfplan.unfinalise(what)
            plan.active = False
        m._selfplan.opt_out()

    def finish(self):
        """At ``merge()``: divide whatever the accumulate launches have not finalised themselves; returns the result."""
        m = self.m
        plan, out = m._plan, m._merged
        pending = plan.done == 0
        if pending.any():
            m._norm_ready()
            m._materialize()
            m._check_state()
            mask = torch.from_numpy(pending.astype(np.uint8)).to(m._image.device)
            lib = N.load()
            dev = m._image.device
            with N.on_device(dev):
                rc = lib.ptb_merge_div_masked(m._image.data_ptr(), m._norm.data_ptr(), out.data_ptr(), m.channels,
                                              m.image_height, m.image_width, mask.data_ptr(), FRESH_ROWS, N.stream_ptr(dev))
            N.bump()
            N.check(rc, "TileMerger.merge")
        return out


class Incremental(_OfMerger):
    """Strategy "incremental": the reference's semantics literally (inference/tiles.py:321-346) -- one ``ptb_deaug_accumulate_t``
    launch per batch adds the weighted tiles to the accumulator in batch order, ``merge()`` divides."""

    def __init__(self, merger):
        self._m = weakref.ref(merger)

    def take(self, batch, coords, xy, xs, ys, views, n_views, varr, reduction, dcode):
        m = self.m
        B = xy.shape[1]
        lib = N.load()
        dev = m._image.device
        th, tw = int(m.weight.shape[1]), int(m.weight.shape[2])
        norm_ptr = m._norm.data_ptr() if m._eager_norm else None

        def launch(fresh_ptr):
            return lib.ptb_deaug_accumulate_t(m._image.data_ptr(), norm_ptr, m.weight.data_ptr(), batch.data_ptr(), dcode, n_views, varr,
                                              reduction, xs, ys, B, m.channels, th, tw, m.image_height, m.image_width, fresh_ptr,
                                              FRESH_ROWS, N.stream_ptr(dev))

        with N.on_device(dev):
            rc = launch(m._fresh.ctypes.data if m._fresh.any() else None)
            if rc == N.EFRESH:  # geometry not block aligned (or a non-default chunk size): zero-fill once, then plain RMW
                N.fresh_fallbacks += 1
                m._materialize()
                rc = launch(None)
        N.bump()
        if rc == -2 and dcode != N.F32:   # shape needs the scalar kernels: take the reference's route (cast, then accumulate)
            if dcode & N.ROUND_SRC:        # (the source of a lazy de-augmentation handle: evaluate it as the eager call would -- a half tensor -- and blend that)
                from ._views import deaug_reduce

                return m._accumulate(deaug_reduce(batch, views if views is not None else [N.IDENT], reduction).float(), coords, None, N.RED_SUM)
            return m._accumulate(batch.float(), coords, views, reduction)
        N.check(rc, "TileMerger.integrate_batch")
        if not m._eager_norm and B:
            m._log.append(xy)

    def merge_into(self, out):
        m = self.m
        m._norm_ready()
        m._materialize()
        m._check_state()
        lib = N.load()
        dev = m._image.device
        with N.on_device(dev):
            rc = lib.ptb_merge_div(m._image.data_ptr(), m._norm.data_ptr(), out.data_ptr(), m.channels,
                                   m.image_height * m.image_width, N.stream_ptr(dev))
        N.bump()
        N.check(rc, "TileMerger.merge")
        return out


# ------------------------------------------------------------------------------------------------ self-planning mergers
# The reference's loop builds `TileMerger(tiler.target_shape, C, tiler.weight)` -- no crop list -- for every image and feeds it the
# same crops in the same order (README.md:201-226).  A merger without `crops=` therefore records the crop sequence it saw
# (at `merge()`), and the NEXT merger of the same geometry and window (or the same one after `reset()`) plans itself from it.
# Round 5: the plan it gets is the one `TileMerger(crops=, defer=True)` gets -- deferred bands: the merger keeps references to the
# model outputs of one launch group and merges every band in the launch that brings its last tile, no accumulator in HBM -- so the
# reference's literal calls reach the headline kernel from the second image of a geometry on.  What the caller never asked for is
# kept away from them:
#   * the FIRST image of a geometry always runs incrementally, and while it does the merger watches the model outputs it is handed
#     (`SelfPlanning.observe`): a batch that lives in the memory of the previous, still referenced one means the model writes into a
#     static buffer (HIP graphs, `out=`) -- such a geometry never defers (its mergers take the planned-block kernels, which read
#     every batch inside `integrate_batch`);
#   * the model outputs kept alive are bounded (`PTB_DEFER_BYTES`, default 4 GiB: rows per launch are halved until the peak fits,
#     else planned blocks);
#   * any deviation -- a batch off the remembered sequence, an image that ends early, a read of `image` / `norm_mask`, `merge_()` --
#     turns the merger back into the ordinary one (`DeferredBands.soften`), without an exception.
# On by default since round 5; `TileMerger(auto_plan=False)`, `tiles.set_auto_plan(False)`, `PTB_AUTO_PLAN=0` or
# `pytorch_toolbelt_amd.set_strict_dropin()` switch it off.
AUTO_MAX = 8            # geometries remembered (each keeps a [1, H', W'] normaliser and <= 4 band tables in HBM once planned)
auto_cache = collections.OrderedDict()   # key -> AutoEntry
auto_lock = threading.RLock()            # mergers of several threads (one inference loop each) share the cache
POOL_MAX = 4            # band plans kept per geometry (a plan carries per-image state: one merger at a time uses it)


def defer_budget():
    import os

    return int(os.environ.get("PTB_DEFER_BYTES", str(4 << 30)))


class AutoEntry:
    __slots__ = ("log", "seen", "need", "parts", "disabled", "static", "tile_bytes", "batch_tiles", "pool", "rows", "no_defer", "unversioned")

    def __init__(self):
        self.log = None        # bytes of the [n, 4] int64 crop sequence of the last merged image
        self.seen = 0          # consecutive merged images that ended with exactly this sequence
        self.need = 1          # repeats required before planning (grows when a planned image deviated)
        self.parts = None      # (xy, remaining0, norm_full, crops4, built event) shared by the mergers planned from `log`
        self.disabled = False  # this geometry cannot be planned / its user reads accumulators or merges partially
        self.static = False    # its model outputs were seen to share memory while alive (static output buffer): never deferred
        self.tile_bytes = 0    # bytes of model output per tile (views x channels x h x w x element size) of the last image
        self.batch_tiles = 0   # tiles per integrate call of the last image (custody is released batch-wise)
        self.pool = []         # free Bands built for `parts` with `rows` rows per launch
        self.rows = None       # rows per launch the budget allows (None: not decided; 0: deferral does not fit / not available)
        self.no_defer = False  # the band kernel does not take this geometry, or an image was over the byte budget
        self.unversioned = False  # its batches carry no version counter and are not the sole-owned sources of lazy handles: never held (planned blocks)


def auto_entry(key, create=False):
    with auto_lock:
        ent = auto_cache.get(key)
        if ent is None and create:
            while len(auto_cache) >= AUTO_MAX:
                auto_cache.popitem(last=False)
            ent = auto_cache[key] = AutoEntry()
        elif ent is not None:
            auto_cache.move_to_end(key)
        return ent


class SelfPlanning(_OfMerger):
    """The policy that gives a merger constructed without ``crops=`` a ``Plan`` -- and, where the geometry and the byte budget allow,
    the band plan of deferred merging (see the comment above).  ``key`` None: this merger never plans itself and every method is a
    no-op."""

    def __init__(self, merger, key):
        self._m, self.key = weakref.ref(merger), key
        self.planned = False      # merger._plan was made here (from the previous image's crops), not by the caller
        self.noted = None         # log length at the last merge() of this image
        self.bands = None         # the Bands this merger has checked out of its geometry's pool
        self.prev = None          # learning pass: deque of (batch, p0, p1) -- the window of model outputs kept alive (observe); False: outputs are static
        self.prev_bytes = 0
        self.tile_bytes = self.batch_tiles = 0

    def __del__(self):
        try:
            self.release()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    # ---------------------------------------------------------------- band plans: a pool per geometry
    def release(self):
        """Hand the band plan back to the geometry's pool (the merger is done with it: dropped, re-planned or degraded)."""
        bands, self.bands = self.bands, None
        if bands is None or self.key is None:
            return
        with auto_lock:
            ent = auto_cache.get(self.key)
            if ent is not None and ent.parts is not None and getattr(bands, "parts_id", None) is ent.parts[0] and len(ent.pool) < POOL_MAX:
                ent.pool.append(bands)

    def _acquire(self, ent, plan):
        """A band plan for ``plan`` whose custody fits the byte budget, or None (then the merger runs planned blocks)."""
        m = self.m
        if ent.static or ent.no_defer or ent.unversioned or ent.rows == 0:
            return None
        with auto_lock:
            if ent.pool:
                bands = ent.pool.pop()
                torch.cuda.current_stream(bands.table.device).wait_event(bands.ready)
                return bands
        th, tw = int(m.weight.shape[1]), int(m.weight.shape[2])
        dev = m._image.device
        rows = ent.rows
        candidates = [rows] if rows else []
        if not candidates:
            r = default_defer_rows()
            while r >= max(4, min(th, 256) // 2):
                candidates.append(r)
                r //= 2
        budget = defer_budget()
        for r in candidates:
            bands = Bands.build(plan, m.channels, th, tw, m.image_height, m.image_width, dev, r)
            if bands is None:
                ent.no_defer = True      # off the band kernel's grid (tile origins / sizes, > 224 tiles per group, > 4 tiles per pixel)
                return None
            bands.parts_id = ent.parts[0]
            if ent.tile_bytes and (bands.peak_tiles() + ent.batch_tiles) * ent.tile_bytes > budget:
                continue
            ent.rows = r
            return bands
        ent.rows = 0
        return None

    def too_big(self, per_tile, B):
        """First batch of a self-deferred image: the model outputs it would keep alive exceed the budget (more views / a wider dtype
        / larger batches than the image the plan was sized for).  Nothing is held yet: this image runs planned blocks, the geometry
        re-sizes its launch groups for what it has seen now."""
        m = self.m
        m._deferred.active = False
        with auto_lock:
            ent = auto_cache.get(self.key)
            if ent is not None:
                ent.rows, ent.pool = None, []
                ent.tile_bytes, ent.batch_tiles = max(ent.tile_bytes, per_tile), max(ent.batch_tiles, B)
        self.tile_bytes, self.batch_tiles = max(self.tile_bytes, per_tile), max(self.batch_tiles, B)
        self.bands = None
        m._deferred.rebind(None, False)

    def attach(self):
        """(Re)plan the merger from the crop sequence its geometry ended the last image(s) with, when there is a stable one."""
        if self.key is None:
            return
        m = self.m
        self.prev = None
        ent = auto_entry(self.key)
        usable = (ent is not None and not ent.disabled and ent.log is not None and ent.seen >= ent.need and not m._window_edited())
        if not usable:
            self.release()
            if self.planned:
                m._plan, self.planned = None, False
            return
        if self.planned and ent.parts is not None and ent.parts[0] is m._plan.xy:
            m._plan.restart()
            if self.bands is None:
                self.bands = self._acquire(ent, m._plan)
            if self.bands is None and not m._plan.blocks:
                m._plan, self.planned = None, False
            return
        self.release()
        if ent.parts is None:
            crops4 = np.frombuffer(ent.log, dtype=np.int64).reshape(-1, 4)
            plan = Plan.build(m, crops4)
            if plan is None and not ent.static and not ent.unversioned:
                plan = Plan.build(m, crops4, blocks=False)      # off the 64 x 32 block grid: deferred bands or nothing
            if plan is None:          # this geometry never plans
                ent.disabled = True
                m._plan, self.planned = None, False
                return
            built = torch.cuda.Event()
            built.record(torch.cuda.current_stream(plan.norm_full.device))
            ent.parts = (plan.xy, plan.remaining0, plan.norm_full, plan.crops4, built)
            ent.pool, ent.rows = [], None
        xy, remaining0, norm_full, crops4, built = ent.parts
        torch.cuda.current_stream(norm_full.device).wait_event(built)      # (the normaliser may have been built on another stream)
        plan = Plan(xy, remaining0 if len(remaining0) else None, norm_full)
        plan.crops4 = crops4
        bands = self._acquire(ent, plan)
        if bands is None and not plan.blocks:      # a band-only plan without bands (budget, static outputs, the band kernel's grid): no plan at all
            ent.disabled = ent.disabled or ent.no_defer or ent.static or ent.unversioned
            m._plan, self.planned = None, False
            return
        m._plan, self.planned = plan, True
        self.bands = bands

    def opt_out(self):
        """The caller touched the accumulators themselves: this geometry stays on the ordinary (exact, unplanned) path from now on."""
        if self.key is not None:
            auto_entry(self.key, create=True).disabled = True

    def observe(self, batch, n_views, lazy=0):
        """Learning pass (an image of a keyed merger that is not planned): remember how much model output a tile brings, and notice
        model outputs that share memory while alive.  Round 6: the batches of the image are kept referenced as far back as a deferred
        image of this geometry could ever keep them -- the byte budget of self-planned deferral (``PTB_DEFER_BYTES``) bounds that
        custody, so it bounds this window too.  With a model that returns fresh tensors the allocator then cannot hand a block of the
        window out again; an overlap means the model writes into a static buffer or cycles through a RING of output buffers (two captured
        graphs, ``out=bufs[i % k]``) no deeper than what deferral would hold -- such a geometry never defers (planned blocks read every
        batch inside ``integrate_batch``); a deeper ring is harmless by construction.  Batches without a version counter that are not
        the sources of lazy handles (``torch.inference_mode()``) are noted too: nothing could tell a later launch that they changed."""
        if self.key is None or self.planned:
            return
        per_tile = n_views * int(np.prod(batch.shape[1:])) * batch.element_size()
        if per_tile > self.tile_bytes:
            self.tile_bytes = per_tile
        self.batch_tiles = max(self.batch_tiles, batch.shape[0] // n_views)
        p0, p1, version = held_entry(batch)
        if version is None and not lazy:
            auto_entry(self.key, create=True).unversioned = True
        window = self.prev
        if window is None:
            window = self.prev = collections.deque()
            self.prev_bytes = 0
        elif window is False:        # (static outputs already seen: nothing more to learn, nothing more to hold)
            return
        for _t, q0, q1 in window:
            if p0 < q1 and q0 < p1:
                auto_entry(self.key, create=True).static = True
                self.prev = False
                return
        window.append((batch, p0, p1))
        self.prev_bytes += p1 - p0
        budget = defer_budget()
        while len(window) > 1 and self.prev_bytes > budget:
            _t, q0, q1 = window.popleft()
            self.prev_bytes -= q1 - q0

    def unprovable(self):
        """A self-deferred image is handed a batch nobody could vouch for later (no version counter, not a lazy handle's source): its
        geometry stops deferring (planned blocks from here on; ``DeferredBands.take`` replays what is held)."""
        with auto_lock:
            ent = auto_cache.get(self.key)
            if ent is not None:
                ent.unversioned = True
                ent.pool = []

    def note(self):
        """At merge(): remember the crop sequence this image was made of (what the next image of this geometry is planned from)."""
        if self.key is None:
            return
        m = self.m
        self.prev = None
        n = len(m._log)
        if self.noted == n:
            return
        ent = auto_entry(self.key, create=True)
        if self.noted is not None or m._eager_norm or m._window_edited():
            ent.disabled = True       # tiles after a merge() / a caller-visible norm_mask / an edited window: not the README loop
            return
        self.noted = n
        if n == 0:
            return
        if self.tile_bytes:
            ent.tile_bytes, ent.batch_tiles = self.tile_bytes, self.batch_tiles
        plan = m._plan
        if self.planned and plan.active and plan.pos == plan.xy.shape[1] and ent.parts is not None and ent.parts[0] is plan.xy:
            ent.seen += 1             # the planned sequence, start to end
            return
        xy = np.concatenate(m._log, axis=1)
        crops4 = np.empty((xy.shape[1], 4), dtype=np.int64)
        crops4[:, 0], crops4[:, 1] = xy[0], xy[1]
        crops4[:, 2], crops4[:, 3] = int(m.weight.shape[2]), int(m.weight.shape[1])
        log = crops4.tobytes()
        if self.planned:              # a planned image that went another way: ask for more evidence before planning again
            ent.need = min(ent.need + 1, 4)
        if ent.log == log:
            ent.seen += 1
        else:
            ent.log, ent.seen, ent.parts, ent.pool, ent.rows, ent.no_defer = log, 1, None, [], None, False

    def unfinalise(self, what):
        """Somebody needs the accumulators of blocks the planned kernels have already turned into results.  A merger that planned
        ITSELF stores the weighted sum of a block next to its merged value (PTB_PLANNED_KEEP_SUMS: one more store of the image per
        image), so the accumulators are complete and exact -- the same bits the unplanned kernels would have left; the merger simply
        goes back to the ordinary path and its geometry stops planning itself."""
        plan = self.m._plan
        warn_once(("unfinalise", self.key), f"TileMerger: {what} after the self-planned kernels had finalised part of the image; the "
                                            "accumulators are complete (self-planned mergers keep them), mergers of this geometry use the "
                                            "ordinary accumulate + merge path from now on (TileMerger(..., auto_plan=False) avoids the switch).")
        plan.done[:] = 0
        plan.active = False
        self.opt_out()
