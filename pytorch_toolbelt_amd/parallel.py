"""Multi-GPU tile merging: one image, tiles sharded over the ranks of one node (one process per GPU, RCCL over xGMI).

The reference has no multi-GPU tile path; the single-device ``TileMerger`` result is the specification
(SURVEY.md 8e).  Design for MI355X's point-to-point xGMI fabric:

* tiles are partitioned into contiguous ranges of the row-major tile sequence by the reference's own
  ``split_across_nodes`` linspace rule (utils/distributed.py:306-309): 361 tiles over 8 ranks = 45 or 46 each (whole
  tile rows, 19 over 8 ranks, would leave one rank with 3 rows = 57 tiles); a rank's tiles cover a horizontal band;
* every rank accumulates its band locally with the fused HIP kernels (no communication on the data path);
* each rank **owns** a range of pixel rows of the result.  The only exchange is the part of a band that lies in
  another rank's rows, and the ownership cut always lies in the middle of the rows two neighbours share: when the boundary
  between two ranks coincides with the start of a tile row, each ships half of the tile_size - tile_step shared rows to the
  other (2 x 10.5 MB for the headline config, in opposite directions of one link); when it falls inside a tile row, the two
  ranks swap two half-height rectangles as wide as their share of that tile row (the same 21 MB in total).  All pairs run concurrently on distinct xGMI links as point-to-point
  send/receive, overlapped with the accumulation of the tiles that do not feed a rectangle.  Nothing is all-reduced: a
  ring all-reduce of the 524 MB accumulator would be per-link bound and ~30x slower than the kernels;
* ``norm_mask`` is data independent, so every rank computes the global normaliser of its owned rows once, locally;
* ``merge()`` adds what it received and divides (a full-width strip is folded into the division kernel), returning
  this rank's band.
"""
from typing import List, Sequence

import numpy as np
import torch

__all__ = ["tile_row_partition", "tile_range_partition", "pixel_row_partition", "pixel_row_cuts", "band_plan", "early_spans", "PendingBand", "shared_exchange", "deferred_geometry", "ShardedTileMerger", "RcclExchange", "all_reduce_sum", "sync_region_statistics", "ms_strip_plan", "ms_flips_image_deaugment_strip",
           "ms_image_deaugment_strip"]


def tile_row_partition(crops: np.ndarray, world: int) -> List[np.ndarray]:
    """Tile indices per rank: contiguous groups of whole tile rows, boundaries at
    ``np.linspace(0, n_rows, world+1, dtype=int)`` (19 rows / 8 ranks -> at most 3 rows: speed-up bound 6.33x)."""
    crops = np.asarray(crops)
    row_y = np.unique(crops[:, 1])
    cuts = np.linspace(0, len(row_y), world + 1, dtype=int)
    order = np.lexsort((crops[:, 0], crops[:, 1]))
    return [order[np.isin(crops[order, 1], row_y[cuts[r]:cuts[r + 1]])].astype(np.int64) for r in range(world)]


def tile_range_partition(crops: np.ndarray, world: int) -> List[np.ndarray]:
    """Tile indices per rank: contiguous ranges of the row-major tile sequence with boundaries at
    ``np.linspace(0, n_tiles, world+1, dtype=int)`` -- the reference's own ``split_across_nodes`` rule
    (utils/distributed.py:306-309).  361 tiles / 8 ranks -> 45 or 46 per rank (speed-up bound 7.85x); a rank boundary may
    fall in the middle of a tile row."""
    crops = np.asarray(crops)
    order = np.lexsort((crops[:, 0], crops[:, 1]))
    cuts = np.linspace(0, len(order), world + 1, dtype=int)
    return [order[cuts[r]:cuts[r + 1]].astype(np.int64) for r in range(world)]


PARTITIONS = {"tiles": tile_range_partition, "rows": tile_row_partition}


def pixel_row_cuts(image_height: int, world: int) -> List[int]:
    """``world + 1`` row positions that split ``image_height`` rows evenly: on the 64-row grid of the band kernel's work items when
    that keeps every share non-empty, else on its 4-row grid, else (tiny images) wherever ``np.linspace`` puts them."""
    for grid in (64, 4, 1):
        cuts = [int(round(v / grid)) * grid for v in np.linspace(0, image_height, world + 1)]
        cuts[0], cuts[-1] = 0, int(image_height)
        if all(b > a for a, b in zip(cuts[:-1], cuts[1:])) or grid == 1:
            return [min(max(c, 0), int(image_height)) for c in cuts]
    return cuts


def pixel_row_partition(crops: np.ndarray, world: int, image_height: int) -> List[np.ndarray]:
    """Communication-free sharding (SURVEY 8e "communication-free alternative"): rank r owns the PIXEL rows
    ``pixel_row_cuts(...)[r:r + 2]`` and gets every tile that touches them, in row-major order -- tiles that straddle a cut are
    handed to both neighbours (their model outputs are computed twice: at the headline geometry 76 tiles per middle rank instead
    of 45), and nothing is ever exchanged: every covering tile of an owned pixel is local, summed in the single-device order."""
    crops = np.asarray(crops)
    order = np.lexsort((crops[:, 0], crops[:, 1]))
    cuts = pixel_row_cuts(image_height, world)
    th = int(crops[0, 3])
    ys = crops[order, 1]
    return [order[(ys < cuts[r + 1]) & (ys + th > cuts[r])].astype(np.int64) for r in range(world)]


def _cover_rects(tiles: np.ndarray, r0: int, r1: int, th: int, tw: int):
    """The part of the union of ``tiles`` ((x, y, ...) rows) that lies on pixel rows ``r0:r1``, as rectangles ``(q0, q1, c0, c1)``: the
    rows are split at the tiles' edges and every run gets the column extent of the tiles that reach it (contiguous tile ranges of a
    row-major sequence: the extent is covered without holes); runs with equal extents are joined."""
    ys = tiles[:, 1]
    edges = sorted({r0, r1} | {int(v) for v in np.concatenate([ys, ys + th]) if r0 < v < r1})
    out = []
    for q0, q1 in zip(edges[:-1], edges[1:]):
        hit = (ys < q1) & (ys + th > q0)
        if not hit.any():
            continue
        c0, c1 = int(tiles[hit, 0].min()), int(tiles[hit, 0].max()) + tw
        if out and out[-1][1] == q0 and out[-1][2:] == (c0, c1):
            out[-1] = (out[-1][0], q1, c0, c1)
        else:
            out.append((q0, q1, c0, c1))
    return out


def _balanced_cut(upper: np.ndarray, lower: np.ndarray, y_s: int, lo: int, hi: int, th: int, tw: int, half: int) -> int:
    """The ownership cut between two neighbouring ranks (``upper`` / ``lower``: their crop rows; ``y_s``: the top of the lower rank's
    first tile): the row in ``lo..hi`` at which the partial sums the upper rank ships down (its tiles' area below the cut) and the ones
    the lower rank ships up (its tiles' area above it) are as EQUAL as the 4-row grid of the band kernel allows -- the two transfers
    share one full-duplex xGMI link, so the larger one is what an image waits for.  Round 5 cut at the top of the lower rank's first
    tile row when it starts one (everything one way) and half a tile below it when the boundary falls inside a tile row (then the rank
    with more tiles of that row shipped up to 18.9 MB at the headline geometry, the other 3 MB); balanced, no direction of any link
    carries more than 11.8 MB of the same ~22 MB total.  A position on the 64-row grid of the kernel's work items is preferred
    when it costs at most 1 % more."""
    def cost(c):
        down = sum((q1 - q0) * (c1 - c0) for q0, q1, c0, c1 in _cover_rects(upper, c, int(upper[:, 1].max()) + th, th, tw)) if c < int(upper[:, 1].max()) + th else 0
        up = sum((q1 - q0) * (c1 - c0) for q0, q1, c0, c1 in _cover_rects(lower, int(lower[:, 1].min()), c, th, tw)) if c > int(lower[:, 1].min()) else 0
        return max(down, up)

    grid = 4 if (lo % 4 == 0 and th % 4 == 0 and y_s % 4 == 0) else 1
    cands = [c for c in range(lo + (-lo) % grid, hi + 1, grid)] or [lo]
    costs = {c: cost(c) for c in cands}
    centre = y_s + half
    best = min(cands, key=lambda c: (costs[c], abs(c - centre)))
    on64 = [c for c in cands if c % 64 == 0 and costs[c] <= 1.01 * costs[best]]
    if on64:
        best = min(on64, key=lambda c: (costs[c], abs(c - centre)))
    return int(best)


def band_plan(crops: np.ndarray, world: int, image_height: int, partition: str = "tiles"):
    """Who accumulates, owns and exchanges what.  Per rank a dict with

    * ``tiles``: its tile indices in ISSUE order -- the tiles feeding an outgoing rectangle first, so that the exchange
      overlaps the accumulation of the others;
    * ``band`` = (a, b): pixel rows touched by its tiles;  ``owned`` = (o0, o1): the rows of the result it produces.
      The cut between consecutive ranks r, s is the row that BALANCES what the two ship to each other (``_balanced_cut``, round 6):
      the middle of the shared rows when s starts a tile row, and near half a tile below the top of s's first tile when the boundary
      falls inside a tile row -- moved towards the rank that holds more tiles of that row.  r holds the rows above the cut, s the rows
      below, and the halo rectangles travel in opposite directions of one full-duplex link at once;
    * ``sends`` = [(dst, r0, r1, c0, c1)], ``recvs`` = [(src, r0, r1, c0, c1)]: absolute pixel rectangles (rows r0:r1,
      columns c0:c1 = the column extent of the sender's tiles on those rows; a neighbour may get two -- the rows its cut takes
      from a full tile row and from the part of a tile row the sender holds) of the sender's partial sums that another
      rank owns.  ``boundary``: the tile indices that must be in before the sends are complete.

    Ranks without tiles have ``band`` = ``owned`` = None."""
    crops = np.asarray(crops)
    if partition == "pixel_rows":
        return _pixel_row_plan(crops, world, image_height)
    parts = PARTITIONS[partition](crops, world)
    tw, th = int(crops[0, 2]), int(crops[0, 3])
    step = np.diff(np.unique(crops[:, 1]))
    half = min(th // 2, int(step.min())) if len(step) else 0
    live = [r for r in range(world) if len(parts[r])]
    plan = [dict(rank=r, tiles=parts[r], band=None, owned=None, sends=[], recvs=[], boundary=np.zeros(0, dtype=np.int64)) for r in range(world)]
    for r in live:
        ys = crops[parts[r], 1]
        plan[r]["band"] = (int(ys.min()), int(ys.max()) + th)
    cut = [0]
    for r, s in zip(live[:-1], live[1:]):
        y_s = int(crops[parts[s][0], 1])
        cut.append(_balanced_cut(crops[parts[r]], crops[parts[s]], y_s, max(cut[-1], y_s), min(image_height, y_s + th), th, tw, half))
    cut.append(image_height)
    for i, r in enumerate(live):
        plan[r]["owned"] = (cut[i], cut[i + 1])
    for s in live:  # every part of s's band that another rank owns travels to that owner
        a, b = plan[s]["band"]
        mine = crops[parts[s]]
        feeding = np.zeros(len(mine), dtype=bool)
        for d in live:
            o0, o1 = plan[d]["owned"]
            r0, r1 = max(a, o0), min(b, o1)
            if d == s or r0 >= r1:
                continue
            for q0, q1, c0, c1 in _cover_rects(mine, r0, r1, th, tw):      # (tight per run of rows: the columns of the tiles that reach them)
                plan[s]["sends"].append((d, q0, q1, c0, c1))
                plan[d]["recvs"].append((s, q0, q1, c0, c1))
            feeding |= (mine[:, 1] < r1) & (mine[:, 1] + th > r0)
        plan[s]["boundary"] = parts[s][feeding]
        plan[s]["tiles"] = np.concatenate([parts[s][feeding], parts[s][~feeding]])
    return plan


def _pixel_row_plan(crops, world, image_height):
    """``band_plan`` of ``partition="pixel_rows"``: no sends, no receives, overlapping tile sets.  A rank whose rows no tile touches
    hands them to the live rank above it (or below, for the first ranks): the owned rows of the live ranks tile the image."""
    parts = pixel_row_partition(crops, world, image_height)
    cuts = pixel_row_cuts(image_height, world)
    th = int(crops[0, 3])
    plan = [dict(rank=r, tiles=parts[r], band=None, owned=None, sends=[], recvs=[], boundary=np.zeros(0, dtype=np.int64)) for r in range(world)]
    live = [r for r in range(world) if len(parts[r])]
    for i, r in enumerate(live):
        ys = crops[parts[r], 1]
        plan[r]["band"] = (int(ys.min()), int(ys.max()) + th)
        o0 = 0 if i == 0 else cuts[r]
        o1 = image_height if i == len(live) - 1 else cuts[live[i + 1]]
        plan[r]["owned"] = (int(o0), int(o1))
    return plan


class _HipOps:
    """The device operations the sharded merger needs, bound to the HIP library (tests inject a CPU stand-in)."""

    @staticmethod
    def new_local(shape, channels, weight, device):
        from .inference.tiles import TileMerger

        return TileMerger(shape, channels, weight, device=device, auto_plan=False)   # (a band accumulator: fed rank-local crops)

    @staticmethod
    def merge_rows(image, norm, out, extra=None, extra_rows=0):
        """out[c] = (image[c] + extra[c] on the first ``extra_rows`` rows) / norm; image [C,h,W] may be a row-slice view."""
        from . import _native as N

        lib = N.load()
        C, h, W = image.shape
        dev = image.device
        with N.on_device(dev):
            rc = lib.ptb_merge_div_ex(image.data_ptr(), norm.data_ptr(), out.data_ptr(), C, h * W, image.stride(0), out.stride(0),
                                      extra.data_ptr() if extra is not None else None,
                                      extra.stride(0) if extra is not None else 0, extra_rows * W, N.stream_ptr(dev))
        N.bump()
        N.check(rc, "ptb_merge_div_ex")
        return out


    @staticmethod
    def add_rect(image, top, rect, buf):
        """image [C, h, W] (rows offset by ``top``) += packed ``buf`` [C, r1 - r0, c1 - c0] on the absolute rectangle."""
        from . import _native as N

        r0, r1, c0, c1 = rect
        dst = image[:, r0 - top:r1 - top, c0:c1]
        dev = image.device
        with N.on_device(dev):
            rc = N.load().ptb_rect_add(dst.data_ptr(), buf.data_ptr(), image.shape[0], r1 - r0, c1 - c0, image.stride(0), image.stride(1),
                                       N.stream_ptr(dev))
        N.bump()
        N.check(rc, "ptb_rect_add")


def deferred_geometry(plan, rank: int, crops: np.ndarray, image_height: int):
    """Rows of rank ``rank``'s band it can finish on its own, and where its band plan has to cut.

    Returns ``(final, cuts)`` in ABSOLUTE rows: ``final = (f0, f1)`` = the longest run of rows of the rank's band that no other
    rank's tile touches, clipped to the rows the rank owns (there the kernels write ``sum / norm`` straight away; everywhere else
    they leave un-normalised partial sums that are exchanged and / or completed in ``merge()``); ``cuts`` = row positions
    that must be band edges and launch boundaries: the ends of ``final``, of the owned rows and of every send / receive
    rectangle (so that the rows a neighbour waits for are finished by their own, early launch)."""
    me = plan[rank]
    if me["band"] is None:
        return (0, 0), []
    crops = np.asarray(crops)
    th = int(crops[0, 3])
    a, b = me["band"]
    o0, o1 = me["owned"]
    others = np.zeros(image_height + 1, dtype=bool)
    for r, p in enumerate(plan):
        if r == rank or p["band"] is None:
            continue
        for y in np.unique(crops[p["tiles"], 1]):
            others[int(y):int(y) + th] = True
    free = ~others[a:b]
    best, start = (0, 0), None
    for i, f in enumerate(np.append(free, False)):      # longest run of rows nobody else touches
        if f and start is None:
            start = i
        elif not f and start is not None:
            if i - start > best[1] - best[0]:
                best = (start, i)
            start = None
    f0, f1 = max(a + best[0], o0), min(a + best[1], o1)
    final = (f0, f1) if f1 > f0 else (0, 0)
    cuts = {o0, o1, final[0], final[1]}
    for _peer, r0, r1, _c0, _c1 in list(me["sends"]) + list(me["recvs"]):
        cuts.update((r0, r1))
    return final, sorted(int(c) for c in cuts)


def early_spans(me, top: int = 0):
    """The row runs (relative to ``top``) of a rank's outgoing rectangles, touching runs joined: the rows a neighbour waits for.  Each is
    an `early` range of the rank's band plan (``ptb_band_plan_create2``): one launch of its own as soon as the tiles feeding it are in."""
    spans = []
    for a_, b_ in sorted({(int(r0) - top, int(r1) - top) for _d, r0, r1, _c0, _c1 in me["sends"]}):
        if spans and a_ <= spans[-1][1]:      # (two rectangles for one neighbour: the rows its cut takes from a full tile row and from a part of one)
            spans[-1] = (spans[-1][0], max(spans[-1][1], b_))
        else:
            spans.append((a_, b_))
    return spans


class _DeferredBand:
    """One rank's band merged without an accumulator: the C band plan of ``TileMerger(defer=True)`` over the rank's own tiles
    (csrc/ptb_bandplan.hip), in the rank's issue order and local row coordinates.  ``out`` [C, rows, W] receives ``sum / norm``
    on the rows the rank finishes alone and un-normalised partial sums on the rows it shares with (or hands to) a neighbour --
    what is exchanged are those partial sums instead of accumulator rectangles, and no accumulator read-modify-write happens
    at all.  The model outputs handed in are held (by reference) until the image is merged."""

    def __init__(self, handle, table, groups, out_shape, norm, weight, xy_abs, top, channels, th, tw):
        self.handle, self.table, self.groups = handle, table, groups      # groups: [(y0, y1, last tile)] local rows
        self.out_shape, self.norm, self.weight = out_shape, norm, weight
        self.out = None              # [C, rows, W] of THIS image: allocated with its first batch, handed to the caller by merge()
        self.launched = 0
        self.xy_abs, self.top = xy_abs, top
        self.xy_abs_rows = np.ascontiguousarray(xy_abs.T)      # [N, 2] (x, y), the layout callers' crop rows compare against
        self._varr = {}
        self.channels, self.th, self.tw = channels, th, tw
        self.pos = 0
        from .inference._merge_modes import HeldBatches      # (the custody contract of TileMerger(defer=True), shared)

        self.held = HeldBatches("ShardedTileMerger(defer=True)")
        self.cfg = None

    def __del__(self):
        try:
            from . import _native as N

            if self.handle:
                N.load().ptb_band_plan_destroy(self.handle)
                self.handle = None
        except Exception:  # noqa: BLE001
            pass

    @staticmethod
    def build(merger, crops, weight, rows, send_buf, recv_buf, shared=None):
        """``send_buf`` / ``recv_buf``: the packed rectangles of the image slot this plan serves; ``shared``: another slot's
        ``_DeferredBand`` of the same merger (its normaliser and window are reused)."""
        import ctypes

        from . import _native as N

        me = merger.plan[merger.rank]
        W = merger.image_width
        th, tw = int(crops[0, 3]), int(crops[0, 2])
        o0, o1 = merger.owned_rows
        if merger.partition == "pixel_rows":
            # communication-free: the plan's rows are exactly the owned pixel rows; tiles hang over both ends and are clipped
            top, bottom = o0, o1
            final, cuts, flags = (o0, o1), [], 1
        else:
            top, bottom = merger.top, merger.bottom
            final, cuts = deferred_geometry(merger.plan, merger.rank, crops, merger.image_height)
            flags = 0
        mine = np.ascontiguousarray(crops[me["tiles"], :2].T.astype(np.int64))        # [2, n] absolute, issue order
        local = mine.copy()
        local[1] -= top
        cut_arr = np.ascontiguousarray(np.array([c - top for c in cuts if top < c < bottom], dtype=np.int64))
        # the rows neighbours wait for (the outgoing rectangles) form ONE early launch group, everything else is merged in groups of
        # `rows` rows that ignore the cuts: 2 launches per rank at N = 8 instead of 6 (ptb_band_plan_create2)
        spans = early_spans(me, top)
        early = np.ascontiguousarray(np.array(spans, dtype=np.int64).reshape(-1)) if (spans and merger.two_phase) else np.zeros(0, dtype=np.int64)
        lib = N.load()
        handle = ctypes.c_void_p()
        nbytes = lib.ptb_band_plan_create3(local[0].ctypes.data_as(N._i64p), local[1].ctypes.data_as(N._i64p), local.shape[1], merger.channels,
                                           th, tw, bottom - top, W, int(rows), final[0] - top if final[1] > final[0] else 0,
                                           final[1] - top if final[1] > final[0] else 0,
                                           cut_arr.ctypes.data_as(N._i64p) if len(cut_arr) else None, len(cut_arr),
                                           early.ctypes.data_as(N._i64p) if len(early) else None, len(early) // 2, flags, ctypes.byref(handle))
        if nbytes < 0:
            return None
        dev = merger.device
        table = torch.empty(max(int(nbytes), 64), dtype=torch.uint8, device=dev)
        with N.on_device(dev):
            rc = lib.ptb_band_plan_upload(handle, table.data_ptr(), N.stream_ptr(dev))
        N.bump()
        N.check(rc, "ShardedTileMerger (deferred band plan)")
        ng = ctypes.c_int()
        lib.ptb_band_plan_info(handle, ctypes.byref(ng), None, None, None, None)
        rows_arr = np.zeros(3 * ng.value, dtype=np.int64)
        lib.ptb_band_plan_info(handle, None, None, None, None, rows_arr.ctypes.data_as(N._i64p))
        groups = [tuple(int(v) for v in rows_arr[3 * g:3 * g + 3]) for g in range(ng.value)]
        if shared is not None:
            norm, w = shared.norm, shared.weight
        else:
            norm = torch.zeros((1, bottom - top, W), device=dev, dtype=torch.float32)
            if merger.norm_owned is not None:
                norm[:, o0 - top:o1 - top] = merger.norm_owned
            w = torch.from_numpy(np.ascontiguousarray(weight, dtype=np.float32)).to(dev).reshape(1, th, tw).contiguous()
        band = _DeferredBand(handle, table, groups, (merger.channels, bottom - top, W), norm, w, mine, top, merger.channels, th, tw)
        band.final = final
        # the outgoing rectangles in the plan's rows + their send buffers, for ptb_band_plan_submit_rank (submit + pack in one C call)
        sends = merger.sends
        band.n_sends = len(sends)
        band.rects = np.ascontiguousarray(np.array([[r0 - top, r1 - top, c0, c1] for _d, r0, r1, c0, c1 in sends], dtype=np.int64).reshape(-1))
        band.send_ptrs = (ctypes.c_void_p * max(len(sends), 1))(*[b.data_ptr() for b in send_buf])
        band.packed = (ctypes.c_int * max(len(sends), 1))()
        band.all_packed = ctypes.c_int(0)
        # recorded (by the C call, on the raw handle) behind the pack of the last outgoing rectangle.  torch creates the hipEvent
        # lazily with the first record(): record once here so that `cuda_event` is a real handle (a NULL handle made
        # the C side skip its record and the communication stream wait for nothing)
        band.ready_event = torch.cuda.Event()
        with torch.cuda.device(dev):
            band.ready_event.record(torch.cuda.current_stream(dev))
        if band.n_sends and not band.ready_event.cuda_event:
            raise RuntimeError("ShardedTileMerger: could not create the pack-complete event of the halo exchange")
        # ... and what ptb_band_plan_finish_rank needs: the incoming rectangles + the owned row ranges that hold partial sums
        recvs = merger.recvs
        band.n_recvs = len(recvs)
        band.recv_rects = np.ascontiguousarray(np.array([[r0 - top, r1 - top, c0, c1] for _s, r0, r1, c0, c1 in recvs], dtype=np.int64).reshape(-1))
        band.recv_ptrs = (ctypes.c_void_p * max(len(recvs), 1))(*[b.data_ptr() for b in recv_buf])
        f0, f1 = final
        ranges = [(o0, o1)] if f1 <= f0 else [(o0, f0), (f1, o1)]
        ranges = [(a_ - top, b_ - top) for a_, b_ in ranges if b_ > a_]
        band.n_ranges = len(ranges)
        band.ranges = np.ascontiguousarray(np.array(ranges, dtype=np.int64).reshape(-1))
        band.fast = {}            # (group, reduction) -> (views array, number of views, reduction code)
        # ctypes views made once (a .ctypes.data_as per call costs ~3 us each)
        band.rects_p = band.rects.ctypes.data_as(N._i64p) if band.n_sends else None
        band.event_p = ctypes.c_void_p(band.ready_event.cuda_event) if band.n_sends else None
        band.all_packed_ref = ctypes.byref(band.all_packed)
        band.recv_rects_p = band.recv_rects.ctypes.data_as(N._i64p) if band.n_recvs else None
        band.ranges_p = band.ranges.ctypes.data_as(N._i64p) if band.n_ranges else None
        return band

    def reset(self):
        from . import _native as N

        N.load().ptb_band_plan_reset(self.handle)
        self.pos, self.cfg, self.launched = 0, None, 0
        self.held.clear()
        for k in range(self.n_sends):
            self.packed[k] = 0
        self.all_packed.value = 1 if self.n_sends == 0 else 0
        self.out = None              # (the previous image's buffer now belongs to whoever merge() gave it to)

    def submit(self, batch, coords_abs, views, reduction):
        """Take the next planned tiles; returns the number of launches.  The tiles must arrive in ``merger.tiles`` order."""
        import ctypes

        from . import _native as N

        B = len(coords_abs)
        if self.pos + B > self.xy_abs.shape[1] or not np.array_equal(coords_abs[:, :2], self.xy_abs_rows[self.pos:self.pos + B]):
            raise RuntimeError("ShardedTileMerger: tiles must be integrated in the order of `merger.tiles` (the deferred band plan "
                               "was built for that order); construct the merger with defer=False for free-form accumulation")
        if batch.dtype not in N.DTYPE_CODES:
            batch = batch.float()
        if batch.requires_grad or not batch.is_contiguous():
            batch = batch.detach().contiguous()
        n_views = len(views) if views is not None else 1
        if batch.shape[0] != B * n_views or tuple(batch.shape[1:]) != (self.channels, self.th, self.tw):
            raise RuntimeError(f"tile batch of shape {tuple(batch.shape)} does not match {B} tiles x {n_views} views of "
                               f"[{self.channels}, {self.th}, {self.tw}]")
        key = tuple(views) if views is not None else None
        varr = self._varr.get(key)
        if varr is None:
            varr = self._varr[key] = N.int_array(list(views)) if views is not None else N.int_array([N.IDENT])
        per_tile = self.channels * self.th * self.tw
        dev = self.norm.device
        span = self.held.admit(batch, any(self.pos <= last < self.pos + B for _y0, _y1, last in self.groups))
        if self.out is None:
            self.out = torch.empty(self.out_shape, device=dev, dtype=torch.float32)
        with N.on_device(dev):
            # one C call: take the batch, launch the groups it completes, pack every outgoing rectangle whose rows are now written
            rc = N.load().ptb_band_plan_submit_rank(self.handle, self.pos, B, batch.data_ptr(), per_tile, B * per_tile, N.DTYPE_CODES[batch.dtype],
                                                    n_views, varr, reduction, self.out.data_ptr(), self.norm.data_ptr(), self.weight.data_ptr(),
                                                    self.n_sends, self.rects.ctypes.data_as(N._i64p), self.send_ptrs, self.packed,
                                                    self.event_p, ctypes.byref(self.all_packed),
                                                    N.stream_ptr(dev))
        N.bump()
        if rc < 0:
            N.check(rc, "ShardedTileMerger.integrate_batch (deferred band)")
        self.held.keep(batch, span)
        self.pos += B
        self.launched += rc
        return rc

    def submit_fast(self, batch, coords_abs, key, views, code):
        """The common call with everything per-call already validated by the caller (a contiguous device tensor of a kernel dtype, an
        int64 [B, 4] array of the next planned crops): one array compare + ONE C call."""
        from . import _native as N

        B, pos = coords_abs.shape[0], self.pos
        ent = self.fast.get(key)
        if ent is None:
            ent = self.fast[key] = (N.int_array(list(views)) if views is not None else N.int_array([N.IDENT]), len(views) if views is not None else 1)
        varr, n_views = ent
        if (pos + B > self.xy_abs.shape[1] or batch.shape != (B * n_views, self.channels, self.th, self.tw)
                or not np.array_equal(coords_abs[:, :2], self.xy_abs_rows[pos:pos + B])):
            return None          # (the general path reports what is wrong)
        due = False
        for _y0, _y1, last in self.groups:
            if pos <= last < pos + B:
                due = True
        span = self.held.admit(batch, due)
        dev = self.norm.device
        if self.out is None:
            self.out = torch.empty(self.out_shape, device=dev, dtype=torch.float32)
        per_tile = self.channels * self.th * self.tw
        with N.on_device(dev):
            rc = N.load().ptb_band_plan_submit_rank(self.handle, pos, B, batch.data_ptr(), per_tile, B * per_tile, N.DTYPE_CODES[batch.dtype],
                                                    n_views, varr, code, self.out.data_ptr(), self.norm.data_ptr(), self.weight.data_ptr(),
                                                    self.n_sends, self.rects_p, self.send_ptrs, self.packed, self.event_p, self.all_packed_ref,
                                                    N.stream_ptr(dev))
        N.bump()
        if rc < 0:
            N.check(rc, "ShardedTileMerger.integrate_batch (deferred band)")
        self.held.keep(batch, span)
        self.pos = pos + B
        self.launched += rc
        return rc

    def rows_launched(self, r0, r1):
        """Every launch group that writes absolute rows r0:r1 has been issued (its last tile is in)."""
        from . import _native as N

        return N.load().ptb_band_plan_rows_launched(self.handle, r0 - self.top, r1 - self.top) == 1

    def complete(self):
        return self.pos == self.xy_abs.shape[1]


class _StreamWork:
    """Handle of one posted exchange: ``wait()`` makes the CURRENT stream wait for exactly that exchange (an event recorded behind it
    on the communication stream), not for whatever else has been posted there since -- a pipelined merger posts the next image's
    exchange before it completes this one."""

    __slots__ = ("event", "device")

    def __init__(self, event, device):
        self.event, self.device = event, device

    def wait(self):
        torch.cuda.current_stream(self.device).wait_event(self.event)


class _HostStagedWork:
    """Point-to-point transfers of DEVICE rectangles under a backend that is not RCCL (gloo: the CPU tests' stand-in, also used with
    real kernels by the one-GPU process tests): staged through host tensors explicitly.  (torch's gloo send / recv take a tensor's raw
    data pointer; handed a device tensor they read and write HBM from a CPU thread, through the BAR, with no ordering against the
    stream that packs or consumes the rectangle -- an intermittent wrong answer, seen as a 1-in-10 failure of the pipelined process
    test.)  ``wait()``: the transfers are done and the received rectangles are in their device buffers, in stream order."""

    __slots__ = ("works", "host_recv", "dev_recv", "host_send")

    def __init__(self, works, host_recv, dev_recv, host_send):
        self.works, self.host_recv, self.dev_recv, self.host_send = works, host_recv, dev_recv, host_send

    def wait(self):
        for w in self.works:
            w.wait()
        for h, d in zip(self.host_recv, self.dev_recv):
            d.copy_(h)
        self.works, self.host_send = [], None


class RcclExchange:
    """An RCCL communicator of this library's own for the halo exchange (``ptb_halo_exchange``: all of a rank's sends and receives as
    one ncclGroup posted from C on a side stream), instead of ``torch.distributed.batch_isend_irecv``.  Collective: every rank of
    ``group`` constructs it (rank 0 creates the unique id, ``torch.distributed`` carries it to the others).  One per process and
    group is enough (``shared_exchange`` keeps one); hand it to every ``ShardedTileMerger(..., exchange=...)``."""

    def __init__(self, device, group=None, dist=None):
        import ctypes

        from . import _native as N

        if dist is None:
            import torch.distributed as dist
        lib = N.load()
        if not lib.ptb_rccl_available():
            raise RuntimeError("RcclExchange: no RCCL library could be bound (librccl.so.1)")
        self.device = torch.device(device)
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        ident = ctypes.create_string_buffer(128)
        if self.rank == 0:
            N.check(lib.ptb_rccl_unique_id(ident), "ptb_rccl_unique_id")
        box = [ident.raw]
        if self.world > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        ident = ctypes.create_string_buffer(box[0], 128)
        comm = ctypes.c_void_p()
        with N.on_device(self.device):
            N.check(lib.ptb_rccl_comm_init(ident, self.world, self.rank, ctypes.byref(comm)), "ptb_rccl_comm_init")
        self.comm = comm
        self.stream = torch.cuda.Stream(device=self.device)       # the exchange runs here, beside the merge kernels

    def post(self, sends, recvs, after_event=None):
        """sends / recvs: [(tensor, peer rank in the group)].  The communication stream waits for ``after_event`` when it was
        recorded (the pack of the last outgoing rectangle), else for everything queued on the current stream so far, then all
        transfers are posted as one group.  Returns a handle whose ``wait()`` joins the current stream with THIS exchange."""
        import ctypes

        from . import _native as N

        cur = torch.cuda.current_stream(self.device)
        if after_event is not None and getattr(after_event, "cuda_event", 0):
            self.stream.wait_event(after_event)
        else:
            self.stream.wait_stream(cur)

        def table(items):
            n = len(items)
            return ((ctypes.c_void_p * max(n, 1))(*[t.data_ptr() for t, _p in items]), (ctypes.c_int64 * max(n, 1))(*[t.numel() for t, _p in items]),
                    (ctypes.c_int * max(n, 1))(*[int(p) for _t, p in items]))

        sp, sc, sr = table(sends)
        rp, rc_, rr = table(recvs)
        with N.on_device(self.device):
            rc = N.load().ptb_halo_exchange(self.comm, len(sends), sp, sc, sr, len(recvs), rp, rc_, rr, ctypes.c_void_p(self.stream.cuda_stream))
        N.bump()
        N.check(rc, "ptb_halo_exchange")
        for t, _p in list(sends) + list(recvs):
            t.record_stream(self.stream)
        done = torch.cuda.Event()
        done.record(self.stream)
        return _StreamWork(done, self.device)

    def wait(self):
        """Join the current stream with everything posted so far."""
        torch.cuda.current_stream(self.device).wait_stream(self.stream)

    def close(self):
        from . import _native as N

        if self.comm:
            N.load().ptb_rccl_comm_destroy(self.comm)
            self.comm = None


_shared_exchanges = {}      # device index -> [(weak reference to the process group | None, RcclExchange | None)]  (None: tried, every rank fell back)
_last_exchange_error = None


def last_exchange_error():
    """Why the last ``shared_exchange`` call fell back to ``torch.distributed`` p2p (a string), or None."""
    return _last_exchange_error


def _group_object(group, dist):
    """The process-group OBJECT a cache entry belongs to (the default group when ``group`` is None); None when it cannot be named
    (a stand-in ``dist`` of the tests)."""
    if group is not None:
        return group
    try:
        return dist.distributed_c10d._get_default_group()
    except Exception:  # noqa: BLE001
        return None


def close_shared_exchanges():
    """Close and forget every cached ``RcclExchange`` (call it before ``destroy_process_group()``; entries whose group has died are
    also dropped whenever ``shared_exchange`` runs)."""
    for entries in _shared_exchanges.values():
        for _ref, ex in entries:
            if ex is not None:
                ex.close()
    _shared_exchanges.clear()


def shared_exchange(device, group=None, dist=None):
    """The process's ``RcclExchange`` for ``group`` on ``device``, created on first use -- COLLECTIVE, like constructing a
    ``ShardedTileMerger`` is: every rank of the group calls it.  Returns None (on every rank alike: the outcome is agreed by an
    all-reduce) when RCCL cannot be bound or the communicator cannot be set up on some rank; the merger then posts its rectangles
    with ``torch.distributed.batch_isend_irecv`` (``last_exchange_error()`` says why).  Entries are kept per process-group OBJECT
    through a weak reference and checked on every hit: a group that was destroyed takes its communicator with it, and a new group that
    happens to get the dead one's ``id()`` never sees a stale exchange."""
    global _last_exchange_error
    import weakref

    if dist is None:
        import torch.distributed as dist
    device = torch.device(device)
    dev_key = device.index if device.index is not None else torch.cuda.current_device()
    pg = _group_object(group, dist)
    entries = _shared_exchanges.setdefault(dev_key, [])
    alive = []
    for ref, ex in entries:
        if ref is not None and ref() is None:          # its group is gone: so is the communicator
            if ex is not None:
                ex.close()
            continue
        alive.append((ref, ex))
    entries[:] = alive
    for ref, ex in entries:
        if (ref() if ref is not None else None) is pg:
            return ex
    ex, err = None, None
    try:
        ex = RcclExchange(device, group=group, dist=dist)
    except Exception as exc:  # noqa: BLE001
        err = exc
    ok = torch.tensor([1 if ex is not None else 0], device=device, dtype=torch.int32)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    if int(ok.item()) == 0:
        if ex is not None:
            ex.close()
        from .inference.tiles import _warn_once

        _last_exchange_error = repr(err) if err is not None else "another rank of the group could not set up its communicator"
        _warn_once(("rccl-exchange",), f"ShardedTileMerger: the library's own RCCL communicator is not available on every rank ({err!r}); "
                                       "the halo exchange is posted with torch.distributed.batch_isend_irecv instead.")
        ex = None
    else:
        _last_exchange_error = None
    try:
        ref = weakref.ref(pg) if pg is not None else None
    except TypeError:      # (a group object that cannot be weakly referenced: keyed by the object itself, kept alive with its entry)
        ref = (lambda obj: (lambda: obj))(pg)
    entries.append((ref, ex))
    return ex


class _ImageSlot:
    """Everything of a ``ShardedTileMerger`` that belongs to ONE image in flight: the band accumulator or the deferred band plan,
    the packed send / receive rectangles and the handles of the posted exchange.  A pipelined merger alternates between two."""

    __slots__ = ("local", "deferred", "send_buf", "recv_buf", "pending", "exchanged", "remaining", "result", "ticket")

    def __init__(self):
        self.local = self.deferred = None
        self.send_buf, self.recv_buf, self.pending = [], [], []
        self.exchanged, self.remaining, self.result, self.ticket = False, {}, None, None


class PendingBand:
    """What ``ShardedTileMerger.merge_async()`` returns: this rank's band of an image whose halo exchange may still be in flight.
    ``result()`` completes it (adds the neighbours' partial sums, divides the shared rows) on the current stream and returns the
    ``[C, o1 - o0, W]`` band (None for a rank that owns no rows); idempotent."""

    __slots__ = ("_merger", "_slot", "_value", "_done")

    def __init__(self, merger, slot):
        self._merger, self._slot, self._value, self._done = merger, slot, None, slot is None

    @property
    def done(self) -> bool:
        return self._done

    def result(self):
        if not self._done:
            slot, self._slot = self._slot, None
            self._value = self._merger._complete(slot)
            self._done = True
            if slot.ticket is self:
                slot.ticket = None
        return self._value


class ShardedTileMerger:
    """Drop-in shaped like ``TileMerger`` for one rank of a sharded merge of ONE image.

    Every rank constructs it with the FULL ``crops`` of the slicer, then feeds only its own tiles -- ``self.tiles``
    (indices into ``crops``, in the order that lets the exchange overlap the accumulation) -- in absolute coordinates.
    ``merge()`` returns this rank's owned rows ``[C, o1 - o0, W]`` (``owned_rows`` gives the absolute range);
    ``gather()`` assembles the full map on every rank.  ``partition``: ``"tiles"`` (contiguous tile ranges, the
    reference's ``split_across_nodes`` rule; default), ``"rows"`` (whole tile rows) or ``"pixel_rows"`` (communication-free:
    every rank owns an equal share of the PIXEL rows and is fed every tile touching them -- boundary tiles are evaluated by
    both neighbours, nothing is exchanged, and the result equals the single-device merge bit for bit).  (A 2-D rank grid --
    ``partition="grid"`` of round 4 -- had no deferred band plan and was removed in round 5: 5.4x simulated at N = 8 against 6.6x.)

    Pipelining (a stream of images): ``merge_async()`` ends an image without waiting for its halo exchange and moves the merger
    on to a second set of buffers; the exchange of image i then runs beside the kernels of image i + 1 and is only joined when
    ``PendingBand.result()`` is called (or, at the latest, when image i + 2 needs the buffers back).  ``merge()`` stays the
    synchronous form (exchange joined inside the same image).
    """

    def __init__(self, image_shape, channels, weight, crops, device, group=None, ops=None, dist=None, partition="tiles", defer=False,
                 defer_rows=None, two_phase=True, exchange="auto", pipeline_depth=2):
        """``defer=True`` (opt-in, like ``TileMerger``): the rank's tiles are merged band by band straight from the model outputs
        (no accumulator).  The contract that comes with it: the batches are kept by reference and read by a LATER launch, so they
        must stay alive and unmodified until ``merge()`` (a reused output buffer or an in-place edit raises), and the tiles must
        be fed in ``self.tiles`` order.

        ``exchange``: ``"auto"`` (default) -- under the ``nccl`` backend (RCCL) on CUDA devices the rectangles travel as ONE
        ncclGroup posted from C on the library's own communicator (``shared_exchange``; collective at construction), anywhere
        else (gloo, CPU stand-ins) as ``torch.distributed.batch_isend_irecv``; ``"torch"`` / None force the latter; or an
        ``RcclExchange`` of the caller's."""
        if dist is None:
            import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.two_phase = bool(two_phase)     # deferred plan: one early launch for the rows neighbours wait for + the rest (else: cut by cut)
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.ops = ops or _HipOps
        self.device = torch.device(device)
        self.channels = channels
        self.partition = partition
        self.image_height, self.image_width = int(image_shape[0]), int(image_shape[1])
        crops = np.asarray(crops)
        self.plan = band_plan(crops, self.world, self.image_height, partition)
        me = self.plan[self.rank]
        self.tiles = me["tiles"]
        self.band = me["band"]
        self.owned_rows = me["owned"]
        self.sends, self.recvs = me["sends"], me["recvs"]
        self.exchange = self._resolve_exchange(exchange)
        self.pipeline_depth = max(1, int(pipeline_depth))
        self._slots, self._cur = [_ImageSlot()], 0
        self.images_async = 0        # images ended with merge_async() (diagnostics)
        if self.band is None:
            return
        a, b = self.band
        o0, o1 = self.owned_rows
        # the owned range may start above the band (rank 0 owns from row 0) or end below it (last rank): cover both
        self.top = min(a, o0)
        self.bottom = max(b, o1)
        # global normaliser of the owned rows: accumulate the window of EVERY tile touching them (data independent)
        th = int(crops[0, 3])
        self.norm_owned = None
        if o1 > o0:
            touching = crops[(crops[:, 1] < o1) & (crops[:, 1] + th > o0)]
            lo = int(min(touching[:, 1].min(), o0)) if len(touching) else o0
            hi = int(max((touching[:, 1] + th).max(), o1)) if len(touching) else o1
            tmp = self.ops.new_local((hi - lo, self.image_width), 1, weight, self.device)
            zeros = torch.zeros((8, 1, int(crops[0, 3]), int(crops[0, 2])), device=self.device)
            shifted = touching.copy()
            shifted[:, 1] -= lo
            for i in range(0, len(shifted), 8):
                tmp.integrate_batch(zeros[:len(shifted[i:i + 8])], shifted[i:i + 8])
            self.norm_owned = tmp.norm_mask[:, o0 - lo:o1 - lo].contiguous()
        # tiles (by origin) whose accumulation must finish before the outgoing rectangles are complete
        self._boundary = {}
        for x, y in crops[me["boundary"], :2]:
            self._boundary[(int(x), int(y))] = self._boundary.get((int(x), int(y)), 0) + 1
        self._crops, self._weight, self._defer_cfg = crops, weight, (bool(defer), defer_rows)
        self._warned_defer = False
        self._fill_slot(self._slots[0])
        self.reset()

    def _resolve_exchange(self, exchange):
        if exchange is None or exchange == "torch":
            return None
        if exchange != "auto":
            return exchange
        import os

        dist = self.dist
        if (os.environ.get("PTB_EXCHANGE", "auto") == "torch" or self.ops is not _HipOps or self.device.type != "cuda" or self.world < 2
                or not hasattr(dist, "get_backend")):
            return None
        try:
            if str(dist.get_backend(self.group)).lower() != "nccl":
                return None
        except Exception:  # noqa: BLE001
            return None
        return shared_exchange(self.device, self.group, dist)

    def _fill_slot(self, slot):
        """Buffers of one image in flight (the second set is only made when ``merge_async()`` is first used)."""
        crops, weight = self._crops, self._weight
        defer, defer_rows = self._defer_cfg
        o0, o1 = self.owned_rows
        channels = self.channels
        slot.send_buf = [torch.empty((channels, r1 - r0, c1 - c0), device=self.device) for _d, r0, r1, c0, c1 in self.sends]
        slot.recv_buf = [torch.empty((channels, r1 - r0, c1 - c0), device=self.device) for _s, r0, r1, c0, c1 in self.recvs]
        # Deferred band merging (opt-in): the rank's tiles are merged band by band straight from the model outputs
        # -- no accumulator read-modify-write, partial sums instead of accumulator rectangles on the rows shared with neighbours.
        # Needs the tiles in `self.tiles` order and a geometry on the 4-pixel grid; otherwise the incremental path below is used.
        if defer and self.ops is _HipOps and o1 > o0:
            from .inference.tiles import _defer_rows_default, _warn_once

            shared = next((s.deferred for s in self._slots if s.deferred is not None), None)
            slot.deferred = _DeferredBand.build(self, crops, weight, defer_rows if defer_rows is not None else _defer_rows_default(),
                                                slot.send_buf, slot.recv_buf, shared=shared)
            if slot.deferred is None:
                _warn_once(("sharded-defer",), "ShardedTileMerger(defer=True): the band plan does not take this geometry (tile origins / "
                                               "ownership cuts off the 4-pixel grid, more than 4 tiles over a pixel); using the "
                                               "incremental accumulate + exchange path.")
        if slot.deferred is None:
            slot.local = self.ops.new_local((self.bottom - self.top, self.image_width), channels, weight, self.device)
        else:
            slot.local = slot.deferred      # (truthy marker: this rank has a band)

    # ------------------------------------------------------------------ the current image's slot, under the names the code grew up with
    @property
    def local(self):
        return self._slots[self._cur].local

    @property
    def _deferred(self):
        return self._slots[self._cur].deferred

    @property
    def _send_buf(self):
        return self._slots[self._cur].send_buf

    @property
    def _recv_buf(self):
        return self._slots[self._cur].recv_buf

    @property
    def _pending(self):
        return self._slots[self._cur].pending

    @_pending.setter
    def _pending(self, v):
        self._slots[self._cur].pending = v

    @property
    def _exchanged(self):
        return self._slots[self._cur].exchanged

    @_exchanged.setter
    def _exchanged(self, v):
        self._slots[self._cur].exchanged = v

    @property
    def _remaining(self):
        return self._slots[self._cur].remaining

    @_remaining.setter
    def _remaining(self, v):
        self._slots[self._cur].remaining = v

    @property
    def _result(self):
        return self._slots[self._cur].result

    @_result.setter
    def _result(self, v):
        self._slots[self._cur].result = v

    # ------------------------------------------------------------------ per-image cycle
    def reset(self):
        """Start a new image in the current buffers: zero the band accumulator and re-arm the exchange."""
        slot = self._slots[self._cur]
        if slot.ticket is not None:
            slot.ticket.result()         # an image ended with merge_async() still lives here: complete it first
        self._wait_pending(slot)
        slot.exchanged, slot.result = False, None
        if slot.local is None:
            return
        if slot.deferred is not None:
            slot.deferred.reset()
            slot.remaining = {}
            return
        if hasattr(slot.local, "reset"):
            slot.local.reset()          # first-touch accumulators: no memset
        else:
            slot.local.image.zero_()
            slot.local.norm_mask.zero_()
        slot.remaining = dict(self._boundary)

    def _shift(self, crop_coords):
        c = np.array(crop_coords.cpu() if torch.is_tensor(crop_coords) else crop_coords, dtype=np.int64).reshape(-1, 4).copy()
        origins = [(int(x), int(y)) for x, y in c[:, :2]]
        c[:, 1] -= self.top
        return c, origins

    def _after_integrate(self, origins):
        rem = self._remaining
        if rem:
            for o in origins:
                n = rem.get(o)
                if n is not None:
                    if n == 1:
                        del rem[o]
                    else:
                        rem[o] = n - 1
        if not rem and not self._exchanged:
            self._start_exchange()

    def _submit_deferred(self, batch, crop_coords, views, code):
        from . import _native as N

        N.require_device(batch, "ShardedTileMerger")
        coords = np.array(crop_coords.cpu() if torch.is_tensor(crop_coords) else crop_coords, dtype=np.int64).reshape(-1, 4)
        launched = self._deferred.submit(batch, coords, views, code)
        if launched and not self._exchanged and self._deferred.all_packed.value:
            self._start_exchange()      # every outgoing rectangle has been written by its launch and packed (in the same C call)

    def integrate_batch(self, batch, crop_coords):
        if len(batch) != len(crop_coords):
            raise ValueError("Number of images in batch does not correspond to number of coordinates")
        from .inference import _lazy

        if type(batch) is _lazy.LazyDeaugment and batch.dtype != torch.float32:
            batch = batch._evaluate()                   # (half-precision handles -- a half tensor, rounded once -- are not fused on the sharded path)
        if type(batch) is _lazy.LazyDeaugment:      # integrate_batch(tta.<group>_image_deaugment(y), crops): fused, like TileMerger
            taken = batch._take_source()
            if taken is not None:
                _lazy.fused += 1
                from .inference._views import REDUCTION_NAMES

                return self.integrate_batch_deaugment(taken[0], crop_coords, group=taken[1], reduction=REDUCTION_NAMES[taken[3]])
        if self._deferred is not None:
            from . import _native as N

            return self._submit_deferred(batch, crop_coords, None, N.RED_SUM)
        c, origins = self._shift(crop_coords)
        self.local.integrate_batch(batch, c)
        self._after_integrate(origins)

    def integrate_batch_deaugment(self, batch, crop_coords, group="d4", reduction="mean"):
        if self._deferred is not None:
            from . import _native as N
            from .inference.tta import DEAUGMENT_VIEWS, _reduction_code

            d = self._deferred
            if (type(crop_coords) is np.ndarray and crop_coords.ndim == 2 and crop_coords.dtype == np.int64 and type(reduction) is str
                    and batch.is_cuda and batch.is_contiguous() and not batch.requires_grad and batch.dtype in N.DTYPE_CODES and len(crop_coords)):
                code = _reduction_code(reduction)
                views = DEAUGMENT_VIEWS.get(group)
                if code is not None and views is not None:
                    launched = d.submit_fast(batch, crop_coords, (group, code), views, code)
                    if launched is not None:
                        if launched and not self._exchanged and d.all_packed.value:
                            self._start_exchange()
                        return

            views = DEAUGMENT_VIEWS[group]
            if len(batch) != len(crop_coords) * len(views):
                raise ValueError("Number of images in batch does not correspond to number of coordinates x views")
            code = _reduction_code(reduction)
            if code is None:
                raise ValueError(f"reduction={reduction!r} cannot be fused into the tile merge")
            return self._submit_deferred(batch, crop_coords, list(views), code)
        c, origins = self._shift(crop_coords)
        self.local.integrate_batch_deaugment(batch, c, group=group, reduction=reduction)
        self._after_integrate(origins)

    def _rect(self, r0, r1, c0, c1, slot=None):
        """View of the band accumulator on an absolute pixel rectangle, valid to READ now: blocks of the rectangle no
        kernel has written yet are zero-filled first (only those -- the interior keeps its first-touch state)."""
        slot = slot or self._slots[self._cur]
        if slot.deferred is not None:      # partial sums written by the band launches (the caller checked rows_launched)
            d = slot.deferred
            return d.out[:, r0 - d.top:r1 - d.top, c0:c1]
        loc = slot.local
        if hasattr(loc, "_zero_fresh"):
            loc._zero_fresh(r0 - self.top, r1 - self.top, c0, c1)
            img = loc._image
        else:
            img = loc.image
        return img[:, r0 - self.top:r1 - self.top, c0:c1]

    def _start_exchange(self):
        """Post all halo sends / receives of the current image as one batch (one ncclGroup: every pair progresses concurrently,
        each on its own xGMI link, both directions of a link at once) on the communication stream; the caller's stream keeps
        accumulating the remaining tiles -- and, after ``merge_async()``, the next image."""
        slot = self._slots[self._cur]
        slot.exchanged = True
        if slot.local is None:
            return
        dist = self.dist
        ops = []
        d = slot.deferred
        if not self.sends and not self.recvs:
            return
        if self.exchange is not None:
            for k, (buf, (_dst, r0, r1, c0, c1)) in enumerate(zip(slot.send_buf, self.sends)):
                if d is None or not d.packed[k]:
                    buf.copy_(self._rect(r0, r1, c0, c1))
            all_packed_in_c = d is not None and d.n_sends and all(d.packed[k] for k in range(d.n_sends))
            work = self.exchange.post([(buf, dst) for buf, (dst, *_r) in zip(slot.send_buf, self.sends)],
                                      [(buf, src) for buf, (src, *_r) in zip(slot.recv_buf, self.recvs)],
                                      after_event=d.ready_event if all_packed_in_c else None)
            slot.pending = [work if work is not None else self.exchange]
            return
        staged = self.device.type == "cuda" and self._p2p_through_host()
        host_send, host_recv = [], []
        for k, (buf, (dst, r0, r1, c0, c1)) in enumerate(zip(slot.send_buf, self.sends)):
            if d is None or not d.packed[k]:
                buf.copy_(self._rect(r0, r1, c0, c1))     # pack the strided rectangle (its tiles are all in)
            if staged:
                buf = buf.cpu()                           # (stream-ordered behind the pack, done when it returns)
                host_send.append(buf)
            ops.append(dist.P2POp(dist.isend, buf, self._global_rank(dst), self.group))
        for buf, (src, *_rect) in zip(slot.recv_buf, self.recvs):
            if staged:
                buf = torch.empty(buf.shape, dtype=buf.dtype)
                host_recv.append(buf)
            ops.append(dist.P2POp(dist.irecv, buf, self._global_rank(src), self.group))
        if ops:
            works = dist.batch_isend_irecv(ops)
            slot.pending = [_HostStagedWork(works, host_recv, slot.recv_buf, host_send)] if staged else works

    def _p2p_through_host(self):
        """Device rectangles under a backend other than RCCL travel through host tensors (see ``_HostStagedWork``)."""
        cached = getattr(self, "_p2p_host", None)
        if cached is None:
            try:
                cached = str(self.dist.get_backend(self.group)).lower() != "nccl"
            except Exception:  # noqa: BLE001  (a stand-in without backends: whatever it does with the tensors is its business)
                cached = False
            self._p2p_host = cached
        return cached

    def _global_rank(self, r):
        if self.group is None:
            return r
        return self.dist.get_global_rank(self.group, r)

    def _wait_pending(self, slot=None):
        slot = slot or self._slots[self._cur]
        for w in slot.pending:
            w.wait()
        slot.pending = []

    def _end_of_image(self, slot):
        """Every tile of the image is in: make sure its exchange has been posted."""
        if slot.deferred is not None and not slot.deferred.complete():
            raise RuntimeError("ShardedTileMerger.merge(): not all of this rank's tiles were integrated")
        if not slot.exchanged:
            self._start_exchange()

    def _complete(self, slot):
        """Join the image's exchange (the current stream waits for it) and finish the owned rows; returns the band."""
        self._wait_pending(slot)
        o0, o1 = self.owned_rows
        if o1 <= o0:
            return None
        return self._merge_deferred(slot, o0, o1) if slot.deferred is not None else self._merge_incremental(slot, o0, o1)

    def merge(self):
        """This rank's owned rows of ``image / norm_mask`` as ``[C, o1 - o0, W]`` (None for a rank that owns no rows)."""
        slot = self._slots[self._cur]
        if slot.local is None:
            return None
        if slot.result is not None:       # a second merge() of the same image: the same tensor (nothing is added or divided twice)
            return slot.result
        self._end_of_image(slot)
        slot.result = self._complete(slot)
        return slot.result

    def merge_async(self) -> PendingBand:
        """End the current image WITHOUT joining its halo exchange and start the next one in the other set of buffers (no
        ``reset()`` needed).  The returned handle's ``result()`` joins the exchange and finishes the band; call it after the next
        image's tiles have been integrated (that is what hides the exchange) -- at the latest it is called for you when the
        image after next needs the buffers back.  All ranks must call ``merge_async()`` / ``merge()`` in the same sequence."""
        slot = self._slots[self._cur]
        if slot.local is None:
            return PendingBand(self, None)
        if slot.result is not None:
            raise RuntimeError("ShardedTileMerger.merge_async(): this image was already merged with merge(); call reset() first")
        self._end_of_image(slot)
        if slot.deferred is not None:
            slot.deferred.held.clear()       # every launch that reads the batches has been issued (stream order keeps their memory safe)
        ticket = slot.ticket = PendingBand(self, slot)
        self.images_async += 1
        # move on: the next image lives in the next slot (made on first use); whatever image still sits there is completed first
        if self.pipeline_depth < 2:
            ticket.result()
        else:
            nxt = (self._cur + 1) % self.pipeline_depth
            while len(self._slots) <= nxt:
                new = _ImageSlot()
                self._fill_slot(new)
                self._slots.append(new)
            self._cur = nxt
        self.reset()
        return ticket

    def _merge_incremental(self, slot, o0, o1):
        # the band accumulator, readable on the rows this rank owns and on every received rectangle (blocks there that no
        # kernel has written are zero-filled; rows owned by other ranks keep their first-touch state: nobody reads them)
        for _src, r0, r1, c0, c1 in [(None, o0, o1, 0, self.image_width)] + list(self.recvs):
            self._rect(r0, r1, c0, c1, slot)
        image = slot.local._image if hasattr(slot.local, "_zero_fresh") else slot.local.image
        extra, extra_rows = None, 0
        for buf, (_src, r0, r1, c0, c1) in zip(slot.recv_buf, self.recvs):
            if extra is None and r0 == o0 and c0 == 0 and c1 == self.image_width:
                extra, extra_rows = buf, r1 - r0     # a full-width strip at the top of the band: folded into the division
            else:
                self.ops.add_rect(image, self.top, (r0, r1, c0, c1), buf)
        out = torch.empty((self.channels, o1 - o0, self.image_width), device=self.device)
        return self.ops.merge_rows(image[:, o0 - self.top:o1 - self.top], self.norm_owned[0], out, extra, extra_rows)

    def _merge_deferred(self, slot, o0, o1):
        """Owned rows from the band plan's output: the rows finished alone already hold ``sum / norm``; the others hold this
        rank's partial sums, get the neighbours' partial sums added and are divided in place (<= 2 row ranges)."""
        from . import _native as N

        d = slot.deferred
        if not d.complete():
            raise RuntimeError("ShardedTileMerger.merge(): not all of this rank's tiles were integrated")
        if d.n_recvs or d.n_ranges:
            # add the neighbours' partial sums, divide the rows that held partial sums: one C call (ptb_rect_add + ptb_merge_div_ex launches)
            with N.on_device(self.device):
                rc = N.load().ptb_band_plan_finish_rank(d.handle, d.out.data_ptr(), d.norm.data_ptr(), d.n_recvs, d.recv_rects_p,
                                                        d.recv_ptrs if d.n_recvs else None, d.n_ranges, d.ranges_p, N.stream_ptr(self.device))
            N.bump()
            N.check(rc, "ShardedTileMerger.merge (deferred band)")
        d.held.clear()
        return d.out[:, o0 - d.top:o1 - d.top]

    def gather(self, band):
        """All-gather the bands into the full ``[C, H, W]`` map on every rank (optional; 52 MB per rank at cfg2)."""
        full = torch.empty((self.channels, self.image_height, self.image_width), device=self.device)
        for r in range(self.world):
            owned = self.plan[r]["owned"]
            if owned is None or owned[1] <= owned[0]:
                continue
            piece = band.contiguous() if r == self.rank else torch.empty((self.channels, owned[1] - owned[0], self.image_width), device=self.device)
            self.dist.broadcast(piece, self._global_rank(r), group=self.group)
            full[:, owned[0]:owned[1]] = piece
        return full


# ---------------------------------------------------------------------------------------------- batch-sharded losses
class _AllReduceSum(torch.autograd.Function):
    """Differentiable sum over the ranks of a process group: every rank receives the total; in backward every rank's
    input receives the sum of the ranks' upstream gradients (the adjoint of a replicated sum)."""

    @staticmethod
    def forward(ctx, x, group, dist):
        ctx.group, ctx.dist = group, dist
        out = x.clone()
        dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().clone()
        ctx.dist.all_reduce(g, op=ctx.dist.ReduceOp.SUM, group=ctx.group)
        return g, None, None


def all_reduce_sum(x: torch.Tensor, group=None, dist=None) -> torch.Tensor:
    """Sum of ``x`` over the ranks of ``group`` with autograd support (a [C]-sized tensor: one tiny RCCL all-reduce)."""
    if dist is None:
        import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return x
    return _AllReduceSum.apply(x, group, dist)


class sync_region_statistics:
    """Context manager: inside it ``DiceLoss`` / ``JaccardLoss`` (and ``FocalDiceJaccardLoss``) evaluate their per-class
    region sums over the batch of ALL ranks, so a batch sharded over the GPUs of a node gives the single-GPU loss value
    (per-rank partials -> one tiny all-reduce -> scalar epilogue).  Each rank's result is then the global
    loss; with DDP's gradient averaging, scale by the world size if the sum of the ranks' gradients is wanted.

        with sync_region_statistics():          # or sync_region_statistics(group)
            loss = dice(logits_shard, labels_shard)
    """

    _active = None

    def __init__(self, group=None, dist=None):
        self.group, self.dist = group, dist

    def __enter__(self):
        self._prev = sync_region_statistics._active
        sync_region_statistics._active = self
        return self

    def __exit__(self, *exc):
        sync_region_statistics._active = self._prev
        return False

    @staticmethod
    def apply(stats):
        ctx = sync_region_statistics._active
        if ctx is None:
            return stats
        return tuple(all_reduce_sum(s, ctx.group, ctx.dist) for s in stats)


# ---------------------------------------------------------------------------------------------- multiscale TTA over ranks
def _source_rows(out_r0: int, out_r1: int, h_in: int, h_out: int, align_corners: bool):
    """Source rows [s0, s1) the bilinear taps of output rows [out_r0, out_r1) touch (one row of slack on both sides:
    the device evaluates the tap positions in fp32)."""
    if h_in == h_out:
        return out_r0, out_r1
    if align_corners:
        scale = (h_in - 1) / (h_out - 1) if h_out > 1 else 0.0
        lo, hi = scale * out_r0, scale * (out_r1 - 1)
    else:
        scale = h_in / h_out
        lo, hi = max(scale * (out_r0 + 0.5) - 0.5, 0.0), max(scale * (out_r1 - 1 + 0.5) - 0.5, 0.0)
    return max(int(np.floor(lo)) - 1, 0), min(int(np.floor(hi)) + 3, h_in)


def ms_strip_plan(source_heights: Sequence[int], out_height: int, world: int, align_corners: bool = True):
    """Row-strip decomposition of ``ms_image_deaugment`` over ``world`` ranks (BASELINE cfg5 "4 x MI355X"):
    rank r produces output rows ``out[r] = (r0, r1)`` (``np.linspace`` cuts) and needs, of scale s, source rows
    ``src[r][s] = (s0, s1)``.  No collective is involved: when the model runs on row strips with that halo, every rank
    already holds what it needs; the strips of the result are simply concatenated (or stay sharded)."""
    cuts = np.linspace(0, out_height, world + 1, dtype=int)
    plan = []
    for r in range(world):
        r0, r1 = int(cuts[r]), int(cuts[r + 1])
        plan.append(dict(rank=r, out=(r0, r1), src=[_source_rows(r0, r1, int(h), out_height, align_corners) if r1 > r0 else (0, 0)
                                                     for h in source_heights]))
    return plan


def ms_image_deaugment_strip(strips, source_heights, src_rows, out_rows, out_size, reduction="mean", align_corners: bool = True):
    """This rank's rows ``out_rows = (r0, r1)`` of ``tta.ms_image_deaugment`` (bilinear, stride 1): ``strips[s]`` holds rows
    ``src_rows[s] = (s0, s1)`` of scale s's ``[B, C, source_heights[s], w_s]`` prediction.  One HIP launch; same
    arithmetic as the full-size call, so the concatenated strips equal it bit for bit."""
    import ctypes

    from . import _native as N
    from .inference.tta import _reduction_code

    code = _reduction_code(reduction)
    if code is None:
        raise NotImplementedError(f"reduction={reduction!r} has no fused multiscale kernel")
    first = strips[0]
    N.require_device(first, "multiscale TTA")
    B, C = int(first.shape[0]), int(first.shape[1])
    ms = []
    for m, (s0, s1) in zip(strips, src_rows):
        N.require_device(m, "multiscale TTA")
        if m.dim() != 4 or m.dtype != torch.float32 or m.shape[0] != B or m.shape[1] != C or m.shape[2] != s1 - s0:
            raise ValueError("every strip must be float32 [B, C, s1 - s0, w_s]")
        ms.append(m.contiguous())
    r0, r1 = int(out_rows[0]), int(out_rows[1])
    ho, wo = int(out_size[0]), int(out_size[1])
    out = torch.empty((B, C, r1 - r0, wo), device=first.device, dtype=torch.float32)
    if out.numel() == 0:
        return out
    ptrs = (ctypes.c_void_p * len(ms))(*[m.data_ptr() for m in ms])
    lib = N.load()
    with N.on_device(first.device):
        rc = lib.ptb_ms_deaug_reduce_strip(ptrs, N.int_array([int(h) for h in source_heights]), N.int_array([int(m.shape[3]) for m in ms]),
                                           N.int_array([int(s0) for s0, _ in src_rows]), N.int_array([int(s1 - s0) for s0, s1 in src_rows]),
                                           len(ms), out.data_ptr(), B * C, ho, wo, r0, r1 - r0, 1 if align_corners else 0, code,
                                           N.stream_ptr(first.device))
    N.bump()
    N.check(rc, "ptb_ms_deaug_reduce_strip")
    return out


def ms_flips_image_deaugment_strip(strips, source_heights, src_rows, out_rows, out_size, group="fliplr", inner_reduction="mean",
                                   reduction="mean", align_corners: bool = True):
    """This rank's rows ``out_rows = (r0, r1)`` of ``tta.ms_flips_image_deaugment`` -- multiscale TTA whose every scale is itself
    flip-augmented (BASELINE configs[4]) -- in ONE pass: ``strips[s]`` holds rows ``src_rows[s] = (s0, s1)`` of the model output of
    scale s for the ``<group>_image_augment``-ed input (``[V*B, C, s1 - s0, w_s]``, chunk-major).  Same arithmetic as the full-size
    call (the taps are computed against ``source_heights``), so the concatenated strips equal it bit for bit.  Groups whose views
    flip rows (flipud, flips, d2) do not come in row strips: compose ``<group>_image_deaugment`` + ``ms_image_deaugment_strip`` there."""
    import ctypes

    from . import _native as N
    from .inference.tta import DEAUGMENT_VIEWS, _reduction_code

    views = DEAUGMENT_VIEWS[group]
    inner, outer = _reduction_code(inner_reduction), _reduction_code(reduction)
    if inner is None or outer is None:
        raise NotImplementedError("the fused strip kernel takes string reductions")
    V = len(views)
    first = strips[0]
    N.require_device(first, "multiscale TTA")
    if first.shape[0] % V:
        raise RuntimeError(f"Input batch size ({first.size(0)}) must be divisible by {V}.")
    B, C = int(first.shape[0]) // V, int(first.shape[1])
    ms = []
    for m, (s0, s1) in zip(strips, src_rows):
        N.require_device(m, "multiscale TTA")
        if m.dim() != 4 or m.dtype != torch.float32 or m.shape[0] != V * B or m.shape[1] != C or m.shape[2] != s1 - s0:
            raise ValueError("every strip must be float32 [V*B, C, s1 - s0, w_s]")
        ms.append(m.contiguous())
    r0, r1 = int(out_rows[0]), int(out_rows[1])
    ho, wo = int(out_size[0]), int(out_size[1])
    out = torch.empty((B, C, r1 - r0, wo), device=first.device, dtype=torch.float32)
    if out.numel() == 0:
        return out
    ptrs = (ctypes.c_void_p * len(ms))(*[m.data_ptr() for m in ms])
    with N.on_device(first.device):
        rc = N.load().ptb_ms_flip_deaug_reduce_strip(ptrs, N.int_array([int(h) for h in source_heights]), N.int_array([int(m.shape[3]) for m in ms]),
                                                     N.int_array([int(s0) for s0, _ in src_rows]), N.int_array([int(s1 - s0) for s0, s1 in src_rows]),
                                                     len(ms), V, N.int_array(list(views)), inner, out.data_ptr(), B * C, ho, wo, r0, r1 - r0,
                                                     1 if align_corners else 0, outer, N.stream_ptr(first.device))
    N.bump()
    N.check(rc, "ptb_ms_flip_deaug_reduce_strip")
    return out
