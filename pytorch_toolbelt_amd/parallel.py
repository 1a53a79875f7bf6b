"""Multi-GPU tile merging: one image, tiles sharded over the ranks of one node (one process per GPU, RCCL over xGMI).

The reference has no multi-GPU tile path; the single-device ``TileMerger`` result is the specification
(SURVEY.md 8e).  Design for MI355X's point-to-point xGMI fabric:

* tiles are partitioned by **tile row** (the reference's ``split_across_nodes`` linspace rule applied to rows,
  utils/distributed.py:306-309), so each rank's tiles cover one horizontal band of the padded image;
* every rank accumulates its band locally with the fused HIP kernels (no communication on the data path);
* each rank **owns** the pixel rows from the top of its band to the top of the next rank's band.  The only exchange
  is the strip of its band that hangs into the next owner's rows (tile_size - tile_step rows, 21 MB for the headline
  config): one point-to-point send to the next rank and one receive from the previous one, all pairs concurrently on
  distinct xGMI links, overlapped with the accumulation of the remaining tile rows.  Nothing is all-reduced: a ring
  all-reduce of the 524 MB accumulator would be per-link bound and ~30x slower than the kernels;
* ``norm_mask`` is data independent, so every rank computes the global normaliser of its owned rows once, locally;
* ``merge()`` adds the received strip and divides in a single HIP kernel, returning this rank's band.
"""
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

__all__ = ["tile_row_partition", "band_plan", "ShardedTileMerger", "all_reduce_sum", "sync_region_statistics", "ms_strip_plan",
           "ms_image_deaugment_strip"]


def tile_row_partition(crops: np.ndarray, world: int) -> List[np.ndarray]:
    """Tile indices per rank: contiguous groups of tile rows, boundaries at ``np.linspace(0, n_rows, world+1, dtype=int)``.
    Within a rank the LAST tile row comes first (it produces the strip that must travel to the next rank, so the
    exchange overlaps the accumulation of the other rows)."""
    crops = np.asarray(crops)
    row_y = np.unique(crops[:, 1])
    cuts = np.linspace(0, len(row_y), world + 1, dtype=int)
    parts = []
    for r in range(world):
        ys = row_y[cuts[r]:cuts[r + 1]]
        order = []
        for y in list(ys[::-1][:1]) + list(ys[:-1]):
            order.extend(np.nonzero(crops[:, 1] == y)[0].tolist())
        parts.append(np.asarray(order, dtype=np.int64))
    return parts


def band_plan(crops: np.ndarray, world: int, image_height: int):
    """Per rank: band rows [a, b) touched by its tiles, owned rows [o0, o1), and the exchange lists.

    Returns a list of dicts(rank, tiles, band=(a,b), owned=(o0,o1), sends=[(dst, r0, r1)], recvs=[(src, r0, r1)]) with
    absolute pixel rows.  Ranks without tiles own nothing."""
    crops = np.asarray(crops)
    parts = tile_row_partition(crops, world)
    th = int(crops[0, 3])
    bands = []
    for p in parts:
        if len(p) == 0:
            bands.append(None)
        else:
            ys = crops[p, 1]
            bands.append((int(ys.min()), int(ys.max()) + th))
    live = [r for r in range(world) if bands[r] is not None]
    plan = []
    for r in range(world):
        plan.append(dict(rank=r, tiles=parts[r], band=bands[r], owned=None, sends=[], recvs=[]))
    for i, r in enumerate(live):
        o0 = bands[r][0] if i else 0
        o1 = bands[live[i + 1]][0] if i + 1 < len(live) else image_height
        plan[r]["owned"] = (o0, o1)
    for s in live:  # every part of s's band owned by another rank travels to that owner
        a, b = bands[s]
        for d in live:
            if d == s:
                continue
            o0, o1 = plan[d]["owned"]
            r0, r1 = max(a, o0), min(b, o1)
            if r0 < r1:
                plan[s]["sends"].append((d, r0, r1))
                plan[d]["recvs"].append((s, r0, r1))
    return plan


class _HipOps:
    """The device operations the sharded merger needs, bound to the HIP library (tests inject a CPU stand-in)."""

    @staticmethod
    def new_local(shape, channels, weight, device):
        from .inference.tiles import TileMerger

        return TileMerger(shape, channels, weight, device=device)

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


class ShardedTileMerger:
    """Drop-in shaped like ``TileMerger`` for one rank of a tile-row sharded merge.

    Every rank constructs it with the FULL ``crops`` of the slicer, then feeds only its own tiles
    (``tile_row_partition(crops, world)[rank]``, boundary row first) in absolute coordinates.  ``merge()`` returns this
    rank's owned rows ``[C, o1 - o0, W]`` (``owned_rows`` gives the absolute range); ``gather()`` assembles the full
    map on every rank.
    """

    def __init__(self, image_shape, channels, weight, crops, device, group=None, ops=None, dist=None):
        if dist is None:
            import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.ops = ops or _HipOps
        self.device = torch.device(device)
        self.channels = channels
        self.image_height, self.image_width = int(image_shape[0]), int(image_shape[1])
        crops = np.asarray(crops)
        self.plan = band_plan(crops, self.world, self.image_height)
        me = self.plan[self.rank]
        self.band = me["band"]
        self.owned_rows = me["owned"]
        self.sends, self.recvs = me["sends"], me["recvs"]
        self.local = None
        self._pending = []
        if self.band is None:
            return
        a, b = self.band
        o0, o1 = self.owned_rows
        # the owned range may start above the band (rank 0 owns from row 0) or end below it (last rank): cover both
        self.top = min(a, o0)
        self.bottom = max(b, o1)
        self.local = self.ops.new_local((self.bottom - self.top, self.image_width), channels, weight, self.device)
        # global normaliser of the owned rows: accumulate the window of EVERY tile touching them (data independent)
        th = int(crops[0, 3])
        touching = crops[(crops[:, 1] < o1) & (crops[:, 1] + th > o0)]
        lo = int(min(touching[:, 1].min(), o0))
        hi = int(max((touching[:, 1] + th).max(), o1))
        tmp = self.ops.new_local((hi - lo, self.image_width), 1, weight, self.device)
        zeros = torch.zeros((8, 1, int(crops[0, 3]), int(crops[0, 2])), device=self.device)
        shifted = touching.copy()
        shifted[:, 1] -= lo
        for i in range(0, len(shifted), 8):
            tmp.integrate_batch(zeros[:len(shifted[i:i + 8])], shifted[i:i + 8])
        self.norm_owned = tmp.norm_mask[:, o0 - lo:o1 - lo].contiguous()
        # tiles (by y) whose accumulation must finish before the outgoing strips are complete
        self._send_rows_y = set()
        for _d, r0, r1 in self.sends:
            for y in np.unique(crops[me["tiles"], 1]):
                if y < r1 and y + th > r0:
                    self._send_rows_y.add(int(y))
        self._tiles_per_y = {int(y): int(np.sum(crops[me["tiles"], 1] == y)) for y in np.unique(crops[me["tiles"], 1])}
        self._send_buf = [torch.empty((channels, r1 - r0, self.image_width), device=self.device) for _d, r0, r1 in self.sends]
        self._recv_buf = [torch.empty((channels, r1 - r0, self.image_width), device=self.device) for _s, r0, r1 in self.recvs]
        self.reset()

    # ------------------------------------------------------------------ per-image cycle
    def reset(self):
        """Start a new image: zero the band accumulator and re-arm the exchange."""
        self._wait_pending()
        self._exchanged = False
        if self.local is None:
            return
        if hasattr(self.local, "reset"):
            self.local.reset()          # first-touch accumulators: no memset
        else:
            self.local.image.zero_()
            self.local.norm_mask.zero_()
        self._remaining = {y: self._tiles_per_y[y] for y in self._send_rows_y}

    def _shift(self, crop_coords):
        c = np.array(crop_coords.cpu() if torch.is_tensor(crop_coords) else crop_coords, dtype=np.int64).reshape(-1, 4).copy()
        ys = c[:, 1].copy()
        c[:, 1] -= self.top
        return c, ys

    def _after_integrate(self, ys):
        for y in ys:
            y = int(y)
            if y in self._remaining:
                self._remaining[y] -= 1
                if self._remaining[y] == 0:
                    del self._remaining[y]
        if not self._remaining and not self._exchanged:
            self._start_exchange()

    def integrate_batch(self, batch, crop_coords):
        if len(batch) != len(crop_coords):
            raise ValueError("Number of images in batch does not correspond to number of coordinates")
        c, ys = self._shift(crop_coords)
        self.local.integrate_batch(batch, c)
        self._after_integrate(ys)

    def integrate_batch_deaugment(self, batch, crop_coords, group="d4", reduction="mean"):
        c, ys = self._shift(crop_coords)
        self.local.integrate_batch_deaugment(batch, c, group=group, reduction=reduction)
        self._after_integrate(ys)

    def _start_exchange(self):
        """Post all strip sends / receives as one batch (one ncclGroup: every pair progresses concurrently, each on
        its own xGMI link) on RCCL's stream; the caller's stream keeps accumulating the remaining tile rows."""
        self._exchanged = True
        if self.local is None:
            return
        dist = self.dist
        ops = []
        for buf, (dst, r0, r1) in zip(self._send_buf, self.sends):
            # pack the strided strip; its rows are complete (all tiles of the boundary row are in), so read the raw
            # accumulator and leave the still-untouched interior rows in their first-touch state
            buf.copy_(self._raw_image()[:, r0 - self.top:r1 - self.top])
            ops.append(dist.P2POp(dist.isend, buf, self._global_rank(dst), self.group))
        for buf, (src, _r0, _r1) in zip(self._recv_buf, self.recvs):
            ops.append(dist.P2POp(dist.irecv, buf, self._global_rank(src), self.group))
        if ops:
            self._pending = dist.batch_isend_irecv(ops)

    def _raw_image(self):
        return getattr(self.local, "_image", None) if hasattr(self.local, "_image") else self.local.image

    def _global_rank(self, r):
        if self.group is None:
            return r
        return self.dist.get_global_rank(self.group, r)

    def _wait_pending(self):
        for w in self._pending:
            w.wait()
        self._pending = []

    def merge(self):
        """This rank's owned rows of ``image / norm_mask`` as ``[C, o1 - o0, W]`` (None for a rank without tiles)."""
        if self.local is None:
            return None
        if not self._exchanged:
            self._start_exchange()
        self._wait_pending()
        o0, o1 = self.owned_rows
        img = self.local.image[:, o0 - self.top:o1 - self.top]  # (property: zero-fills anything never written)
        out = torch.empty((self.channels, o1 - o0, self.image_width), device=self.device)
        extra, extra_rows = None, 0
        if self.recvs:
            if len(self.recvs) > 1 or self.recvs[0][1] != o0:
                raise NotImplementedError("more than one rank overlaps these rows (tile_step < tile_size / 2 with one-row ranks)")
            extra, extra_rows = self._recv_buf[0], self.recvs[0][2] - self.recvs[0][1]
        return self.ops.merge_rows(img, self.norm_owned[0], out, extra, extra_rows)

    def gather(self, band):
        """All-gather the bands into the full ``[C, H, W]`` map on every rank (optional; 52 MB per rank at cfg2)."""
        full = torch.empty((self.channels, self.image_height, self.image_width), device=self.device)
        for r in range(self.world):
            owned = self.plan[r]["owned"]
            if owned is None:
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
