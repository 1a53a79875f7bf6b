"""CPU, world_size > 1 over gloo: the tile-row sharded merger (partitioning, strip exchange, band merge, gather).

The HIP kernels cannot run here, so the per-rank device operations are replaced by the numpy oracle through the
merger's ``ops`` seam; everything else (plan, ownership, point-to-point exchange, ordering) is the product code.
The result must equal the single-process merge of all tiles (the reference's TileMerger semantics)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import tiles_oracle as TO
from oracle import tta_oracle as AO
from pytorch_toolbelt_amd.parallel import ShardedTileMerger, band_plan, early_spans, tile_range_partition, tile_row_partition


class OracleLocal:
    """CPU stand-in for TileMerger backed by the numpy oracle (shares memory with the torch tensors)."""

    def __init__(self, shape, channels, weight, device):
        self.image = torch.zeros((channels, shape[0], shape[1]))
        self.norm_mask = torch.zeros((1, shape[0], shape[1]))
        self.weight = np.asarray(weight)[None].astype(np.float32)

    def _state(self):
        return dict(image=self.image.numpy(), norm_mask=self.norm_mask.numpy(), weight=self.weight)

    def integrate_batch(self, batch, coords):
        TO.merger_integrate(self._state(), batch.numpy(), coords)

    def integrate_batch_deaugment(self, batch, coords, group="d4", reduction="mean"):
        TO.merger_integrate(self._state(), AO.image_deaugment(batch.numpy(), group, reduction), coords)


class OracleOps:
    new_local = OracleLocal

    @staticmethod
    def merge_rows(image, norm, out, extra=None, extra_rows=0):
        total = image.clone()
        if extra is not None:
            total[:, :extra_rows] += extra
        out.copy_(total / norm)
        return out

    @staticmethod
    def add_rect(image, top, rect, buf):
        r0, r1, c0, c1 = rect
        image[:, r0 - top:r1 - top, c0:c1] += buf


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, shape, tile, step, C, group, partition, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        geom = TO.slicer_geometry(shape, tile, step)
        w = TO.pyramid_window(*tile)[0]
        crops = geom["crops"]
        V = {"d4": 8, "d2": 4, None: 1}[group]
        rng = np.random.default_rng(11)
        outs = rng.standard_normal((V, len(crops), C, *tile)).astype(np.float32)
        m = ShardedTileMerger(geom["target_shape"], C, w, crops, device="cpu", ops=OracleOps, partition=partition)
        mine = m.tiles
        assert sorted(mine.tolist()) == sorted({"tiles": tile_range_partition, "rows": tile_row_partition}[partition](crops, world)[rank].tolist())
        for image_no in range(2):  # two images back to back: reset() must re-arm the exchange
            m.reset()
            for b0 in range(0, len(mine), 3):
                idx = mine[b0:b0 + 3]
                if group is None:
                    m.integrate_batch(torch.from_numpy(outs[0, idx] * (image_no + 1)), crops[idx])
                else:
                    batch = np.concatenate([outs[k, idx] for k in range(V)]) * (image_no + 1)
                    m.integrate_batch_deaugment(torch.from_numpy(batch), crops[idx], group=group)
            band = m.merge()
            assert (band is None) == (len(mine) == 0 or m.owned_rows[1] <= m.owned_rows[0])
            full = m.gather(band)
        if rank == 0:
            q.put(full.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,shape,tile,step,C,group,partition", [
    (2, (300, 200), (64, 64), (32, 32), 2, "d4", "tiles"),    # 9 x 6 tiles: the boundary falls inside a tile row
    (3, (300, 200), (64, 64), (32, 32), 2, None, "tiles"),
    (3, (300, 200), (64, 64), (32, 32), 2, None, "rows"),
    (2, (200, 260), (48, 80), (48, 40), 1, "d2", "rows"),     # no vertical overlap: nothing to exchange
    (2, (200, 260), (48, 80), (48, 40), 1, "d2", "tiles"),    # ... but a mid-row boundary still swaps half tiles
    (4, (150, 100), (64, 64), (32, 32), 1, None, "rows"),     # 4 tile rows for 4 ranks: every rank is a boundary rank
    (4, (90, 200), (64, 64), (32, 32), 1, None, "tiles"),     # 2 x 6 tiles over 4 ranks: 3 ranks inside one tile row
    (3, (100, 100), (64, 64), (16, 16), 1, None, "tiles"),    # step < tile / 2: a pixel is covered by 3 ranks
    (4, (40, 40), (64, 64), (32, 32), 1, None, "tiles"),      # one tile, four ranks: three of them idle
])
def test_sharded_equals_single(world, shape, tile, step, C, group, partition):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, shape, tile, step, C, group, partition, q)) for r in range(world)]
    for p in procs:
        p.start()
    full = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    geom = TO.slicer_geometry(shape, tile, step)
    w = TO.pyramid_window(*tile)[0]
    V = {"d4": 8, "d2": 4, None: 1}[group]
    rng = np.random.default_rng(11)
    outs = rng.standard_normal((V, len(geom["crops"]), C, *tile)).astype(np.float32) * 2
    st = TO.merger_new(geom["target_shape"], C, w)
    red = outs[0] if group is None else AO.image_deaugment(np.concatenate(list(outs)), group, "mean")
    TO.merger_integrate(st, red, geom["crops"])
    np.testing.assert_allclose(full, TO.merger_merge(st), rtol=0, atol=1e-5)


def test_partition_and_plan():
    geom = TO.slicer_geometry((5000, 5000), 512, 256)
    crops = geom["crops"]
    # whole tile rows: 19 rows over 8 ranks
    parts = tile_row_partition(crops, 8)
    assert sorted(np.concatenate(parts).tolist()) == list(range(361))
    assert max(len(p) for p in parts) == 57 and min(len(p) for p in parts) == 38   # 3 or 2 tile rows of 19
    plan = band_plan(crops, 8, 5120, "rows")
    rows = []
    for r, p in enumerate(plan):
        o0, o1 = p["owned"]
        rows.append((o0, o1))
        # round 6: the ownership cut lies in the MIDDLE of the 256 rows two neighbours share -- each ships 128 rows to the other
        # (10.5 MB in either direction of the link) instead of one of them shipping all 256 one way
        down = [(r + 1, o1, o1 + 128, 0, 5120)] if r < 7 else []
        up = [(r - 1, o0 - 128, o0, 0, 5120)] if r > 0 else []
        assert p["sends"] == up + down, (r, p["sends"])
        assert p["recvs"] == [(r - 1, o0, o0 + 128, 0, 5120)] * (r > 0) + [(r + 1, o1 - 128, o1, 0, 5120)] * (r < 7)
        if 0 < r < 7:
            assert o0 == int(crops[parts[r][0], 1]) + 128
        # the tiles feeding an outgoing strip are issued first: the rank's first and last tile rows
        nb = len(p["boundary"])
        assert nb == 19 * (len(up) + len(down)) and set(crops[p["tiles"][:nb], 1].tolist()) == (
            ({int(crops[parts[r], 1].min())} if up else set()) | ({int(crops[parts[r], 1].max())} if down else set()))
    assert rows[0][0] == 0 and rows[-1][1] == 5120 and all(rows[i][1] == rows[i + 1][0] for i in range(7))
    # contiguous tile ranges (the reference's split_across_nodes rule): 45 or 46 tiles each
    parts = tile_range_partition(crops, 8)
    assert np.concatenate(parts).tolist() == list(range(361)) and sorted({len(p) for p in parts}) == [45, 46]
    plan = band_plan(crops, 8, 5120)
    cover = np.zeros(5120, dtype=int)
    for r, p in enumerate(plan):
        o0, o1 = p["owned"]
        cover[o0:o1] += 1
        assert sorted(p["tiles"].tolist()) == parts[r].tolist()
        nb = len(p["boundary"])
        assert p["tiles"][:nb].tolist() == p["boundary"].tolist() and nb <= len(parts[r])
        for d, r0, r1, c0, c1 in p["sends"]:
            assert abs(d - r) == 1 and 0 < r1 - r0 <= 512 and (r, r0, r1, c0, c1) in plan[d]["recvs"]
            q0, q1 = plan[d]["owned"]
            assert q0 <= r0 and r1 <= q1                           # the receiver owns those rows
            mine = crops[p["boundary"]]
            hit = mine[(mine[:, 1] < r1) & (mine[:, 1] + 512 > r0)]
            assert c0 == hit[:, 0].min() and c1 == hit[:, 0].max() + 512      # tight: the columns of the sender's tiles on those rows
            other = crops[np.setdiff1d(parts[r], p["boundary"])]   # no tile outside `boundary` touches a sent rectangle
            assert not ((other[:, 1] < r1) & (other[:, 1] + 512 > r0)).any()
        # every pixel of the rank's band that another rank owns is in exactly one outgoing rectangle
        a, b = p["band"]
        foreign = np.zeros((b - a, 5120 // 256), dtype=int)        # (256-pixel column cells: tile origins are multiples of 256)
        for x, y in crops[parts[r], :2]:
            foreign[y - a:y - a + 512, x // 256:(x + 512) // 256] = 1
        foreign[max(o0, a) - a:min(o1, b) - a] = 0
        sent = np.zeros_like(foreign)
        for _d, r0, r1, c0, c1 in p["sends"]:
            sent[r0 - a:r1 - a, c0 // 256:c1 // 256] += 1
        assert sent.max() <= 1 and (sent >= foreign).all()
        assert len(p["recvs"]) <= 4                                # (the one-launch finish of a rank takes up to four rectangles)
    assert (cover == 1).all()
    # halo volume per rank and direction (round 6): the cut balances the two directions of every link -- no direction carries
    # more than 11.8 MB at C = 4 (it was 18.9 MB one way and 3.1 MB the other on the worst link with the cut half a tile below the row top)
    one_way = {}
    for r, p in enumerate(plan):
        for d, r0, r1, c0, c1 in p["sends"]:
            one_way[(r, d)] = one_way.get((r, d), 0) + 4 * 4 * (r1 - r0) * (c1 - c0)
    assert max(one_way.values()) <= 11.8e6 and len(one_way) == 14, max(one_way.values())
    # more ranks than tiles: the surplus ranks own nothing and exchange nothing
    small = TO.slicer_geometry((100, 100), 64, 32)["crops"]
    plan = band_plan(small, 16, 128)
    live = [p for p in plan if p["owned"] is not None]
    assert len(live) == len(small) and live[0]["owned"][0] == 0 and live[-1]["owned"][1] == 128
    assert all(p["sends"] == [] and p["recvs"] == [] and len(p["tiles"]) == 0 for p in plan if p["owned"] is None)


# ------------------------------------------------------------------ batch-sharded region losses
def _dice_from_stats(inter, pred_mass, true_mass):
    score = 2.0 * inter / (pred_mass + true_mass).clamp_min(1e-7)
    return ((1.0 - score) * (true_mass > 0)).mean()


def _stats_worker(rank, world, port, q):
    from pytorch_toolbelt_amd.parallel import sync_region_statistics

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(5)
        probs = torch.rand((4, 3, 50), generator=g)          # the whole batch; this rank owns images [2r, 2r+2)
        onehot = (torch.rand((4, 3, 50), generator=g) < 0.4).float()
        p = probs[2 * rank:2 * rank + 2].clone().requires_grad_(True)
        t = onehot[2 * rank:2 * rank + 2]
        local = ((p * t).sum((0, 2)), p.sum((0, 2)), t.sum((0, 2)))
        with sync_region_statistics():
            inter, pm, tm = sync_region_statistics.apply(local)
        loss = _dice_from_stats(inter, pm, tm)
        loss.backward()
        assert sync_region_statistics._active is None
        assert sync_region_statistics.apply(local) is local            # outside the context: untouched
        q.put((rank, float(loss), p.grad.numpy()))
    finally:
        dist.destroy_process_group()


def test_region_statistics_all_reduce_matches_single_process():
    """World size 2 over gloo: the synchronised Dice value equals the single-process value on the whole batch, and each
    rank's input gradient equals WORLD x its slice of the single-process gradient (every rank back-propagates the same
    global loss, and the backward all-reduce sums those identical upstream gradients)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_stats_worker, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    got = sorted([q.get(timeout=120) for _ in range(world)])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    g = torch.Generator().manual_seed(5)
    probs = torch.rand((4, 3, 50), generator=g).requires_grad_(True)
    onehot = (torch.rand((4, 3, 50), generator=g) < 0.4).float()
    ref = _dice_from_stats((probs * onehot).sum((0, 2)), probs.sum((0, 2)), onehot.sum((0, 2)))
    ref.backward()
    for rank, loss, grad in got:
        assert abs(loss - float(ref)) < 1e-6
        assert np.allclose(grad, world * probs.grad[2 * rank:2 * rank + 2].numpy(), atol=1e-7)


# ------------------------------------------------------------------ plan fuzzing without processes
class _FakeDist:
    """Rank / world stand-in: the ranks are played one after the other and their rectangles are handed over by hand."""

    def __init__(self, rank, world):
        self.rank, self.world = rank, world

    def get_rank(self, group=None):
        return self.rank

    def get_world_size(self, group=None):
        return self.world


def _simulate(shape, tile, step, world, partition, C=1, seed=0):
    geom = TO.slicer_geometry(shape, tile, step)
    w = TO.pyramid_window(*((tile, tile) if isinstance(tile, int) else tile))[0]
    crops = geom["crops"]
    rng = np.random.default_rng(seed)
    th, tw = int(crops[0, 3]), int(crops[0, 2])
    outs = rng.standard_normal((len(crops), C, th, tw)).astype(np.float32)
    ranks = []
    for r in range(world):
        m = ShardedTileMerger(geom["target_shape"], C, w, crops, device="cpu", ops=OracleOps, dist=_FakeDist(r, world), partition=partition)
        m._start_exchange = lambda: None
        if m.local is not None:
            m.reset()
            for b0 in range(0, len(m.tiles), 5):
                idx = m.tiles[b0:b0 + 5]
                m.integrate_batch(torch.from_numpy(outs[idx]), crops[idx])
                # the exchange must not be armed before every boundary tile is in, and must be armed right after
                done = set(m.tiles[:b0 + 5].tolist())
                assert (len(m._remaining) == 0) == set(m.plan[r]["boundary"].tolist()).issubset(done)
        ranks.append(m)
    H, W = geom["target_shape"]
    full = torch.full((C, H, W), float("nan"))
    owned = np.zeros((H, W), dtype=int)
    for m in ranks:
        if m.local is None:
            continue
        for buf, (src, r0, r1, c0, c1) in zip(m._recv_buf, m.recvs):
            buf.copy_(ranks[src]._rect(r0, r1, c0, c1))
        m._exchanged = True
        band = m.merge()
        o0, o1 = m.owned_rows
        owned[o0:o1] += 1
        if band is not None:
            assert tuple(band.shape) == (C, o1 - o0, W)
            full[:, o0:o1] = band
    assert (owned == 1).all()
    st = TO.merger_new(geom["target_shape"], C, w)
    TO.merger_integrate(st, outs, crops)
    np.testing.assert_allclose(full.numpy(), TO.merger_merge(st), rtol=0, atol=1e-5)
    return ranks


def test_plan_fuzz_in_process():
    rng = np.random.default_rng(2024)
    for case in range(60):
        th, tw = int(rng.choice([16, 24, 32, 48])), int(rng.choice([16, 32, 40]))
        sh, sw = int(rng.integers(max(1, th // 4), th + 1)), int(rng.integers(max(1, tw // 4), tw + 1))
        shape = (int(rng.integers(th, 6 * th)), int(rng.integers(tw, 6 * tw)))
        world = int(rng.integers(1, 9))
        partition = {0: "rows", 1: "tiles", 2: "pixel_rows"}[case % 3]
        try:
            _simulate(shape, (th, tw), (sh, sw), world, partition, C=int(rng.integers(1, 3)), seed=case)
        except Exception as e:
            raise AssertionError(f"case {case}: shape={shape} tile={(th, tw)} step={(sh, sw)} world={world} partition={partition}: {e}") from e


def test_headline_geometry_eight_ranks_in_process():
    """BASELINE cfg3 geometry (5000 x 5000, 512 / 256, 8 ranks), one channel: every rank's band, the bidirectional halo
    rectangles and the owned-row merge reproduce the single-device result."""
    ranks = _simulate((5000, 5000), 512, 256, 8, "tiles", C=1, seed=3)
    assert [len(m.tiles) for m in ranks] == [45, 45, 45, 45, 45, 45, 45, 46]
    assert max(m.bottom - m.top for m in ranks) <= 1280       # band accumulators: at most 4 tile rows + ownership slack


@pytest.mark.parametrize("world,partition", [(2, "tiles"), (4, "tiles"), (8, "tiles"), (8, "rows"), (3, "rows")])
def test_deferred_geometry_of_every_rank(world, partition):
    """What the deferred band plan of a rank is built from (headline geometry): the rows a rank finishes alone are touched by
    none of the other ranks' tiles and lie inside the rows it owns; the cuts contain the ends of the owned rows, of the final rows
    and of every exchanged rectangle, all on the 4-pixel grid; together the ranks' owned rows tile the image."""
    from pytorch_toolbelt_amd.parallel import deferred_geometry

    geom = TO.slicer_geometry((5000, 5000), 512, 256)
    crops = geom["crops"]
    plan = band_plan(crops, world, 5120, partition)
    covered = np.zeros(5120, dtype=int)
    final_rows = 0
    for r in range(world):
        (f0, f1), cuts = deferred_geometry(plan, r, crops, 5120)
        o0, o1 = plan[r]["owned"]
        covered[o0:o1] += 1
        assert all(c % 4 == 0 for c in cuts) and {o0, o1} <= set(cuts)
        for _peer, r0, r1, _c0, _c1 in plan[r]["sends"] + plan[r]["recvs"]:
            assert {r0, r1} <= set(cuts)
        if f1 > f0:
            assert o0 <= f0 < f1 <= o1 and {f0, f1} <= set(cuts)
            for q in range(world):
                if q != r:
                    ys = crops[plan[q]["tiles"], 1]
                    assert not ((ys < f1) & (ys + 512 > f0)).any(), "a row finished alone is touched by another rank's tile"
            for _src, r0, r1, _c0, _c1 in plan[r]["recvs"]:
                assert r1 <= f0 or r0 >= f1
            final_rows += f1 - f0
    assert (covered == 1).all()
    assert final_rows >= (5120 - world * 1024)     # at most ~2 tile heights per rank are shared with neighbours


@pytest.mark.parametrize("world", [2, 4, 8])
def test_rank_band_plan_launches_by_class(world):
    """ptb_band_plan_create2 (host-side planning, runs without a GPU): with the outgoing rectangles' rows given as `early` ranges a
    rank's plan has ONE launch group per run of such rows (towards the upper / the lower neighbour) -- complete as soon as the tiles
    feeding it are in, which the rank issues first -- and the other rows in groups that do not break at the cuts (<= 3 launches per rank
    at N = 8 instead of 4-6);
    ptb_band_plan_rows_launched follows the exact rows of each group."""
    import ctypes

    from pytorch_toolbelt_amd import _native as N
    from pytorch_toolbelt_amd.parallel import deferred_geometry

    lib = N.load()
    geom = TO.slicer_geometry((5000, 5000), 512, 256)
    crops = geom["crops"]
    plan = band_plan(crops, world, 5120, "tiles")
    for r in range(world):
        me = plan[r]
        a, b = me["band"]
        o0, o1 = me["owned"]
        top, bottom = min(a, o0), max(b, o1)
        final, cuts = deferred_geometry(plan, r, crops, 5120)
        local = np.ascontiguousarray(crops[me["tiles"], :2].T.astype(np.int64))
        local[1] -= top
        cut_arr = np.ascontiguousarray(np.array([c - top for c in cuts if top < c < bottom], dtype=np.int64))
        spans = early_spans(me, top)
        assert all(b0 > a1 for (_a0, a1), (b0, _b1) in zip(spans, spans[1:])) and len(spans) <= 2      # (towards the upper / the lower neighbour)
        early = np.ascontiguousarray(np.array(spans, dtype=np.int64).reshape(-1))
        counts = {}
        for two_phase in (False, True):
            handle = ctypes.c_void_p()
            e = early if two_phase else np.zeros(0, dtype=np.int64)
            nbytes = lib.ptb_band_plan_create2(local[0].ctypes.data_as(N._i64p), local[1].ctypes.data_as(N._i64p), local.shape[1], 4, 512, 512,
                                               bottom - top, 5120, 1024, final[0] - top, final[1] - top, cut_arr.ctypes.data_as(N._i64p), len(cut_arr),
                                               e.ctypes.data_as(N._i64p) if len(e) else None, len(e) // 2, ctypes.byref(handle))
            assert nbytes > 0
            ng = ctypes.c_int()
            lib.ptb_band_plan_info(handle, ctypes.byref(ng), None, None, None, None)
            rows = np.zeros(3 * ng.value, dtype=np.int64)
            last_group = np.zeros(local.shape[1], dtype=np.int64)
            lib.ptb_band_plan_info(handle, None, None, None, last_group.ctypes.data_as(N._i64p), rows.ctypes.data_as(N._i64p))
            counts[two_phase] = ng.value
            if two_phase and spans:
                n_boundary = len(me["boundary"])
                # one early group per outgoing row run (round 6), first in the plan; each complete once the boundary tiles (issued first) are in
                assert all(rows[3 * g + 2] < n_boundary for g in range(len(spans))), "an early group waits for a tile that feeds no outgoing rectangle"
                assert sorted((int(rows[3 * g]), int(rows[3 * g + 1])) for g in range(len(spans))) == spans
                assert all(lib.ptb_band_plan_rows_launched(handle, s0, s1) == 0 for s0, s1 in spans)      # nothing launched yet
            lib.ptb_band_plan_destroy(handle)
        assert counts[True] <= counts[False]
        if world == 8:
            assert counts[True] <= 3, counts


# ------------------------------------------------------------------ pipelined exchange (merge_async) and the communication-free partition
def _pipeline_worker(rank, world, port, shape, tile, step, C, partition, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        geom = TO.slicer_geometry(shape, tile, step)
        w = TO.pyramid_window(*tile)[0]
        crops = geom["crops"]
        rng = np.random.default_rng(23)
        outs = rng.standard_normal((len(crops), C, *tile)).astype(np.float32)
        m = ShardedTileMerger(geom["target_shape"], C, w, crops, device="cpu", ops=OracleOps, partition=partition)
        mine = m.tiles

        def feed(image_no):
            for b0 in range(0, len(mine), 3):
                idx = mine[b0:b0 + 3]
                m.integrate_batch(torch.from_numpy(outs[idx] * (image_no + 1)), crops[idx])

        n_images = 5
        tickets, bands = [], []
        for image_no in range(n_images):          # image i's exchange is only joined after image i + 1 was fed
            feed(image_no)
            tickets.append(m.merge_async())
            if image_no >= 1:
                bands.append(tickets[image_no - 1].result())
        bands.append(tickets[-1].result())
        assert all(t.done for t in tickets) and tickets[0].result() is bands[0]
        assert m.images_async == (n_images if m.local is not None else 0)
        if m.local is not None:
            assert len(m._slots) == 2, "a pipelined merger alternates between two sets of buffers"
        # never calling result() in time is fine too: the slot's image is completed when its buffers are needed again
        lazy_tickets = []
        for image_no in range(3):
            feed(image_no)
            lazy_tickets.append(m.merge_async())
        if m.local is not None:
            assert lazy_tickets[0].done and not lazy_tickets[2].done
        late = [t.result() for t in lazy_tickets]
        # the synchronous form of the same images: the same bits
        for image_no in range(n_images):
            m.reset()
            feed(image_no)
            band = m.merge()
            assert (band is None) == (bands[image_no] is None)
            if band is not None:
                assert torch.equal(band, bands[image_no]), f"pipelined image {image_no} differs from the synchronous merge"
                if image_no < 3:
                    assert torch.equal(band, late[image_no])
        full = m.gather(bands[1])
        if rank == 0:
            q.put(full.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,shape,tile,step,C,partition", [
    (2, (300, 200), (64, 64), (32, 32), 2, "tiles"),
    (3, (300, 200), (64, 64), (32, 32), 1, "rows"),
    (5, (300, 200), (64, 64), (32, 32), 1, "tiles"),
    (8, (300, 200), (64, 64), (32, 32), 1, "tiles"),
    (8, (40, 40), (64, 64), (32, 32), 1, "tiles"),        # one tile, eight ranks: seven of them idle through the whole pipeline
    (3, (300, 200), (64, 64), (32, 32), 2, "pixel_rows"),
])
def test_pipelined_merge_equals_synchronous(world, shape, tile, step, C, partition):
    """merge_async(): image i's halo exchange stays in flight while image i + 1 is integrated (second set of buffers) and is joined
    by PendingBand.result() -- bit-identical to the synchronous merge() of the same image, and the gathered map equals the
    single-process merge."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, world, port, shape, tile, step, C, partition, q)) for r in range(world)]
    for p in procs:
        p.start()
    full = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    geom = TO.slicer_geometry(shape, tile, step)
    w = TO.pyramid_window(*tile)[0]
    rng = np.random.default_rng(23)
    outs = rng.standard_normal((len(geom["crops"]), C, *tile)).astype(np.float32) * 2
    st = TO.merger_new(geom["target_shape"], C, w)
    TO.merger_integrate(st, outs, geom["crops"])
    want = TO.merger_merge(st)
    if partition == "pixel_rows":
        assert np.array_equal(full, want, equal_nan=True), "the communication-free partition must reproduce the single-device bits"
    else:
        np.testing.assert_allclose(full, want, rtol=0, atol=1e-5)


@pytest.mark.parametrize("world", [1, 2, 3, 5, 8, 16])
def test_pixel_row_partition_is_bit_identical_to_one_device(world):
    """partition="pixel_rows": every rank owns an equal share of the pixel rows and holds every tile touching them; no rectangle
    travels, and because every covering tile of an owned pixel is local and summed in the global order, the assembled map is the
    single-device merge bit for bit (ragged geometry, uncovered rows included)."""
    from pytorch_toolbelt_amd.parallel import pixel_row_cuts, pixel_row_partition

    rng = np.random.default_rng(world)
    for shape, tile, step in [((300, 200), (64, 64), (32, 32)), ((130, 90), (48, 32), (20, 32)), ((64, 64), (64, 64), (64, 64))]:
        geom = TO.slicer_geometry(shape, tile, step)
        crops = geom["crops"]
        H, W = geom["target_shape"]
        w = TO.pyramid_window(*tile)[0]
        outs = rng.standard_normal((len(crops), 2, *tile)).astype(np.float32)
        cuts = pixel_row_cuts(H, world)
        assert cuts[0] == 0 and cuts[-1] == H and all(b >= a for a, b in zip(cuts, cuts[1:]))
        parts = pixel_row_partition(crops, world, H)
        full = np.full((2, H, W), np.nan, dtype=np.float32)
        owned = np.zeros(H, dtype=int)
        for r in range(world):
            m = ShardedTileMerger((H, W), 2, w, crops, device="cpu", ops=OracleOps, dist=_FakeDist(r, world), partition="pixel_rows")
            assert m.sends == [] and m.recvs == [] and m.tiles.tolist() == parts[r].tolist()
            if m.local is None:
                continue
            for b0 in range(0, len(m.tiles), 4):
                idx = m.tiles[b0:b0 + 4]
                m.integrate_batch(torch.from_numpy(outs[idx]), crops[idx])
            band = m.merge_async().result()
            o0, o1 = m.owned_rows
            owned[o0:o1] += 1
            full[:, o0:o1] = band.numpy()
        assert (owned == 1).all()
        st = TO.merger_new((H, W), 2, w)
        TO.merger_integrate(st, outs, crops)
        assert np.array_equal(full, TO.merger_merge(st), equal_nan=True)
    # headline geometry: balanced rows on the 64-row grid, 57 .. 76 tiles per rank at N = 8 (against 45 / 46 when tiles are not shared)
    geom = TO.slicer_geometry((5000, 5000), 512, 256)
    if world == 8:
        parts = pixel_row_partition(geom["crops"], 8, 5120)
        assert pixel_row_cuts(5120, 8) == [640 * i for i in range(9)]
        assert [len(p) for p in parts] == [57, 76, 76, 76, 76, 76, 76, 57]


def test_rank_band_plan_clips_tiles_to_the_owned_pixel_rows():
    """ptb_band_plan_create3(PTB_PLAN_CLIP_ROWS) (host-side planning, runs without a GPU): the plan of a pixel-row rank covers exactly
    its owned rows, accepts tiles that hang over both ends, and is refused without the flag."""
    import ctypes

    from pytorch_toolbelt_amd import _native as N
    from pytorch_toolbelt_amd.parallel import band_plan

    lib = N.load()
    geom = TO.slicer_geometry((5000, 5000), 512, 256)
    crops = geom["crops"]
    plan = band_plan(crops, 8, 5120, "pixel_rows")
    for r in (0, 1, 7):
        o0, o1 = plan[r]["owned"]
        local = np.ascontiguousarray(crops[plan[r]["tiles"], :2].T.astype(np.int64))
        local[1] -= o0
        assert (local[1] < 0).any() or r == 0
        for flags, ok in ((0, False), (1, True)):
            handle = ctypes.c_void_p()
            nbytes = lib.ptb_band_plan_create3(local[0].ctypes.data_as(N._i64p), local[1].ctypes.data_as(N._i64p), local.shape[1], 4, 512, 512,
                                               o1 - o0, 5120, 1024, 0, o1 - o0, None, 0, None, 0, flags, ctypes.byref(handle))
            if not ok:
                assert nbytes == -4, nbytes            # PTB_EBOUNDS: a tile outside the plan's rows
                continue
            assert nbytes > 0
            ng, nb, ni = ctypes.c_int(), ctypes.c_int(), ctypes.c_int64()
            lib.ptb_band_plan_info(handle, ctypes.byref(ng), ctypes.byref(nb), ctypes.byref(ni), None, None)
            rows = np.zeros(3 * ng.value, dtype=np.int64)
            lib.ptb_band_plan_info(handle, None, None, None, None, rows.ctypes.data_as(N._i64p))
            assert rows[0] == 0 and rows[3 * ng.value - 2] == o1 - o0
            # 640 owned rows x 5120 columns in 64 x 64 (or 64 x 32) items: every owned pixel exactly once
            assert ni.value in (640 * 5120 // (64 * 64), 640 * 5120 // (64 * 32))
            lib.ptb_band_plan_destroy(handle)
    assert lib.ptb_band_plan_create3(None, None, 1, 1, 4, 4, 4, 4, 4, 0, 4, None, 0, None, 0, 2, None) == -1


def test_host_staged_work_delivers_after_the_transfers():
    """``_HostStagedWork`` (device rectangles under a non-RCCL backend travel through host tensors): ``wait()`` first joins every
    transfer, then copies what was received into the device-side buffers -- never the other way round."""
    from pytorch_toolbelt_amd.parallel import _HostStagedWork

    order = []
    host = [torch.zeros(2, 3), torch.zeros(4)]
    dev = [torch.full((2, 3), -1.0), torch.full((4,), -1.0)]

    class _W:
        def __init__(self, k):
            self.k = k

        def wait(self):
            order.append(self.k)
            host[self.k].fill_(float(self.k + 1))      # the data arrives with the transfer's completion

    w = _HostStagedWork([_W(0), _W(1)], host, dev, [torch.ones(1)])
    assert float(dev[0].max()) == -1.0
    w.wait()
    assert order == [0, 1] and float(dev[0].min()) == 1.0 and float(dev[1].min()) == 2.0
    assert w.works == [] and w.host_send is None
    w.wait()                                           # idempotent: nothing left to join, the same data again
    assert float(dev[1].max()) == 2.0
