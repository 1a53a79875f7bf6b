"""CPU: the host-side launch planner of the accumulate kernels (cell decomposition + first-touch bitmap), exercised
through the ptb_debug_plan test hook with random tile layouts.  Invariants: within a launch group cells are disjoint,
their union is the union of the group's tiles, every cell lists exactly the tiles covering it in ascending batch order;
groups cover the batch in order; freshness flags agree with a pixel-level simulation."""
import ctypes

import numpy as np
import pytest
from hypothesis import given, settings
from hypothesis import strategies as st

from pytorch_toolbelt_amd import _native as N


def plan(xs, ys, th, tw, H, W, chunk_rows=32, fresh=None, fresh_rows=32):
    lib = N.load()
    cap = 4096
    out = (ctypes.c_int * (12 * cap))()
    rc = lib.ptb_debug_plan(N.i64_array(list(xs)), N.i64_array(list(ys)), len(xs), th, tw, H, W, chunk_rows,
                            fresh.ctypes.data if fresh is not None else None, fresh_rows, out, cap)
    if rc < 0:
        return rc, None
    arr = np.frombuffer(out, dtype=np.int32)[:12 * rc].reshape(rc, 12).copy()
    return rc, arr


def check_plan(xs, ys, th, tw, H, W, cells):
    B = len(xs)
    seen_tiles = []
    for g in np.unique(cells[:, 0]):
        grp = cells[cells[:, 0] == g]
        tiles = sorted({t for row in grp for t in row[8:8 + row[7]]})
        seen_tiles.append(tiles)
        cover = np.zeros((H, W), dtype=np.int32)   # how many cells of this group claim each pixel
        for t in tiles:
            assert 0 <= t < B
        tile_mask = np.zeros((H, W), dtype=bool)
        for t in tiles:
            tile_mask[ys[t]:ys[t] + th, xs[t]:xs[t] + tw] = True
        for row in grp:
            _, ox, oy, w, h, _fresh, _ce, nt = row[:8]
            assert w > 0 and h > 0 and 1 <= nt <= 4
            cover[oy:oy + h, ox:ox + w] += 1
            lst = list(row[8:8 + nt])
            assert lst == sorted(lst)
            # the cell's cover list == tiles of the group that contain the cell (checked at its corners: cells never
            # straddle a tile edge, so all pixels of a cell share one cover set)
            for (py, px) in ((oy, ox), (oy + h - 1, ox + w - 1)):
                want = [t for t in tiles if xs[t] <= px < xs[t] + tw and ys[t] <= py < ys[t] + th]
                assert lst == want
        assert cover.max() <= 1, "cells of one launch overlap"
        assert np.array_equal(cover.astype(bool), tile_mask), "cells do not tile the union of the group's tiles"
    flat = [t for g in seen_tiles for t in g]
    assert flat == list(range(B)), "launch groups must partition the batch in order"


@settings(max_examples=60, deadline=None)
@given(st.data())
def test_random_layouts(data):
    th = data.draw(st.sampled_from([8, 12, 16, 24, 31]))
    tw = data.draw(st.sampled_from([8, 16, 20, 28, 33]))
    H = data.draw(st.integers(th, 96))
    W = data.draw(st.integers(tw, 96))
    B = data.draw(st.integers(1, 24))
    xs = [data.draw(st.integers(0, W - tw)) for _ in range(B)]
    ys = [data.draw(st.integers(0, H - th)) for _ in range(B)]
    n, cells = plan(xs, ys, th, tw, H, W)
    assert n > 0
    check_plan(xs, ys, th, tw, H, W, cells)


def test_slicer_row_of_eight_and_wraparound():
    from pytorch_toolbelt_amd.inference.tiles import ImageSlicer

    s = ImageSlicer((5000, 5000, 3), 512, 256, weight="mean")
    xs, ys = s.crops[:8, 0], s.crops[:8, 1]
    n, cells = plan(xs, ys, 512, 512, 5120, 5120)
    assert n == 9 and sorted(cells[:, 7].tolist()) == [1, 1] + [2] * 7          # 9 cells: 7 two-tile, 2 one-tile
    assert (cells[:-1, 7] >= cells[1:, 7]).all()                                 # heavy cells first
    assert cells[-1, 6] == 9 * (256 // 64) * (512 // 32)                         # chunk prefix = total 64x32 chunks
    xs, ys = s.crops[16:24, 0], s.crops[16:24, 1]                                # batch wrapping from tile row 0 to row 1
    n, cells = plan(xs, ys, 512, 512, 5120, 5120)
    check_plan(list(xs), list(ys), 512, 512, 5120, 5120, cells)
    # 16-fold cover (step = size / 4) cannot go into one launch: the planner splits the batch
    d = ImageSlicer((64, 64), 32, 8, weight="mean")
    n, cells = plan(d.crops[:, 0], d.crops[:, 1], 32, 32, *d.target_shape)
    assert len(np.unique(cells[:, 0])) > 1
    check_plan(list(d.crops[:, 0]), list(d.crops[:, 1]), 32, 32, *d.target_shape, cells)


def test_first_touch_bitmap():
    from pytorch_toolbelt_amd.inference.tiles import ImageSlicer

    s = ImageSlicer((1000, 1500, 3), 512, 256, weight="mean")
    H, W = s.target_shape
    fresh = np.ones(((H + 31) // 32, (W + 63) // 64), dtype=np.uint8)
    written = np.zeros((H, W), dtype=bool)
    row_len = len(np.unique(s.crops[:, 0]))
    for b0 in range(0, len(s.crops) - 1, row_len):          # one tile row per batch (the last single tile is done below)
        idx = range(b0, min(b0 + row_len, len(s.crops)))
        xs, ys = s.crops[idx, 0], s.crops[idx, 1]
        n, cells = plan(xs, ys, 512, 512, H, W, 32, fresh, 32)
        assert n > 0
        for row in cells:
            _, ox, oy, w, h, fr = row[:6]
            region = written[oy:oy + h, ox:ox + w]
            assert region.all() or not region.any(), "a cell must be uniformly fresh or uniformly written"
            assert fr == (0 if region.any() else 1)
            written[oy:oy + h, ox:ox + w] = True
    assert not fresh.any() and written.all()
    # a tile whose footprint is partly written is cut into rectangles of uniform freshness (no zero-fill needed)
    fresh = np.ones_like(fresh)
    fresh[:8, :4] = 0                                   # upper-left quarter of the first tile already written
    before = fresh.copy()
    n, cells = plan([0], [0], 512, 512, H, W, 32, fresh, 32)
    assert n == 3
    area = 0
    for row in cells:
        _, ox, oy, w, h, fr = row[:6]
        blocks = before[oy // 32:(oy + h + 31) // 32, ox // 64:(ox + w + 63) // 64]
        assert blocks.all() if fr else not blocks.any()
        area += w * h
    assert area == 512 * 512 and not fresh[:16, :8].any()
    # unaligned tiles never use first-touch stores
    fresh = np.ones_like(fresh)
    rc, _ = plan([4], [0], 512, 512, H, W, 32, fresh, 32)
    assert rc == N.EFRESH
    rc, _ = plan([0], [0], 512, 512, H, W, 64, fresh, 32)   # bitmap granularity != chunk rows
    assert rc == N.EFRESH
    rc, _ = plan([W], [0], 512, 512, H, W)
    assert rc == -4


def _c_band_plan(crops, C, th, tw, H, W, rows, final=None, cuts=()):
    """ptb_band_plan_create + ptb_band_plan_info (host-side planning only: no GPU needed)."""
    import ctypes

    lib = N.load()
    xy = np.ascontiguousarray(np.asarray(crops, dtype=np.int64)[:, :2].T)
    handle = ctypes.c_void_p()
    cut_arr = np.asarray(cuts, dtype=np.int64)
    nbytes = lib.ptb_band_plan_create(xy[0].ctypes.data_as(N._i64p), xy[1].ctypes.data_as(N._i64p), xy.shape[1], C, th, tw, H, W, rows,
                                      final[0] if final else 0, final[1] if final else H, cut_arr.ctypes.data_as(N._i64p) if len(cut_arr) else None,
                                      len(cut_arr), ctypes.byref(handle))
    if nbytes < 0:
        return int(nbytes), None
    ng, nb, ni = ctypes.c_int(), ctypes.c_int(), ctypes.c_int64()
    assert lib.ptb_band_plan_info(handle, ctypes.byref(ng), ctypes.byref(nb), ctypes.byref(ni), None, None) == 0
    last_group = np.zeros(xy.shape[1], dtype=np.int64)
    rows_arr = np.zeros(3 * ng.value, dtype=np.int64)
    assert lib.ptb_band_plan_info(handle, None, None, None, last_group.ctypes.data_as(N._i64p), rows_arr.ctypes.data_as(N._i64p)) == 0
    lib.ptb_band_plan_destroy(handle)
    assert nbytes == ni.value * 64
    return int(nbytes), dict(groups=rows_arr.reshape(-1, 3), bands=nb.value, items=ni.value, last_group=last_group)


@pytest.mark.parametrize("item_rows", [64, 32])
def test_band_plan_of_the_deferred_merger(item_rows):
    """ptb_band_plan_create: bands = rows between consecutive tile edges, grouped into launches of ~rows_per_launch rows; the
    tile that completes each group, the last group reading each tile, the work-item count (64 columns x `item_rows` rows each:
    ptb_set_tunable key 11, default 64) -- for the headline geometry and some irregular ones."""
    from oracle import tiles_oracle as TO

    assert N.load().ptb_set_tunable(11, item_rows) == 0
    try:
        _band_plan_checks(TO, item_rows)
    finally:
        assert N.load().ptb_set_tunable(11, 64) == 0


def _band_plan_checks(TO, item_rows):

    crops = TO.slicer_geometry((5000, 5000), 512, 256)["crops"]
    # one band per launch: the 20 bands of the headline geometry
    _, p = _c_band_plan(crops, 4, 512, 512, 5120, 5120, 256)
    assert p["bands"] == 20 and len(p["groups"]) == 20
    assert [(int(g[0]), int(g[1])) for g in p["groups"]] == [(256 * k, 256 * k + 256) for k in range(20)]
    assert [int(g[2]) for g in p["groups"]] == [19 * min(k, 18) + 18 for k in range(20)]   # the last tile of tile row k (19 and 18: row 18)
    assert p["items"] == 20 * (5120 // 64) * (256 // item_rows)
    assert p["last_group"][0] == 1 and p["last_group"][19] == 2 and p["last_group"][360] == 19
    # 1024 rows per launch (the default): 5 launches of 4 bands, every pixel row in exactly one group
    _, p = _c_band_plan(crops, 4, 512, 512, 5120, 5120, 1024)
    assert [(int(g[0]), int(g[1])) for g in p["groups"]] == [(1024 * k, 1024 * k + 1024) for k in range(5)]
    assert [int(g[2]) for g in p["groups"]] == [19 * 3 + 18, 19 * 7 + 18, 19 * 11 + 18, 19 * 15 + 18, 360]   # tile row r covers rows 256 r .. 256 r + 512
    assert p["last_group"][0] == 0 and p["last_group"][19 * 3] == 1 and p["last_group"][360] == 4
    assert p["items"] == (5120 // 64) * (5120 // item_rows)
    # a launch never takes more than 224 tiles: one launch for the whole image is cut into several
    _, p = _c_band_plan(crops, 4, 512, 512, 5120, 5120, 1 << 20)
    assert len(p["groups"]) == 2 and int(p["groups"][0][0]) == 0 and int(p["groups"][-1][1]) == 5120
    # uncovered rows / columns are planned too (they merge to NaN like the reference's 0 / 0)
    _, p = _c_band_plan(np.array([[64, 64, 128, 128], [256, 64, 128, 128]]), 1, 128, 128, 320, 512, 64)
    assert int(p["groups"][0][0]) == 0 and int(p["groups"][-1][1]) == 320
    assert p["items"] == (512 // 64) * (320 // item_rows)      # bands of 64, 128, 128 rows
    # step < tile / 4: more than 4 tiles over a pixel -> not deferrable
    dense = TO.slicer_geometry((600, 600), 256, 48)["crops"]
    assert _c_band_plan(dense, 1, 256, 256, 640, 640, 256)[0] == -2
    # tile origins off the 4-pixel grid -> not deferrable
    odd = TO.slicer_geometry((300, 300), 130, 65)["crops"]
    assert _c_band_plan(odd, 1, 130, 130, int(odd[:, 1].max()) + 130, int(odd[:, 0].max()) + 130, 256)[0] == -2
    # a tile outside the map
    assert _c_band_plan(np.array([[0, 0, 64, 64], [480, 0, 64, 64]]), 1, 64, 64, 64, 512, 64)[0] == -4
    # caller-given cuts are band edges and launch boundaries (multi-GPU ownership rows)
    _, p = _c_band_plan(crops, 4, 512, 512, 5120, 5120, 1024, final=(128, 896), cuts=[128, 896])
    assert [(int(g[0]), int(g[1])) for g in p["groups"]][:4] == [(0, 128), (128, 896), (896, 1792), (1792, 2816)]   # groups restart at a cut
    assert p["bands"] == 22


def test_custody_of_a_deferred_merger_is_what_the_byte_budget_assumes():
    """``Bands.peak_tiles()`` (inference/_merge_modes.py) -- the most tiles a deferred merger holds at any launch, what self-planned
    deferral sizes its launch groups by (PTB_DEFER_BYTES) -- against a simulation of the custody rules on the C plan: a batch stays
    held until the last launch group that reads one of its tiles has gone out."""
    from oracle import tiles_oracle as TO
    from pytorch_toolbelt_amd.inference._merge_modes import Bands

    for shape, tile, step, rows, want in (((5000, 5000), 512, 256, 1024, 95), ((5000, 5000), 512, 256, 512, 57), ((5000, 5000), 512, 256, 256, 38),
                                          ((900, 420), 128, 64, 128, 18), ((900, 420), 128, 64, 1024, 84)):
        geom = TO.slicer_geometry(shape, tile, step)
        crops = geom["crops"]
        H, W = geom["target_shape"]
        _, p = _c_band_plan(crops, 2, tile, tile, H, W, rows)
        groups = [tuple(int(v) for v in g) for g in p["groups"]]
        lasts = [g[2] for g in groups]
        b = Bands(None, None, groups, p["bands"], p["last_group"], all(x <= y for x, y in zip(lasts, lasts[1:])))
        assert b.monotone and b.peak_tiles() == want, (shape, rows, b.peak_tiles())
        # simulate: tiles arrive one by one; after tile t every group whose last tile is t launches; a tile leaves custody when the last
        # group reading it has launched
        held, peak, done = [], 0, 0
        for t in range(len(crops)):
            held.append(t)
            peak = max(peak, len(held))
            while done < len(groups) and groups[done][2] <= t:
                done += 1
            held = [h for h in held if p["last_group"][h] >= done]
        assert peak == want and not held
    # the headline geometry under the default budget: 95 tiles + one batch of 8, 33.5 MB each = 3.46 GB < 4 GiB
    assert (95 + 8) * 8 * 4 * 512 * 512 * 4 < 4 << 30
