"""GPU: the N > 1 path end to end with the REAL device code -- `world` processes share the one GPU of the test box and
talk over gloo (RCCL refuses two ranks on one device); each runs ShardedTileMerger with the HIP kernels, exchanges its
halo rectangles point to point and merges its band.  The gathered result must equal the numpy ORACLE's single-device merge
(the reference's sequential loop, inference/tiles.py:321-346) within 1e-5 -- a pixel on a rank boundary sums its own rank's
tiles first, then the neighbour's partial sum, see parallel.py -- and, as a second check, the single-process HIP TileMerger."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, partition, defer, q):
    import torch.distributed as dist

    from oracle import tiles_oracle as TO
    from pytorch_toolbelt_amd.inference.tiles import TileMerger
    from pytorch_toolbelt_amd.parallel import ShardedTileMerger

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        geom = TO.slicer_geometry((1100, 900), (256, 256), (128, 128))
        w = TO.pyramid_window(256, 256)[0]
        crops = geom["crops"]
        C = 2
        g = torch.Generator(device="cpu").manual_seed(7)
        views = torch.randn((len(crops), 8, C, 256, 256), generator=g)           # identical on every rank
        m = ShardedTileMerger(geom["target_shape"], C, w, crops, device=dev, partition=partition, defer=defer)
        assert (m._deferred is not None) == (defer and m.owned_rows[1] > m.owned_rows[0])
        mine = m.tiles
        kept = first_band = torch.zeros(1)
        for image_no in range(2):
            m.reset()
            for b0 in range(0, len(mine), 4):
                idx = mine[b0:b0 + 4]
                batch = views[idx].transpose(0, 1).reshape(-1, C, 256, 256).to(dev) * (image_no + 1)
                m.integrate_batch_deaugment(batch, crops[idx], group="d4")
            band = m.merge()
            assert m.merge() is band, "merge() must be idempotent"
            full = m.gather(band)
            if image_no == 0 and band is not None:
                first_band = band.clone()      # the result of image 0 must survive image 1 (no shared output buffer)
                kept = band
        assert torch.equal(kept, first_band), "the band returned for the first image was overwritten by the second"
        if m._deferred is not None:
            # the one-launch finish (received partial sums added + division, ptb_band_plan_finish_rank) against its separate
            # ptb_rect_add / ptb_merge_div_ex launches: the same additions in the same order, so the same bits
            from pytorch_toolbelt_amd import _native as N

            assert N.load().ptb_set_tunable(18, 0) == 0
            try:
                m.reset()
                for b0 in range(0, len(mine), 4):
                    idx = mine[b0:b0 + 4]
                    m.integrate_batch_deaugment(views[idx].transpose(0, 1).reshape(-1, C, 256, 256).to(dev) * 2, crops[idx], group="d4")
                again = m.merge()
            finally:
                N.load().ptb_set_tunable(18, 1)
            assert torch.equal(again, band), "fused finish differs from the separate launches"
        if rank == 0:
            from oracle import tta_oracle as AO

            st = TO.merger_new(geom["target_shape"], C, w)
            for b0 in range(0, len(crops), 4):
                idx = np.arange(b0, min(b0 + 4, len(crops)))
                y = (views[idx].transpose(0, 1).reshape(-1, C, 256, 256) * 2).numpy()
                TO.merger_integrate(st, AO.image_deaugment(y, "d4", "mean"), crops[idx])
            oracle = torch.from_numpy(TO.merger_merge(st)).to(dev)
            ref = TileMerger(geom["target_shape"], C, w, device=dev)
            for b0 in range(0, len(crops), 4):
                idx = np.arange(b0, min(b0 + 4, len(crops)))
                ref.integrate_batch_deaugment(views[idx].transpose(0, 1).reshape(-1, C, 256, 256).to(dev) * 2, crops[idx], group="d4")
            want = ref.merge()
            q.put((bool(torch.isfinite(full).all()), float((full - oracle).abs().max()), float((full - want).abs().max())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("defer", [True, False], ids=["deferred-bands", "incremental"])
@pytest.mark.parametrize("world,partition", [(2, "tiles"), (3, "tiles"), (3, "rows"), (5, "tiles")])      # (5: a middle rank receives FOUR rectangles -- the one-launch finish at its limit)
def test_sharded_merger_processes_on_one_gpu(world, partition, defer):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, partition, defer, q)) for r in range(world)]
    for p in procs:
        p.start()
    finite, maxdiff, vs_single = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert finite and maxdiff <= 1e-5 and vs_single <= 1e-5, (maxdiff, vs_single)


def test_rect_add_and_partial_zero_fill():
    """ptb_rect_add on aligned / unaligned rectangles, and TileMerger._zero_fresh touching only the requested blocks."""
    from pytorch_toolbelt_amd import _native as N
    from pytorch_toolbelt_amd.inference.tiles import TileMerger
    from pytorch_toolbelt_amd.parallel import _HipOps

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    for (r0, r1, c0, c1) in [(32, 96, 64, 320), (5, 77, 3, 201), (0, 200, 0, 384), (199, 200, 383, 384)]:
        image = torch.randn((3, 200, 384), device=dev)
        want = image.clone()
        buf = torch.randn((3, r1 - r0, c1 - c0), device=dev)
        want[:, r0:r1, c0:c1] += buf
        before = N.calls
        _HipOps.add_rect(image[:, 10:] if r0 >= 10 else image, 10 if r0 >= 10 else 0, (r0, r1, c0, c1), buf)
        assert N.calls == before + 1 and torch.equal(image, want)
    m = TileMerger((256, 512), 2, np.ones((64, 64), dtype=np.float32), device=dev)
    m._image.fill_(7.0)                      # stale content of the uninitialised buffer
    m.integrate_batch(torch.ones((1, 2, 64, 64), device=dev), [(64, 32, 64, 64)])
    m._zero_fresh(32, 96, 0, 256)           # rows 32:96 x cols 0:256 -> blocks rows 1..2, cols 0..3
    got = m._image.clone()
    assert torch.equal(got[:, 32:96, 64:128], torch.ones((2, 64, 64), device=dev))
    assert float(got[:, 32:96, 0:64].abs().max()) == 0 and float(got[:, 32:96, 128:256].abs().max()) == 0
    assert float((got[:, :32] - 7).abs().max()) == 0 and float((got[:, 96:] - 7).abs().max()) == 0 and float((got[:, :, 256:] - 7).abs().max()) == 0
    assert m._fresh[1:3, :4].sum() == 0 and m._fresh.sum() == m._fresh.size - 8
    full = m.image                           # the rest is zero-filled on the first public read
    assert float(full.sum()) == 2 * 64 * 64


def test_halo_exchange_over_rccl_single_rank():
    """ptb_halo_exchange (one ncclGroup of ncclSend / ncclRecv posted from C, RCCL bound at run time) with the only topology a
    one-GPU box offers: a one-rank communicator sending two rectangles to itself.  Through the C ABI and through
    parallel.RcclExchange (side stream, event hand-off).  More than one rank has never been run (no multi-GPU box in the pool)."""
    import ctypes

    from pytorch_toolbelt_amd import _native as N
    from pytorch_toolbelt_amd.parallel import RcclExchange

    lib = N.load()
    assert lib.ptb_rccl_available() == 1
    dev = torch.device("cuda:0")
    ident = ctypes.create_string_buffer(128)
    assert lib.ptb_rccl_unique_id(ident) == 0 and any(ident.raw)
    comm = ctypes.c_void_p()
    assert lib.ptb_rccl_comm_init(ident, 1, 0, ctypes.byref(comm)) == 0 and comm.value
    a, b = torch.randn((4, 64, 320), device=dev), torch.randn((4, 32, 128), device=dev)
    ra, rb = torch.zeros_like(a), torch.zeros_like(b)
    ptrs = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])            # noqa: E731
    counts = lambda ts: (ctypes.c_int64 * len(ts))(*[t.numel() for t in ts])                # noqa: E731
    peers = (ctypes.c_int * 2)(0, 0)
    rc = lib.ptb_halo_exchange(comm, 2, ptrs([a, b]), counts([a, b]), peers, 2, ptrs([ra, rb]), counts([ra, rb]), peers, N.stream_ptr(dev))
    assert rc == 0, lib.ptb_last_hip_error()
    torch.cuda.synchronize()
    assert torch.equal(ra, a) and torch.equal(rb, b)
    assert lib.ptb_halo_exchange(comm, 0, None, None, None, 0, None, None, None, N.stream_ptr(dev)) == 0
    assert lib.ptb_halo_exchange(None, 0, None, None, None, 0, None, None, None, N.stream_ptr(dev)) == -1
    assert lib.ptb_rccl_comm_destroy(comm) == 0

    class _One:
        @staticmethod
        def get_rank(group=None):
            return 0

        @staticmethod
        def get_world_size(group=None):
            return 1

    ex = RcclExchange(dev, dist=_One)
    src = torch.randn((4, 16, 256), device=dev)
    dst = torch.zeros_like(src)
    ev = torch.cuda.Event()
    src.mul_(2.0)
    ev.record()
    ex.post([(src, 0)], [(dst, 0)], after_event=ev)
    ex.wait()
    assert torch.equal(dst, src)          # (the current stream waited for the side stream)
    ex.close()


def test_shared_exchange_belongs_to_the_group_object_not_to_its_id():
    """the per-process cache of RcclExchange objects was keyed by id(group); a destroyed group's id can be handed to a
    new group, which then got a communicator bound to the dead one (or a stale "fell back" None).  Entries hold a weak reference to the
    group OBJECT: same group -> same exchange, another group -> another one, a destroyed group -> its exchange is closed and gone."""
    import os
    import subprocess
    import sys

    from conftest import ROOT

    code = (
        "import os, torch, torch.distributed as dist\n"
        "from pytorch_toolbelt_amd import parallel as P\n"
        "dev = torch.device('cuda:0'); torch.cuda.set_device(dev)\n"
        "dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)\n"
        "e1 = P.shared_exchange(dev); assert isinstance(e1, P.RcclExchange) and P.shared_exchange(dev) is e1 and P.last_exchange_error() is None\n"
        "g = dist.new_group([0]); e2 = P.shared_exchange(dev, g); assert isinstance(e2, P.RcclExchange) and e2 is not e1 and P.shared_exchange(dev, g) is e2\n"
        "del g\n"
        "dist.destroy_process_group()\n"
        "dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)\n"
        "e3 = P.shared_exchange(dev); assert isinstance(e3, P.RcclExchange) and e3 is not e1 and e3 is not e2\n"
        "assert e1.comm is None, 'the exchange of the destroyed default group must have been closed'\n"
        "a = torch.randn(1024, device=dev); b = torch.zeros_like(a); e3.post([(a, 0)], [(b, 0)]); e3.wait(); torch.cuda.synchronize(); assert torch.equal(a, b)\n"
        "P.close_shared_exchanges(); assert e3.comm is None and not P._shared_exchanges\n"
        "dist.destroy_process_group(); print('OK')\n")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


# ------------------------------------------------------------------ pipelined exchange + communication-free partition with the real kernels
def _pipeline_worker(rank, world, port, partition, defer, backend, q):
    import torch.distributed as dist

    from oracle import tiles_oracle as TO
    from oracle import tta_oracle as AO
    from pytorch_toolbelt_amd.parallel import RcclExchange, ShardedTileMerger

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    same_gpu = backend == "gloo"
    dev = torch.device("cuda", 0 if same_gpu else rank)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        geom = TO.slicer_geometry((1100, 900), (256, 256), (128, 128))
        w = TO.pyramid_window(256, 256)[0]
        crops = geom["crops"]
        C = 2
        g = torch.Generator(device="cpu").manual_seed(9)
        views = torch.randn((len(crops), 8, C, 256, 256), generator=g)           # identical on every rank
        m = ShardedTileMerger(geom["target_shape"], C, w, crops, device=dev, partition=partition, defer=defer)
        if backend == "nccl":
            assert isinstance(m.exchange, RcclExchange), "under the nccl backend the rectangles must travel as the library's own ncclGroup"
        mine = m.tiles

        def feed(scale):
            for b0 in range(0, len(mine), 4):
                idx = mine[b0:b0 + 4]
                m.integrate_batch_deaugment(views[idx].transpose(0, 1).reshape(-1, C, 256, 256).to(dev) * scale, crops[idx], group="d4")

        tickets, bands = [], []
        for i in range(4):
            feed(i + 1)
            tickets.append(m.merge_async())
            if i:
                bands.append(tickets[i - 1].result())
        bands.append(tickets[-1].result())
        kept = [None if b is None else b.clone() for b in bands]
        for i in range(4):             # the synchronous form: same bits; and the pipelined results were not overwritten meanwhile
            m.reset()
            feed(i + 1)
            band = m.merge()
            if band is not None:
                assert torch.equal(band, bands[i]) and torch.equal(bands[i], kept[i]), f"image {i}"
        full = m.gather(bands[1])
        if rank == 0:
            st = TO.merger_new(geom["target_shape"], C, w)
            for b0 in range(0, len(crops), 4):
                idx = np.arange(b0, min(b0 + 4, len(crops)))
                y = (views[idx].transpose(0, 1).reshape(-1, C, 256, 256) * 2).numpy()
                TO.merger_integrate(st, AO.image_deaugment(y, "d4", "mean"), crops[idx])
            oracle = torch.from_numpy(TO.merger_merge(st)).to(dev)
            q.put((bool(torch.isfinite(full).all()), float((full - oracle).abs().max()), len(m._slots)))
    finally:
        dist.destroy_process_group()


def _run_pipeline(world, partition, defer, backend):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, world, port, partition, defer, backend, q)) for r in range(world)]
    for p in procs:
        p.start()
    finite, maxdiff, slots = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert finite and maxdiff <= 1e-5 and slots == 2, (finite, maxdiff, slots)
    return maxdiff


@pytest.mark.parametrize("defer", [True, False], ids=["deferred-bands", "incremental"])
@pytest.mark.parametrize("world,partition", [(2, "tiles"), (3, "tiles"), (3, "pixel_rows"), (4, "rows")])
def test_pipelined_sharded_merger_processes_on_one_gpu(world, partition, defer):
    """merge_async() with the real kernels (processes sharing the GPU, gloo): image i's exchange in flight while image i + 1 is
    merged into the second set of buffers; results bit-identical to the synchronous merge(), within 1e-5 of the oracle."""
    _run_pipeline(world, partition, defer, "gloo")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
@pytest.mark.parametrize("defer", [True, False], ids=["deferred-bands", "incremental"])
def test_halo_exchange_over_rccl_two_real_ranks(defer):
    """Two ranks on two GPUs under the nccl backend: the default exchange is the library's own ncclGroup (ptb_halo_exchange on a side
    stream behind the pack event), pipelined over four images."""
    _run_pipeline(2, "tiles", defer, "nccl")


@pytest.mark.parametrize("world", [2, 3, 5])
def test_pixel_row_ranks_reproduce_the_single_device_bits(world):
    """partition="pixel_rows" on the deferred band plan (ptb_band_plan_create3 with PTB_PLAN_CLIP_ROWS: every tile that touches the
    owned pixel rows is read on exactly those rows): the assembled map equals the single-device TileMerger bit for bit, nothing
    is exchanged.  The ranks are played one after the other in this process."""
    from oracle import tiles_oracle as TO
    from pytorch_toolbelt_amd.inference.tiles import TileMerger
    from pytorch_toolbelt_amd.parallel import ShardedTileMerger

    class _One:
        def __init__(self, rank, world):
            self.rank, self.world = rank, world

        def get_rank(self, group=None):
            return self.rank

        def get_world_size(self, group=None):
            return self.world

    dev = torch.device("cuda:0")
    geom = TO.slicer_geometry((1100, 900), (256, 256), (128, 128))
    w = TO.pyramid_window(256, 256)[0]
    crops = geom["crops"]
    C = 3
    g = torch.Generator(device="cpu").manual_seed(3)
    views = torch.randn((len(crops), 8, C, 256, 256), generator=g).to(dev)
    ref = TileMerger(geom["target_shape"], C, w, device=dev)
    for b0 in range(0, len(crops), 4):
        idx = np.arange(b0, min(b0 + 4, len(crops)))
        ref.integrate_batch_deaugment(views[idx].transpose(0, 1).reshape(-1, C, 256, 256).contiguous(), crops[idx], group="d4")
    want = ref.merge()
    for defer in (True, False):
        full = torch.full_like(want, float("nan"))
        for r in range(world):
            m = ShardedTileMerger(geom["target_shape"], C, w, crops, device=dev, dist=_One(r, world), partition="pixel_rows", defer=defer)
            assert (m._deferred is not None) == defer and not m.sends and not m.recvs
            for image_no in range(2):
                for b0 in range(0, len(m.tiles), 4):
                    idx = m.tiles[b0:b0 + 4]
                    m.integrate_batch_deaugment(views[idx].transpose(0, 1).reshape(-1, C, 256, 256).contiguous(), crops[idx], group="d4")
                band = m.merge_async().result()
            o0, o1 = m.owned_rows
            full[:, o0:o1] = band
        assert torch.equal(full, want), f"defer={defer}: max diff {float((full - want).abs().max())}"
