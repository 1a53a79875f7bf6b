"""GPU: the N > 1 path end to end with the REAL device code -- `world` processes share the one GPU of the test box and
talk over gloo (RCCL refuses two ranks on one device); each runs ShardedTileMerger with the HIP kernels, exchanges its
overlap strip point to point and merges its band.  The gathered result must equal the single-process TileMerger bit for
bit (same fp32 order per pixel: a rank's own tiles first, then the neighbour's strip -- see parallel.py)."""
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


def _worker(rank, world, port, q):
    import torch.distributed as dist

    from oracle import tiles_oracle as TO
    from pytorch_toolbelt_amd.inference.tiles import TileMerger
    from pytorch_toolbelt_amd.parallel import ShardedTileMerger, tile_row_partition

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
        m = ShardedTileMerger(geom["target_shape"], C, w, crops, device=dev)
        mine = tile_row_partition(crops, world)[rank]
        for image_no in range(2):
            m.reset()
            for b0 in range(0, len(mine), 4):
                idx = mine[b0:b0 + 4]
                batch = views[idx].transpose(0, 1).reshape(-1, C, 256, 256).to(dev) * (image_no + 1)
                m.integrate_batch_deaugment(batch, crops[idx], group="d4")
            full = m.gather(m.merge())
        if rank == 0:
            ref = TileMerger(geom["target_shape"], C, w, device=dev)
            for b0 in range(0, len(crops), 4):
                idx = np.arange(b0, min(b0 + 4, len(crops)))
                ref.integrate_batch_deaugment(views[idx].transpose(0, 1).reshape(-1, C, 256, 256).to(dev) * 2, crops[idx], group="d4")
            want = ref.merge()
            q.put((bool(torch.isfinite(full).all()), float((full - want).abs().max()), bool(torch.equal(full, want))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_merger_processes_on_one_gpu(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    finite, maxdiff, equal = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert finite and maxdiff <= 1e-5, maxdiff
