"""Per-rank cost of the sharded merge WITHOUT a fabric: plays rank r of an N-rank job on the one GPU of a dev box with the
exchange stubbed out (the halo rectangles are packed, and the receive buffers are used as they are), so what is timed is
everything a rank does locally per image -- kernels and host issue.  The xGMI transfer itself is not modelled.

    python tools/shard_sim.py [--world 8] [--partition tiles] [--steps 30]
"""
import argparse
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from pytorch_toolbelt_amd.inference.tiles import ImageSlicer  # noqa: E402
from pytorch_toolbelt_amd.parallel import ShardedTileMerger  # noqa: E402


class _Work:
    def wait(self):
        return True


class FakeDist:
    def __init__(self, rank, world):
        self.rank, self.world = rank, world

    def get_rank(self, group=None):
        return self.rank

    def get_world_size(self, group=None):
        return self.world

    class P2POp:
        def __init__(self, op, tensor, peer, group=None):
            self.tensor = tensor

    @staticmethod
    def isend(*a, **k):
        pass

    @staticmethod
    def irecv(*a, **k):
        pass

    @staticmethod
    def batch_isend_irecv(ops):
        return [_Work()]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--partition", default="tiles")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--placement-tries", type=int, default=6, help="candidate placements of the rank's model outputs (1 = first allocation)")
    ap.add_argument("--ranks", default="", help="comma-separated ranks to play (default: all)")
    ap.add_argument("--no-defer", action="store_true", help="incremental accumulate + exchange path (round-1 behaviour) instead of the band plan")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    slicer = ImageSlicer((5000, 5000, 3), 512, 256, weight="pyramid")
    C, V, B = 4, 8, 8
    gc.collect()
    gc.freeze()
    gc.disable()   # a generation-2 pass (30-45 ms) inside one rank's loop would masquerade as a slow rank
    worst = 0.0
    for r in ([int(x) for x in args.ranks.split(",")] if args.ranks else range(args.world)):
        m = ShardedTileMerger(slicer.target_shape, C, slicer.weight, slicer.crops, device=dev, dist=FakeDist(r, args.world), partition=args.partition, defer=not args.no_defer)
        for buf in m._recv_buf:
            buf.zero_()
        crops = slicer.crops[m.tiles]
        batches = [(b0, min(len(crops), b0 + B)) for b0 in range(0, len(crops), B)]
        outs = torch.randn((V * len(crops), C, 512, 512), device=dev)
        state = {"outs": outs}

        def step():
            outs = state["outs"]
            m.reset()
            for b0, b1 in batches:
                m.integrate_batch_deaugment(outs[V * b0:V * b1], crops[b0:b1], group="d4", reduction="mean")
            return m.merge()

        for _ in range(5):
            step()
        torch.cuda.synchronize()
        if args.placement_tries > 1:
            # which ~36 GB region of device memory backs the model outputs decides 10-15 % of the loop (bench.py):
            # candidates one region apart, the fastest kept -- what a rank of a long-lived job would do once at start-up
            def quick(t):
                state["outs"] = t
                step()
                q0, q1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                q0.record()
                for _ in range(10):
                    step()
                q1.record()
                torch.cuda.synchronize()
                return q0.elapsed_time(q1) / 10

            from pytorch_toolbelt_amd.placement import choose_placement

            outs, rep = choose_placement(lambda: torch.randn_like(outs), quick, outs.numel() * 4, dev, first=outs, fixed_count=args.placement_tries)
            state["outs"] = outs
            times = rep["by_candidate"]
            placed = " placement candidates " + "/".join(f"{t:.3f}" for t in times) + ";"
            torch.cuda.empty_cache()
        else:
            placed = ""
        # three timed blocks, the fastest counts: a dev box shows rare 30-70 ms host stalls (allocator / other tenants) that
        # have nothing to do with the rank being played
        devms, host = float("inf"), float("inf")
        per = []
        for _blk in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            for _ in range(args.steps):
                t1 = time.perf_counter()
                step()
                per.append((time.perf_counter() - t1) * 1e3)
            e1.record()
            host = min(host, (time.perf_counter() - t0) / args.steps * 1e3)
            torch.cuda.synchronize()
            devms = min(devms, e0.elapsed_time(e1) / args.steps)
        worst = max(worst, devms)
        halo = sum((r1 - r0) * (c1 - c0) * C * 4 for _d, r0, r1, c0, c1 in m.sends) / 1e6
        print(f"rank {r}/{args.world} [{args.partition}, {'deferred bands' if m._deferred is not None else 'incremental'}]: {len(crops)} tiles, {len(batches)} launches, boundary tiles {len(m.plan[r]['boundary'])}, "
              f"owned rows {m.owned_rows}, halo out {halo:.1f} MB:{placed} {devms:.3f} ms per image (host issue {host:.3f} ms; "
              f"median step {sorted(per)[len(per) // 2]:.3f}, worst step {max(per):.3f})")
        del m, outs
    print(f"slowest rank {worst:.3f} ms per image -> {25.0 / worst * 1e3:.0f} MP/s if the exchange hides completely")


if __name__ == "__main__":
    main()
