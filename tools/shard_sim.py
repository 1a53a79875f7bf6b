"""Per-rank cost of the sharded merge on ONE GPU + a model of the fabric: plays rank r of an N-rank job on the one GPU of a dev box
with the exchange stubbed out (the halo rectangles are packed and the receive buffers are used as they are), so what is MEASURED
is everything a rank does locally per image -- kernels and host issue, the point in the step at which its outgoing rectangles
are packed (`early`), and the per-image cost in pipelined mode (merge_async) -- and what is MODELLED is the xGMI transfer:

  per directed link (rank s -> rank d)   bytes(s, d) / BW + LATENCY       BW in {50, 100, 150} GB/s per link and direction
                                                                           (xGMI: ~153 GB/s nominal per link, full duplex; every pair
                                                                           of the 8 GPUs has its own link; LATENCY = 20 us per group)
  latency mode (merge() per image)       a pair's transfer starts when BOTH ends have posted (each posts once its own rectangles are
                                         packed): done(d) = max over peers p of max(early_d, early_p) + max(bytes(d,p), bytes(p,d)) / BW
                                         + LATENCY;  image(d) = max(compute_d - finish_d, done(d)) + finish_d;  image = max over ranks
  pipelined mode (merge_async())         the exchange of image i runs beside the kernels of image i + 1 and the two streams only meet
                                         at result(i): per image max(compute_d, busiest link of d / BW + LATENCY), max over ranks
  exposed exchange                       image time - slowest rank's compute

    python tools/shard_sim.py [--world 8] [--partition tiles|rows|pixel_rows] [--steps 30] [--out profiles/r04_shard_sim.txt]
"""
import argparse
import gc
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from pytorch_toolbelt_amd.inference.tiles import ImageSlicer, TileMerger  # noqa: E402
from pytorch_toolbelt_amd.parallel import ShardedTileMerger  # noqa: E402

LATENCY_MS = 0.020
LINK_GBS = (50.0, 100.0, 150.0)


class _Work:
    def wait(self):
        return True


class FakeDist:
    """Rank / world stand-in whose `batch_isend_irecv` transfers nothing but records WHEN the merger posted its exchange."""

    def __init__(self, rank, world):
        self.rank, self.world = rank, world
        self.post_events = []

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

    def batch_isend_irecv(self, ops):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self.post_events.append(ev)
        return [_Work()]


def play_rank(args, slicer, r, dev, C=4, V=8, B=8):
    fd = FakeDist(r, args.world)
    m = ShardedTileMerger(slicer.target_shape, C, slicer.weight, slicer.crops, device=dev, dist=fd, partition=args.partition, defer=not args.no_defer)
    if m.local is None:
        return None
    for buf in m._recv_buf:
        buf.zero_()
    crops = slicer.crops[m.tiles]
    batches = [(b0, min(len(crops), b0 + B)) for b0 in range(0, len(crops), B)]
    outs = torch.randn((V * len(crops), C, 512, 512), device=dev)
    state = {"outs": outs}

    def feed():
        o = state["outs"]
        for b0, b1 in batches:
            m.integrate_batch_deaugment(o[V * b0:V * b1], crops[b0:b1], group="d4", reduction="mean")

    def step():
        m.reset()
        feed()
        return m.merge()

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    placed = ""
    if args.placement_tries > 1:
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
        placed = " placement candidates " + "/".join(f"{t:.3f}" for t in rep["by_candidate"]) + ";"
        torch.cuda.empty_cache()
    # ---- synchronous steps: device time per image, host issue, and where in the step the exchange is posted
    devms, host = float("inf"), float("inf")
    for _blk in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(args.steps):
            step()
        e1.record()
        host = min(host, (time.perf_counter() - t0) / args.steps * 1e3)
        torch.cuda.synchronize()
        devms = min(devms, e0.elapsed_time(e1) / args.steps)
    early, finish = [], []
    for _ in range(10):
        fd.post_events.clear()
        s0, s1, s2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        m.reset()
        s0.record()
        feed()
        s1.record()
        m.merge()
        s2.record()
        torch.cuda.synchronize()
        if fd.post_events:
            early.append(s0.elapsed_time(fd.post_events[0]))
        finish.append(s1.elapsed_time(s2))
    early_ms = float(np.median(early)) if early else 0.0
    finish_ms = float(np.median(finish))
    # ---- pipelined steps: merge_async() per image, the previous image completed after the next one was fed
    pend = None
    m.reset()                 # (the last synchronous image still sits in the current buffers: merge_async() only resets the NEXT ones)
    for _ in range(5):
        feed()
        t = m.merge_async()
        if pend is not None:
            pend.result()
        pend = t
    torch.cuda.synchronize()
    pipe = float("inf")
    for _blk in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            feed()
            t = m.merge_async()
            pend.result()
            pend = t
        e1.record()
        torch.cuda.synchronize()
        pipe = min(pipe, e0.elapsed_time(e1) / args.steps)
    pend.result()
    out_b, in_b = {}, {}       # bytes per directed link: a neighbour may get several rectangles (round 6: tight rectangles per row run)
    for d, r0, r1, c0, c1 in m.sends:
        out_b[int(d)] = out_b.get(int(d), 0) + (r1 - r0) * (c1 - c0) * C * 4
    for s_, r0, r1, c0, c1 in m.recvs:
        in_b[int(s_)] = in_b.get(int(s_), 0) + (r1 - r0) * (c1 - c0) * C * 4
    rec = dict(rank=r, tiles=len(crops), compute_ms=devms, pipelined_ms=pipe, early_ms=early_ms, finish_ms=finish_ms, host_ms=host, out=out_b, inn=in_b,
               owned=m.owned_rows, mode="deferred bands" if m._deferred is not None else "incremental", boundary=len(m.plan[r]["boundary"]), placed=placed)
    del m, outs
    state.clear()
    torch.cuda.empty_cache()
    return rec


def single_gpu_ms(slicer, dev, steps, C=4, V=8, B=8):
    """The one-GPU merge of the same image with the same merger family (TileMerger(crops=, defer=True)), first allocation."""
    crops = slicer.crops
    m = TileMerger(slicer.target_shape, C, slicer.weight, device=dev, crops=crops, defer=True)
    batches = [(b0, min(len(crops), b0 + B)) for b0 in range(0, len(crops), B)]
    outs = [torch.randn((V * (b1 - b0), C, 512, 512), device=dev) for b0, b1 in batches]

    def step():
        m.reset()
        for t, (b0, b1) in zip(outs, batches):
            m.integrate_batch_deaugment(t, crops[b0:b1], group="d4", reduction="mean")
        return m.merge()

    for _ in range(20):
        step()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            step()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / steps)
    del m, outs
    torch.cuda.empty_cache()
    return best


def fabric_model(recs, gbs):
    """(latency-mode image ms, pipelined image ms) for a link bandwidth of `gbs` GB/s per direction."""
    by = {r["rank"]: r for r in recs}
    lat, pipe = 0.0, 0.0
    for d in recs:
        done = 0.0
        peers = set(d["out"]) | set(d["inn"])
        for p in peers:
            nbytes = max(d["out"].get(p, 0), d["inn"].get(p, 0))
            start = max(d["early_ms"], by[p]["early_ms"] if p in by else d["early_ms"])
            done = max(done, start + nbytes / (gbs * 1e6) + LATENCY_MS)
        lat = max(lat, max(d["compute_ms"] - d["finish_ms"], done) + d["finish_ms"])
        link = max([0] + [max(d["out"].get(p, 0), d["inn"].get(p, 0)) for p in peers])
        pipe = max(pipe, max(d["pipelined_ms"], link / (gbs * 1e6) + LATENCY_MS if link else 0.0))
    return lat, pipe


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--partition", default="tiles")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--placement-tries", type=int, default=1, help="candidate placements of the rank's model outputs (1 = first allocation)")
    ap.add_argument("--ranks", default="", help="comma-separated ranks to play (default: all)")
    ap.add_argument("--no-defer", action="store_true", help="incremental accumulate + exchange path (round-1 behaviour) instead of the band plan")
    ap.add_argument("--single-ms", type=float, default=0.0, help="one-GPU ms per image to compute speed-ups against (0: measure it here)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    slicer = ImageSlicer((5000, 5000, 3), 512, 256, weight="pyramid")
    gc.collect()
    gc.freeze()
    gc.disable()   # a generation-2 pass (30-45 ms) inside one rank's loop would masquerade as a slow rank
    single = args.single_ms or single_gpu_ms(slicer, dev, args.steps)
    recs = []
    for r in ([int(x) for x in args.ranks.split(",")] if args.ranks else range(args.world)):
        rec = play_rank(args, slicer, r, dev)
        if rec is None:
            print(f"rank {r}/{args.world}: no tiles")
            continue
        recs.append(rec)
        links = ", ".join(f"->{p}: {rec['out'].get(p, 0) / 1e6:.1f} out / {rec['inn'].get(p, 0) / 1e6:.1f} in" for p in sorted(set(rec["out"]) | set(rec["inn"]))) or "none"
        print(f"rank {r}/{args.world} [{args.partition}, {rec['mode']}]: {rec['tiles']} tiles (boundary {rec['boundary']}), owned rows {rec['owned']};{rec['placed']} "
              f"compute {rec['compute_ms']:.3f} ms per image (pipelined {rec['pipelined_ms']:.3f}; rectangles packed at {rec['early_ms']:.3f} ms, completion of the "
              f"shared rows {rec['finish_ms']:.3f} ms; host issue {rec['host_ms']:.3f} ms); MB per link {links}")
    if not recs:
        return
    worst = max(r["compute_ms"] for r in recs)
    worst_pipe = max(r["pipelined_ms"] for r in recs)
    print(f"one GPU (TileMerger(crops=, defer=True), first allocation): {single:.3f} ms per image")
    print(f"slowest rank: {worst:.3f} ms per image synchronous, {worst_pipe:.3f} ms pipelined -> {single / worst:.2f}x / {single / worst_pipe:.2f}x if the exchange cost nothing")
    if any(r["out"] or r["inn"] for r in recs):
        for gbs in LINK_GBS:
            lat, pipe = fabric_model(recs, gbs)
            print(f"  {gbs:5.0f} GB/s per link: latency mode {lat:.3f} ms per image ({single / lat:.2f}x, exposed exchange {max(lat - worst, 0.0):.3f} ms) | "
                  f"pipelined {pipe:.3f} ms per image ({single / pipe:.2f}x, exposed exchange {max(pipe - worst_pipe, 0.0):.3f} ms)")
    else:
        print("  no exchange in this partition: the figures above are the prediction")


if __name__ == "__main__":
    main()
