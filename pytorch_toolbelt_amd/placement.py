"""Placing a long-lived buffer pool by measurement (MI355X).

Which part of device memory backs a large buffer decides 10-15 % of the speed of the tile-walking kernels of this package: sixteen
12 GB pools allocated one after the other and all kept run the same tiled merge at 1.94 ... 2.25 ms per image, each pool reproducible
to 0.2 %, in runs of ~36 GB of the physical address space (``tools/placement_map.py``, ``profiles/r02_placement_map.txt``,
DESIGN.md section 5).  Which regions are fast differs from box to box and cannot be asked for, but it can be measured: a serving
process that keeps its model-output pool for hours can afford a second at start-up to allocate a few candidate pools one region
apart, run its own hot loop on each, keep the fastest and give the rest back.  ``choose_placement`` is that procedure; ``bench.py``
and ``tools/shard_sim.py`` use it in their untimed set-up and report every candidate.

No reference counterpart (the reference has no device-memory management); nothing here touches results.
"""
from typing import Any, Callable, Dict, Optional, Tuple

import torch

__all__ = ["choose_placement", "REGION_BYTES"]

REGION_BYTES = 36 << 30       # the granularity at which the speed level was seen to change (one HBM3E stack's worth of addresses)


def choose_placement(allocate: Callable[[], Any], measure: Callable[[Any], float], nbytes: int, device, *, first: Any = None, max_tries: int = 8,
                     stop_ratio: float = 0.89, fixed_count: Optional[int] = None, region_bytes: int = REGION_BYTES,
                     reserve_bytes: int = 40 << 30, free_bytes: Optional[Callable[[], int]] = None,
                     make_spacer: Optional[Callable[[int], Any]] = None) -> Tuple[Any, Dict[str, Any]]:
    """Return ``(pool, report)``: the fastest of several candidate pools.

    ``allocate()`` creates one candidate (any object that keeps its device tensors alive; ``nbytes`` = its size), ``measure(pool)``
    runs the caller's hot loop on it and returns a time (any unit).  Candidates are created one at a time and ALL kept while the
    search runs -- so each is backed by different memory -- with the space up to the next ``region_bytes`` boundary held as well, so
    consecutive candidates sit in different regions.  The search stops when a candidate is in the fast class (``min <= stop_ratio *
    max`` over at least 3 candidates), after ``max_tries`` candidates, or when free memory gets short; with ``fixed_count`` exactly
    that many candidates are measured whatever they show (ranks of a distributed job must run the same number of steps).  Everything
    but the winner is released to the driver (``torch.cuda.empty_cache()``).  ``first``: an already allocated pool to start from.
    ``report``: ``{"by_candidate": [...], "chosen": index}``.  ``free_bytes`` / ``make_spacer`` replace the device queries (free memory,
    the allocation that holds the gap) -- the CPU tests drive the search logic through them."""
    device = torch.device(device)
    on_gpu = device.type == "cuda"
    if free_bytes is None:
        free_bytes = lambda: torch.cuda.mem_get_info(device)[0]      # noqa: E731
    if make_spacer is None:
        make_spacer = lambda n: torch.empty(n, device=device, dtype=torch.uint8)      # noqa: E731
    cands, times, spacers = [first if first is not None else allocate()], [], []
    while True:
        times.append(float(measure(cands[-1])))
        if fixed_count is not None:
            if len(cands) >= fixed_count:
                break
        else:
            found = len(cands) >= 3 and min(times) <= stop_ratio * max(times)
            if found or len(cands) >= max_tries or free_bytes() < nbytes + (24 << 30):
                break
        try:
            gap = min(region_bytes - nbytes, free_bytes() - nbytes - reserve_bytes)
            if gap > (1 << 30):
                spacers.append(make_spacer(gap))
            cands.append(allocate())
        except RuntimeError:          # out of memory on a shared device: decide among what exists
            if fixed_count is None:
                break
            cands.append(cands[-1])   # (keeps the number of measure() calls equal on every rank)
    chosen = min(range(len(times)), key=lambda i: times[i])
    pool = cands[chosen]
    del cands, spacers
    if on_gpu:
        torch.cuda.empty_cache()
    return pool, {"by_candidate": [round(t, 4) for t in times], "chosen": chosen}
