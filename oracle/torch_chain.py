"""Oracle (test infrastructure only): the reference's CPU data flow for the headline path as multi-threaded torch-CPU ops.

`bench.py`'s `cpu_baseline` leg times THIS on all host cores of the GPU box (the reference itself cannot travel there).  It is
the op chain the reference executes on its default device ("cpu"), op for op, so that the baseline is what a user of the
reference gets on that host -- not the single-core numpy oracle:

  * de-augment (tta.py:442-467 + :63-96): chunk the [V*B, C, h, w] model output into V pieces, apply the inverse view of
    each piece (flip = copy, transpose = strided view, as torch.rot90 / .flip / .transpose do), `torch.stack` (one copy of
    every view), `.mean(dim=0)` (other reductions: the formulas of functional.py:250-333);
  * integrate (tiles.py:321-339): per tile `image[:, y:y+h, x:x+w] += tile * weight; norm_mask[...] += weight`, sequentially;
  * merge (tiles.py:345-346): `image / norm_mask`.

Views come from tta_oracle's (T, fr, fc) tables.  tests/test_fullsize_cpu.py checks it against the numpy oracle and the
reference's golden vectors, so the timed thing is known to compute the right answer.
"""
import torch

from . import tta_oracle as AO


def apply_view(x: torch.Tensor, view) -> torch.Tensor:
    T, fr, fc = view
    dims = [d for d, f in ((2, fr), (3, fc)) if f]
    y = x.flip(dims) if dims else x          # torch flips materialise (like the rot90 / flip calls of the reference)
    return y.transpose(2, 3) if T else y


def image_deaugment(y: torch.Tensor, group: str = "d4", reduction: str = "mean") -> torch.Tensor:
    views = AO.DEAUG_VIEWS[group]
    if y.size(0) % len(views):
        raise RuntimeError("batch is not divisible by the number of views")
    stack = torch.stack([apply_view(c, v) for c, v in zip(torch.chunk(y, len(views)), views)])
    if reduction == "mean":
        return stack.mean(dim=0)
    if reduction == "sum":
        return stack.sum(dim=0)
    if reduction == "gmean":
        return stack.log().mean(dim=0).exp()
    raise KeyError(reduction)


class Merger:
    """TileMerger on its default device (tiles.py:295-346)."""

    def __init__(self, target_shape, channels, weight):
        self.weight = torch.from_numpy(weight).unsqueeze(0).to(torch.float32)
        self.image = torch.zeros((channels, target_shape[0], target_shape[1]), dtype=torch.float32)
        self.norm_mask = torch.zeros((1, target_shape[0], target_shape[1]), dtype=torch.float32)

    def integrate_batch(self, batch, crops):
        for tile, (x, y, w, h) in zip(batch, crops):
            x, y, w, h = int(x), int(y), int(w), int(h)
            self.image[:, y:y + h, x:x + w] += tile * self.weight
            self.norm_mask[:, y:y + h, x:x + w] += self.weight

    def merge(self):
        return self.image / self.norm_mask


def host_description():
    """(physical cores, logical cpus, model string) of this host."""
    import os

    logical = os.cpu_count() or 1
    try:
        import psutil

        physical = psutil.cpu_count(logical=False) or logical
    except Exception:  # noqa: BLE001
        physical = logical
    try:
        allowed = len(os.sched_getaffinity(0))
    except Exception:  # noqa: BLE001
        allowed = logical
    model = "unknown CPU"
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except Exception:  # noqa: BLE001
        pass
    return max(1, min(physical, allowed)), logical, model
