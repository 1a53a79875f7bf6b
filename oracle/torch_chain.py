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


# ------------------------------------------------------------------------------------------------ secondary configs (bench.py)
# The op chains the reference runs for BASELINE configs[3] (losses) and configs[4] (multiscale + fliplr TTA), as torch ops on
# whatever device the inputs live on; `bench.py` times them on the host CPU next to the HIP kernels (baseline only).
def binary_focal_multiclass_dice_jaccard(logits: torch.Tensor, labels: torch.Tensor):
    """BinaryFocalLoss() + DiceLoss("multiclass") + JaccardLoss("multiclass") on [B, C, H, W] logits and int64 [B, H, W] labels,
    op for op as the reference evaluates them (losses/focal.py:62-77 -> functional.py:60-107 with the module defaults alpha =
    None, gamma = 2; losses/dice.py:66-124; losses/jaccard.py:62-121; functional.py:188-247).  Returns the three scalars."""
    import torch.nn.functional as F

    C = logits.size(1)
    onehot = F.one_hot(labels, C).movedim(-1, 1).float()                     # the dense target BinaryFocalLoss is fed with
    p = torch.sigmoid(logits)
    ce = F.binary_cross_entropy_with_logits(logits, onehot, reduction="none")
    pt = p * onehot + (1 - p) * (1 - onehot)
    focal = ((1.0 - pt).pow(2.0) * ce).mean()

    def region(score_fn, eps=1e-7):
        prob = logits.log_softmax(dim=1).exp().view(logits.size(0), C, -1)
        true = F.one_hot(labels.view(labels.size(0), -1), C).permute(0, 2, 1)
        inter = torch.sum(prob * true.type_as(prob), dim=(0, 2))
        card = torch.sum(prob + true.type_as(prob), dim=(0, 2))
        loss = 1.0 - score_fn(inter, card, eps)
        return (loss * (true.sum((0, 2)) > 0).to(loss.dtype)).mean()

    dice = region(lambda i, c, e: (2.0 * i) / c.clamp_min(e))
    jaccard = region(lambda i, c, e: i / (c - i).clamp_min(e))
    return focal, dice, jaccard


def lovasz_softmax(probas: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """LovaszLoss() on [B, C, H, W] probabilities, classes="present", whole batch (losses/lovasz.py:92-140, :23-38): per class
    |fg - p| sorted descending, dotted with the gradient of the Jaccard extension."""
    C = probas.size(1)
    flat = probas.movedim(1, -1).reshape(-1, C)
    lab = labels.reshape(-1)
    losses = []
    for c in range(C):
        fg = (lab == c).type_as(flat)
        if fg.sum() == 0:
            continue
        err, perm = torch.sort((fg - flat[:, c]).abs(), 0, descending=True)
        gts = fg[perm]
        inter = gts.sum() - gts.cumsum(0)
        union = gts.sum() + (1.0 - gts).cumsum(0)
        jac = 1.0 - inter / union
        if len(gts) > 1:
            jac[1:] = jac[1:] - jac[:-1]
        losses.append(torch.dot(err, jac))
    return sum(losses) / max(len(losses), 1)


def ms_fliplr_deaugment(outputs, size_offsets, reduction="gmean", align_corners=False):
    """configs[4]: every scale's fliplr-TTA output de-augmented (tta.py:287-300), resized back with F.interpolate and the scales
    merged (tta.py:645-689) -- two views per scale, `reduction` inside every scale and across the scales."""
    import torch.nn.functional as F

    per_scale = [image_deaugment(y, "fliplr", reduction) for y in outputs]
    restored = []
    for fmap, off in zip(per_scale, size_offsets):
        restored.append(fmap if off == 0 else F.interpolate(fmap, size=(fmap.size(2) - off, fmap.size(3) - off), mode="bilinear",
                                                            align_corners=align_corners))
    stack = torch.stack(restored)
    if reduction == "gmean":
        return stack.log().mean(dim=0).exp()
    return stack.mean(dim=0)
