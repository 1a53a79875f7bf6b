"""Shared body of DiceLoss / JaccardLoss: fused region statistics + a [C]-sized scalar epilogue."""
import torch

from ..utils.torch_utils import to_tensor
from . import _host as H
from . import _kernels as K

BINARY_MODE = "binary"
MULTICLASS_MODE = "multiclass"
MULTILABEL_MODE = "multilabel"


def prepare_classes(mode, classes):
    if classes is None:
        return None
    assert mode != BINARY_MODE, "Masking classes is not supported with mode=binary"
    return to_tensor(classes, dtype=torch.long)


def _dense_target(y_true, x, bs, C, y_pred):
    """Dense 0/1 (or soft) target as fp32 [B, C, HW]; the kernels take B, C, HW from the prediction, so a target with another
    element count (half-resolution mask, wrong channel count) must fail here instead of being read past its end."""
    dense = K._f32c(y_true.to(device=x.device), "region loss")
    if dense.numel() != x.numel():
        raise RuntimeError(f"target shape {tuple(y_true.shape)} does not match prediction shape {tuple(y_pred.shape)}")
    return dense.reshape(bs, C, -1)


def region_statistics(y_pred, y_true, mode, from_logits, ignore_index):
    """(I, P, T) per class over batch and pixels: one HIP pass over the logits; the softmax / sigmoid probabilities, the
    one-hot targets and the ignore mask are formed in registers (reference losses/dice.py:68-111)."""
    assert y_true.size(0) == y_pred.size(0)
    if not y_pred.is_cuda:      # host tensors: torch algebra (losses/_host.py), still summed over the ranks inside sync_region_statistics
        from ..parallel import sync_region_statistics

        return sync_region_statistics.apply(H.region_statistics(y_pred, y_true, mode, from_logits, ignore_index))
    bs = y_true.size(0)
    x = K._f32c(y_pred, "region loss")
    if mode == MULTICLASS_MODE:
        x = x.reshape(bs, x.size(1), -1)
        labels = y_true.to(device=x.device).reshape(bs, -1)
        if labels.dtype != torch.int64:
            labels = labels.long()
        labels = labels.contiguous()
        if labels.size(1) != x.size(2):
            raise RuntimeError(f"target shape {tuple(y_true.shape)} does not match prediction shape {tuple(y_pred.shape)}")
        prob = K.PROB_SOFTMAX if from_logits else K.PROB_IDENTITY
        stats = K.RegionStats.apply(x, labels, None, prob, ignore_index is not None, int(ignore_index) if ignore_index is not None else 0, 0.0)
    else:
        C = 1 if mode == BINARY_MODE else x.size(1)
        x = x.reshape(bs, C, -1)
        dense = _dense_target(y_true, x, bs, C, y_pred)
        prob = K.PROB_SIGMOID if from_logits else K.PROB_IDENTITY
        stats = K.RegionStats.apply(x, None, dense, prob, ignore_index is not None, 0, float(ignore_index) if ignore_index is not None else 0.0)
    from ..parallel import sync_region_statistics   # batch sharded over ranks: sum the [C] partials (no-op by default)

    return sync_region_statistics.apply((stats[0].float(), stats[1].float(), stats[2].float()))


def finish(scores, true_mass, log_loss, eps, classes):
    """1 - score (or -log score), zero for classes without positives, optional class subset, mean (dice.py:115-131)."""
    loss = -torch.log(scores.clamp_min(eps)) if log_loss else 1.0 - scores
    loss = loss * (true_mass > 0).to(loss.dtype)
    if classes is not None:
        loss = loss[classes.to(loss.device)]
    return loss.mean()


def _inputs(y_pred, y_true, mode, from_logits, ignore_index):
    """(x [B, C, HW], labels | None, dense | None, prob, has_ignore, ignore_label, ignore_value) for the kernels."""
    assert y_true.size(0) == y_pred.size(0)
    bs = y_true.size(0)
    x = K._f32c(y_pred, "region loss")
    has_ign = ignore_index is not None
    if mode == MULTICLASS_MODE:
        x = x.reshape(bs, x.size(1), -1)
        labels = y_true.to(device=x.device).reshape(bs, -1)
        if labels.dtype != torch.int64:
            labels = labels.long()
        labels = labels.contiguous()
        if labels.size(1) != x.size(2):
            raise RuntimeError(f"target shape {tuple(y_true.shape)} does not match prediction shape {tuple(y_pred.shape)}")
        return x, labels, None, (K.PROB_SOFTMAX if from_logits else K.PROB_IDENTITY), has_ign, int(ignore_index) if has_ign else 0, 0.0
    C = 1 if mode == BINARY_MODE else x.size(1)
    x = x.reshape(bs, C, -1)
    dense = _dense_target(y_true, x, bs, C, y_pred)
    return x, None, dense, (K.PROB_SIGMOID if from_logits else K.PROB_IDENTITY), has_ign, 0, float(ignore_index) if has_ign else 0.0


_mask_cache = {}


def _class_mask(classes, C, device):
    """uint8 [C] device mask of the selected classes and their count, built once per (classes, C, device) -- not per forward
    (a pageable host-to-device copy would block the host on every call).  (None, n) when `classes` holds duplicates: the
    reference indexes loss[classes], so duplicates would count twice and the caller composes the torch tail instead."""
    key = (tuple(int(c) for c in classes.reshape(-1).tolist()), int(C), str(device))
    hit = _mask_cache.get(key)
    if hit is None:
        host = torch.zeros(C, dtype=torch.uint8)
        host[list(key[0])] = 1
        n_sel = len(key[0])
        hit = (host.to(device) if int(host.sum()) == n_sel else None, n_sel)
        if len(_mask_cache) > 64:
            _mask_cache.clear()
        _mask_cache[key] = hit
    return hit


def fused_region_loss(y_pred, y_true, mode, from_logits, ignore_index, dice_weight, jaccard_weight, smooth, eps, log_loss, classes,
                      focal=None):
    """The whole loss as one autograd node (kernel + scalar-epilogue kernel), or None when the statistics have to be
    all-reduced over ranks first (``parallel.sync_region_statistics``) -- the caller then composes the torch tail.

    ``focal`` = None or dict(weight, gamma, alpha): adds ``weight * mean sigmoid focal loss`` from the same pass."""
    from ..parallel import sync_region_statistics

    if sync_region_statistics._active is not None or not y_pred.is_cuda:
        return None           # (host tensors: the caller composes region_statistics + the torch tail)
    x, labels, dense, prob, has_ign, ign_label, ign_value = _inputs(y_pred, y_true, mode, from_logits, ignore_index)
    C = x.size(1)
    mask, n_sel = None, C
    if classes is not None:
        mask, n_sel = _class_mask(classes, C, x.device)
        if mask is None:
            return None
    flags = K.SEG_HAS_IGNORE if has_ign else 0
    gamma = alpha = 0.0
    scale = 0.0
    if focal is not None:
        flags |= (K.SEG_HAS_ALPHA if focal["alpha"] is not None else 0) | K.SEG_NO_TERM    # (no normalized focal here: sums[1] is never read)
        gamma, alpha = float(focal["gamma"]), float(focal["alpha"] or 0.0)
        scale = float(focal["weight"]) / x.numel()
    return K.RegionLoss.apply(x, labels, dense, None, flags, prob, gamma, alpha, 0.0, ign_label, ign_value, focal is not None, scale,
                              float(dice_weight), float(jaccard_weight), float(smooth), float(eps), bool(log_loss), mask, n_sel)
