"""Lovasz hinge / Lovasz-softmax losses (drop-in for ``pytorch_toolbelt.losses.lovasz``).

Algorithm of Berman et al. 2018 as used by the reference (losses/lovasz.py): sort the per-pixel errors of one class in
decreasing order, weight them by the discrete gradient of the Jaccard index along that order, sum.  Here all classes
(and all images when ``per_image``) are segments of ONE hand-written segmented radix sort followed by one fused scan/dot kernel,
with no host synchronisation (the reference syncs once per class to test ``fg.sum() == 0``).
"""
from typing import Optional, Union

import torch
from torch.nn.modules.loss import _Loss

from .. import _native as N
from . import _host as H
from . import _kernels as K

__all__ = ["BinaryLovaszLoss", "LovaszLoss"]

_SOFTMAX, _HINGE = 0, 1
_CHUNK = 2048
KEY_ONLY_FORWARD = True    # False: a forward without gradient also sorts (key, index << 1 | fg) pairs (A/B and tests)
FUSED_TAIL = True          # False: the gscale * coef product and the zero "gradient" of fg_total as launches in front of the backward kernel (A/B and tests)
BINNED_GRADIENT = True     # False: the gradient is scattered to pixel order in the forward (ptb_lovasz_fwd / ptb_lovasz_bwd; A/B and tests)


class _LovaszSegments(torch.autograd.Function):
    """Per-segment Lovasz dot products [S] (float64) and per-segment foreground counts [S] (int32).  With ``present_only`` set
    (True: classes="present", False: all classes) the first result is the module's scalar instead (float32): the mean over
    the selected classes and over the groups is one more kernel (``ptb_lovasz_reduce``), not a dozen [S]-sized torch launches."""

    @staticmethod
    def forward(ctx, pred, labels, flabels, mode, per_image, has_ignore, ignore_label, ignore_value, want_grad=True, present_only=None):
        if mode == _SOFTMAX:
            B, C, HW = pred.shape
        else:
            (B, HW), C = pred.shape, 1
        groups = B if per_image else 1
        P = HW if per_image else B * HW
        S = groups * C
        n = P * S
        dev = pred.device
        alloc = torch.empty if n > 0 else torch.zeros     # (the forward entry point zeroes / fills both itself)
        seg_loss = alloc(S, dtype=torch.float64, device=dev)
        fg_total = alloc(S, dtype=torch.int32, device=dev)
        # want_grad False (the caller saw no_grad / a detached input): forward only, the per-pixel gradient is never written
        want_grad = bool(want_grad and ctx.needs_input_grad[0])
        gpix = torch.empty(0, dtype=torch.float32, device=dev)       # (only the scattered-gradient path fills one: n floats)
        binned = None
        ctx.set_materialize_grads(not FUSED_TAIL)      # (no zero tensor for the int32 fg_total output's "gradient": one launch less per backward)
        if n > 0 and not want_grad and KEY_ONLY_FORWARD:
            # evaluation / no_grad: a key-only sort (ptb_lovasz_fwd_keys) -- the foreground flag rides in the key, no (index, fg)
            # values exist: half the bytes per pass, two work arrays instead of four
            lib = N.load()
            keys = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(2)]
            chunk = torch.empty(S * ((P + _CHUNK - 1) // _CHUNK), dtype=torch.int32, device=dev)
            with N.on_device(dev):
                tb = lib.ptb_lovasz_temp_bytes(P, S)
                if tb < 0:
                    raise RuntimeError("ptb_lovasz_temp_bytes failed")
                temp = torch.empty(max(int(tb), 1), dtype=torch.uint8, device=dev)
                args = (pred.data_ptr(), K._ptr(labels), K._ptr(flabels), B, C, HW, mode, 1 if per_image else 0,
                        1 if has_ignore else 0, ignore_label, ignore_value, keys[0].data_ptr(), keys[1].data_ptr(),
                        chunk.data_ptr(), fg_total.data_ptr(), seg_loss.data_ptr(), temp.data_ptr(), int(tb))
                rc = lib.ptb_lovasz_fwd_keys(*args, N.stream_ptr(dev))
            N.bump()
            N.check(rc, "ptb_lovasz_fwd_keys")
        elif n > 0:
            lib = N.load()
            # (four allocations, not two [2, n] ones: the backward keeps only the binned pair keys[1] / vals[1] alive)
            keys = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(2)]
            vals = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(2)]
            chunk = torch.empty(S * ((P + _CHUNK - 1) // _CHUNK), dtype=torch.int32, device=dev)
            with N.on_device(dev):
                tb = lib.ptb_lovasz_temp_bytes(P, S)
                if tb < 0:
                    raise RuntimeError("ptb_lovasz_temp_bytes failed")
                temp = torch.empty(max(int(tb), 1), dtype=torch.uint8, device=dev)
                head = (pred.data_ptr(), K._ptr(labels), K._ptr(flabels), B, C, HW, mode, 1 if per_image else 0,
                        1 if has_ignore else 0, ignore_label, ignore_value, keys[0].data_ptr(), keys[1].data_ptr(),
                        vals[0].data_ptr(), vals[1].data_ptr(), chunk.data_ptr(), fg_total.data_ptr(), seg_loss.data_ptr())
                tail = (temp.data_ptr(), int(tb), N.stream_ptr(dev))
                rc = N.PTB_EUNSUPPORTED
                if want_grad and BINNED_GRADIENT:
                    # the gradient stays binned by pixel block (one more pass of the sort's scatter instead of n random writes);
                    # keys[1] / vals[1] hold the pairs for the backward kernel
                    rc = lib.ptb_lovasz_fwd_binned(*head, None, *tail)
                    if rc >= 0:
                        binned = (vals[1], keys[1], int(rc))
                        rc = 0
                if rc == N.PTB_EUNSUPPORTED:
                    if want_grad:
                        gpix = torch.empty(n, dtype=torch.float32, device=dev)
                    rc = lib.ptb_lovasz_fwd(*head, gpix.data_ptr() if want_grad else None, *tail)
            N.bump()
            N.check(rc, "ptb_lovasz_fwd")
        coef_unit = None
        if present_only is not None:
            loss = torch.empty((), dtype=torch.float32, device=dev)
            coef_unit = torch.empty(S, dtype=torch.float32, device=dev)
            with N.on_device(dev):
                rc = N.load().ptb_lovasz_reduce(seg_loss.data_ptr(), fg_total.data_ptr(), groups, C, 1 if present_only else 0, loss.data_ptr(),
                                                coef_unit.data_ptr(), N.stream_ptr(dev))
            N.check(rc, "ptb_lovasz_reduce")
            seg_loss = loss
        if binned is not None:
            ctx.save_for_backward(pred, labels, flabels, gpix, coef_unit, binned[0], binned[1])
        else:
            ctx.save_for_backward(pred, labels, flabels, gpix, coef_unit)
        ctx.block_log2 = binned[2] if binned is not None else None
        ctx.cfg = (B, C, HW, mode, per_image, has_ignore, ignore_label, ignore_value)
        ctx.mark_non_differentiable(fg_total)
        return seg_loss, fg_total

    @staticmethod
    def backward(ctx, g_loss, _g_fg):
        if ctx.block_log2 is not None:
            pred, labels, flabels, gpix, coef_unit, bgrad, bvals = ctx.saved_tensors
        else:
            pred, labels, flabels, gpix, coef_unit = ctx.saved_tensors
        B, C, HW, mode, per_image, has_ignore, ignore_label, ignore_value = ctx.cfg
        if g_loss is None:                     # (set_materialize_grads(False): nothing flows into the loss)
            return (None,) * 10
        grad = torch.empty_like(pred)          # (the kernel writes every element)
        if pred.numel():
            lib = N.load()
            g32 = g_loss.to(torch.float32)
            if FUSED_TAIL and ctx.block_log2 is not None and coef_unit is not None and g32.numel() == 1 and g32.device == pred.device:
                # the incoming gradient of the scalar loss stays a device scalar: the kernel forms gscale * coef_unit[s] itself
                g32 = g32.contiguous()
                with N.on_device(pred.device):
                    rc = lib.ptb_lovasz_bwd_binned2(pred.data_ptr(), K._ptr(labels), K._ptr(flabels), coef_unit.data_ptr(), g32.data_ptr(),
                                                    bvals.data_ptr(), bgrad.data_ptr(), grad.data_ptr(), B, C, HW, mode, 1 if per_image else 0,
                                                    1 if has_ignore else 0, ignore_label, ignore_value, ctx.block_log2, N.stream_ptr(pred.device))
                N.bump()
                N.check(rc, "ptb_lovasz_bwd_binned2")
                return grad, None, None, None, None, None, None, None, None, None
            coef = (g32 * coef_unit if coef_unit is not None else g32).contiguous()
            with N.on_device(pred.device):
                if ctx.block_log2 is not None:
                    rc = lib.ptb_lovasz_bwd_binned(pred.data_ptr(), K._ptr(labels), K._ptr(flabels), coef.data_ptr(), bvals.data_ptr(), bgrad.data_ptr(),
                                                   grad.data_ptr(), B, C, HW, mode, 1 if per_image else 0, 1 if has_ignore else 0, ignore_label,
                                                   ignore_value, ctx.block_log2, N.stream_ptr(pred.device))
                else:
                    rc = lib.ptb_lovasz_bwd(pred.data_ptr(), K._ptr(labels), K._ptr(flabels), coef.data_ptr(), gpix.data_ptr(), grad.data_ptr(),
                                        B, C, HW, mode, 1 if per_image else 0, 1 if has_ignore else 0, ignore_label, ignore_value,
                                        N.stream_ptr(pred.device))
            N.bump()
            N.check(rc, "ptb_lovasz_bwd")
        return grad, None, None, None, None, None, None, None, None, None


def _lovasz_hinge(logits, labels, per_image=True, ignore_index=None):
    """Binary Lovasz hinge loss: logits [B, H, W] (any real), labels [B, H, W] in {0, 1}; ``ignore_index`` marks void pixels.
    per_image averages the per-image losses; an image with only void pixels contributes 0."""
    if not logits.is_cuda:
        return H.lovasz_hinge(logits, labels.to(logits.device), per_image, ignore_index)
    x = K._f32c(logits, "BinaryLovaszLoss")
    B = x.shape[0]
    x = x.reshape(B, -1)
    y = labels.to(device=x.device, dtype=torch.float32).reshape(B, -1).contiguous()
    if y.shape[1] != x.shape[1]:
        raise RuntimeError(f"target shape {tuple(labels.shape)} does not match logits shape {tuple(logits.shape)}")
    if x.numel() == 0:
        seg_loss, _fg = _LovaszSegments.apply(x, None, y, _HINGE, bool(per_image), ignore_index is not None, 0,
                                              float(ignore_index) if ignore_index is not None else 0.0, torch.is_grad_enabled() and x.requires_grad)
        return seg_loss.mean().float() if per_image else seg_loss[0].float()
    loss, _fg = _LovaszSegments.apply(x, None, y, _HINGE, bool(per_image), ignore_index is not None, 0,
                                      float(ignore_index) if ignore_index is not None else 0.0, torch.is_grad_enabled() and x.requires_grad, False)
    return loss


def _lovasz_softmax(probas, labels, classes="present", per_image=False, ignore_index=None):
    """Multi-class Lovasz-softmax: ``probas`` [B, C, H, W] are class PROBABILITIES (no softmax is applied, as in the
    reference); labels [B, H, W] in [0, C).  classes: "present" (average over classes that occur), "all", or a list."""
    if not probas.is_cuda:
        return H.lovasz_softmax(probas, labels.to(probas.device), classes, per_image, ignore_index)
    if probas.dim() == 3:
        probas = probas.unsqueeze(1)
    x = K._f32c(probas, "LovaszLoss")
    B, C = x.shape[0], x.shape[1]
    if C == 1 and len(classes) > 1:
        raise ValueError("Sigmoid output possible only with 1 class")   # reference lovasz.py:129-131 (always hit for C == 1)
    x = x.reshape(B, C, -1)
    lab = labels.to(device=x.device, dtype=torch.int64).reshape(B, -1).contiguous()
    if lab.shape[1] != x.shape[2]:
        raise RuntimeError(f"target shape {tuple(labels.shape)} does not match probabilities shape {tuple(probas.shape)}")
    if classes in ("present", "all") and x.numel() > 0:
        loss, _fg = _LovaszSegments.apply(x, lab, None, _SOFTMAX, bool(per_image), ignore_index is not None,
                                          int(ignore_index) if ignore_index is not None else 0, 0.0, torch.is_grad_enabled() and x.requires_grad,
                                          classes == "present")
        return loss
    seg_loss, fg = _LovaszSegments.apply(x, lab, None, _SOFTMAX, bool(per_image), ignore_index is not None,
                                         int(ignore_index) if ignore_index is not None else 0, 0.0, torch.is_grad_enabled() and x.requires_grad)
    groups = B if per_image else 1
    seg_loss = seg_loss.view(groups, C)
    if classes == "present":
        use = (fg.view(groups, C) > 0).to(seg_loss.dtype)
    elif classes == "all":
        use = torch.ones_like(seg_loss)
    else:
        use = torch.zeros_like(seg_loss)
        use[:, list(classes)] = 1.0
    count = use.sum(dim=1)
    per_group = (seg_loss * use).sum(dim=1) / count.clamp_min(1.0)   # mean over the selected classes; 0 when none
    return per_group.mean().float() if per_image else per_group[0].float()


class BinaryLovaszLoss(_Loss):
    def __init__(self, per_image: bool = False, ignore_index: Optional[Union[int, float]] = None):
        super().__init__()
        self.ignore_index = ignore_index
        self.per_image = per_image

    def forward(self, logits, target):
        return _lovasz_hinge(logits, target, per_image=self.per_image, ignore_index=self.ignore_index)


class LovaszLoss(_Loss):
    def __init__(self, per_image=False, ignore=None):
        super().__init__()
        self.ignore = ignore
        self.per_image = per_image

    def forward(self, logits, target):
        return _lovasz_softmax(logits, target, per_image=self.per_image, ignore_index=self.ignore)
