"""Autograd functions over the elementwise + reduce loss kernels (csrc/ptb_pointwise.hip; include/ptb_hip.h
``ptb_pointwise_loss_fwd / _apply``, ``ptb_soft_ce_fwd / _bwd``).

Same contract as ``_kernels``: the forward returns the few sums a loss needs as a float64 tensor (plus, on request, the
unreduced map); means / normalisation / class-balance weights are ordinary torch algebra on scalars, so backward only
needs d(loss)/d(sum), handed to the kernels as device arrays -- nothing synchronises with the host.
"""
import torch

from .. import _native as N
from ._kernels import _ptr, check_labels, finalize, new_sums

SOFT_BCE, BALANCED_BCE, QFL, WING, LOGCOSH, SOFT_F1 = range(6)
F_IGNORE, F_SMOOTH = 1, 2


def as_f32(t, what):
    N.require_device(t, what)
    return (t if t.dtype == torch.float32 else t.float()).contiguous()


class PointwiseSums(torch.autograd.Function):
    """sums [4] float64 (meaning depends on ``kind``, see the header) and, when ``want_elem``, the per-element loss."""

    @staticmethod
    def forward(ctx, x, t, chan_w, chan_pw, kind, flags, p0, p1, p2, ignore_value, C, HW, want_elem):
        sums, _ = new_sums(4, x.device, with_flag=False)
        elem = torch.empty_like(x) if want_elem else None
        lib = N.load()
        with N.on_device(x.device):
            rc = lib.ptb_pointwise_loss_fwd(kind, x.data_ptr(), t.data_ptr(), _ptr(chan_w), _ptr(chan_pw), sums.data_ptr(), _ptr(elem),
                                            x.numel(), C, HW, flags, p0, p1, p2, ignore_value, N.stream_ptr(x.device))
        N.bump()
        N.check(rc, "ptb_pointwise_loss_fwd")
        ctx.save_for_backward(x, t, chan_w, chan_pw)
        ctx.cfg = (kind, flags, p0, p1, p2, ignore_value, C, HW)
        ctx.has_elem = want_elem
        return finalize(sums), (elem if want_elem else x.new_empty(0))

    @staticmethod
    def backward(ctx, g_sums, g_elem):
        x, t, chan_w, chan_pw = ctx.saved_tensors
        kind, flags, p0, p1, p2, ignore_value, C, HW = ctx.cfg
        g = g_sums.to(torch.float32)
        if kind == BALANCED_BCE:     # d s0 / dx = t (1 - p), d s1 / dx = -(1 - t) p; the kernel evaluates -(k0 .. - k1 ..)
            coef = torch.stack([-g[0], -g[1]])
        else:
            coef = torch.stack([g[0], g[1]])
        grad_elem = None
        if ctx.has_elem and g_elem is not None and g_elem.numel():
            # grad = (g_sums[0] + g_elem) * dL: fold the scalar into the map and use multiplier 1
            grad_elem = (g_elem.to(torch.float32) + coef[0]).contiguous()
            coef = torch.stack([torch.ones((), device=x.device), coef[1]])
        coef = coef.contiguous()
        grad = torch.empty_like(x)
        lib = N.load()
        with N.on_device(x.device):
            rc = lib.ptb_pointwise_loss_apply(kind, 0, x.data_ptr(), t.data_ptr(), _ptr(chan_w), _ptr(chan_pw), coef.data_ptr(),
                                              _ptr(grad_elem), grad.data_ptr(), x.numel(), C, HW, flags, p0, p1, p2, ignore_value,
                                              N.stream_ptr(x.device))
        N.bump()
        N.check(rc, "ptb_pointwise_loss_apply")
        return (grad,) + (None,) * 12


class BalancedElementwise(torch.autograd.Function):
    """Unreduced balanced BCE: -(w[0] * t * logsigmoid(x) + w[1] * (1 - t) * logsigmoid(-x)) with device weights w [2]."""

    @staticmethod
    def forward(ctx, x, t, w, flags, ignore_value):
        out = torch.empty_like(x)
        w = w.to(torch.float32).contiguous()
        lib = N.load()
        with N.on_device(x.device):
            rc = lib.ptb_pointwise_loss_apply(BALANCED_BCE, 1, x.data_ptr(), t.data_ptr(), None, None, w.data_ptr(), None, out.data_ptr(),
                                              x.numel(), 1, 1, flags, 0.0, 0.0, 0.0, ignore_value, N.stream_ptr(x.device))
        N.bump()
        N.check(rc, "ptb_pointwise_loss_apply")
        ctx.save_for_backward(x, t, w)
        ctx.cfg = (flags, ignore_value)
        return out

    @staticmethod
    def backward(ctx, g):
        x, t, w = ctx.saved_tensors
        flags, ignore_value = ctx.cfg
        grad = torch.empty_like(x)
        g = g.to(torch.float32).contiguous()
        lib = N.load()
        with N.on_device(x.device):
            rc = lib.ptb_pointwise_loss_apply(BALANCED_BCE, 0, x.data_ptr(), t.data_ptr(), None, None, w.data_ptr(), g.data_ptr(),
                                              grad.data_ptr(), x.numel(), 1, 1, flags, 0.0, 0.0, 0.0, ignore_value, N.stream_ptr(x.device))
        N.bump()
        N.check(rc, "ptb_pointwise_loss_apply")
        return grad, None, None, None, None


class SoftCESums(torch.autograd.Function):
    """float64 scalar sum over non-ignored pixels of (1 - eps) * nll + eps / C * smooth for [B, C, HW] logits / int64
    [B, HW] labels (nll = lse - x_t, smooth = C * lse - sum_c x_c); optionally the per-pixel values as [B, HW]."""

    @staticmethod
    def forward(ctx, x, labels, eps, has_ignore, ignore_label, want_map):
        B, C, HW = x.shape
        sums, flag = new_sums(4, x.device)
        pix = torch.empty((B, HW), dtype=torch.float32, device=x.device) if want_map else None
        lib = N.load()
        with N.on_device(x.device):
            rc = lib.ptb_soft_ce_fwd(x.data_ptr(), labels.data_ptr(), sums.data_ptr(), _ptr(pix), flag.data_ptr(), B, C, HW, eps,
                                     1 if has_ignore else 0, ignore_label, N.stream_ptr(x.device))
        N.bump()
        N.check(rc, "ptb_soft_ce_fwd")
        check_labels(flag)
        ctx.save_for_backward(x, labels)
        ctx.cfg = (eps, has_ignore, ignore_label)
        ctx.has_map = want_map
        s = finalize(sums, flag)
        return (1.0 - eps) * s[0] + (eps / C) * s[1], (pix if want_map else x.new_empty(0))

    @staticmethod
    def backward(ctx, g_total, g_pix):
        x, labels = ctx.saved_tensors
        eps, has_ignore, ignore_label = ctx.cfg
        B, C, HW = x.shape
        coef = g_total.to(torch.float32).reshape(1)
        grad_pix = None
        if ctx.has_map and g_pix is not None and g_pix.numel():
            grad_pix = (g_pix.to(torch.float32) + coef[0]).contiguous()
            coef = torch.ones(1, device=x.device)
        coef = coef.contiguous()
        grad = torch.empty_like(x)
        lib = N.load()
        with N.on_device(x.device):
            rc = lib.ptb_soft_ce_bwd(x.data_ptr(), labels.data_ptr(), coef.data_ptr(), _ptr(grad_pix), grad.data_ptr(), B, C, HW, eps,
                                     1 if has_ignore else 0, ignore_label, N.stream_ptr(x.device))
        N.bump()
        N.check(rc, "ptb_soft_ce_bwd")
        return grad, None, None, None, None, None


class BiTemperedBinarySums(torch.autograd.Function):
    """float64 sum of the per-element binary bi-tempered losses (+ optional per-element map) of fp32 x, t [n]."""

    @staticmethod
    def forward(ctx, x, t, t1, t2, smoothing, iters, has_ignore, ignore_value, want_elem):
        sums, _ = new_sums(4, x.device, with_flag=False)
        elem = torch.empty_like(x) if want_elem else None
        lib = N.load()
        with N.on_device(x.device):
            rc = lib.ptb_bitempered_binary_fwd(x.data_ptr(), t.data_ptr(), sums.data_ptr(), _ptr(elem), x.numel(), t1, t2, smoothing, iters,
                                               1 if has_ignore else 0, ignore_value, N.stream_ptr(x.device))
        N.bump()
        N.check(rc, "ptb_bitempered_binary_fwd")
        ctx.save_for_backward(x, t)
        ctx.cfg = (t1, t2, smoothing, iters, has_ignore, ignore_value)
        ctx.has_elem = want_elem
        return finalize(sums)[0], (elem if want_elem else x.new_empty(0))

    @staticmethod
    def backward(ctx, g_sum, g_elem):
        x, t = ctx.saved_tensors
        t1, t2, smoothing, iters, has_ignore, ignore_value = ctx.cfg
        coef = g_sum.to(torch.float32).reshape(1)
        grad_elem = None
        if ctx.has_elem and g_elem is not None and g_elem.numel():
            grad_elem = (g_elem.to(torch.float32) + coef[0]).contiguous()
            coef = torch.ones(1, device=x.device)
        coef = coef.contiguous()
        grad = torch.empty_like(x)
        lib = N.load()
        with N.on_device(x.device):
            rc = lib.ptb_bitempered_binary_bwd(x.data_ptr(), t.data_ptr(), coef.data_ptr(), _ptr(grad_elem), grad.data_ptr(), x.numel(),
                                               t1, t2, smoothing, iters, 1 if has_ignore else 0, ignore_value, N.stream_ptr(x.device))
        N.bump()
        N.check(rc, "ptb_bitempered_binary_bwd")
        return (grad,) + (None,) * 8


class BiTemperedRows(torch.autograd.Function):
    """Unreduced bi-tempered loss of fp32 activations [R, K] against dense targets [R, K] (``ptb_bitempered_rows``, one wave per row)."""

    @staticmethod
    def forward(ctx, act, onehot, t1, t2, smoothing, iters):
        R, Kc = act.shape
        loss = torch.empty(R, dtype=torch.float32, device=act.device)
        lib = N.load()
        with N.on_device(act.device):
            rc = lib.ptb_bitempered_rows(act.data_ptr(), onehot.data_ptr(), None, loss.data_ptr(), R, Kc, t1, t2, smoothing, iters, 0,
                                         N.stream_ptr(act.device))
        N.bump()
        N.check(rc, "ptb_bitempered_rows")
        ctx.save_for_backward(act, onehot)
        ctx.cfg = (t1, t2, smoothing, iters)
        return loss

    @staticmethod
    def backward(ctx, g):
        act, onehot = ctx.saved_tensors
        t1, t2, smoothing, iters = ctx.cfg
        g = g.to(torch.float32).contiguous()
        grad = torch.empty_like(act)
        lib = N.load()
        with N.on_device(act.device):
            rc = lib.ptb_bitempered_rows(act.data_ptr(), onehot.data_ptr(), g.data_ptr(), grad.data_ptr(), act.shape[0], act.shape[1], t1, t2,
                                         smoothing, iters, 1, N.stream_ptr(act.device))
        N.bump()
        N.check(rc, "ptb_bitempered_rows (backward)")
        return grad, None, None, None, None, None
