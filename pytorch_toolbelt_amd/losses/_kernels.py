"""Autograd functions over the fused loss kernels of libptb_hip.so (ptb_seg_loss_fwd & friends).

Each function returns the *sums* the loss needs as a small float64 tensor; the remaining scalar algebra (normalise,
clamp, 1 - score, log, mean over classes) is ordinary differentiable torch code on [C]-sized tensors, so the backward
kernels only need d(loss)/d(sum) -- delivered as device arrays, without a host synchronisation.
"""
import os
import threading

import torch

from .. import _native as N

SEG_FOCAL, SEG_STATS, SEG_HAS_IGNORE, SEG_HAS_ALPHA, SEG_REDUCED, SEG_MASK_FOCAL_TERM, SEG_ELEMWISE, SEG_NO_TERM = 1, 2, 4, 8, 16, 32, 64, 128
PROB_SOFTMAX, PROB_SIGMOID, PROB_IDENTITY = 0, 1, 2
SUM_SLOTS = 64  # PTB_SUM_SLOTS: the kernels spread their fp64 atomics over this many copies of the sums

# Labels outside [0, C): the reference's F.one_hot raises on CPU and trips a device-side assert on GPU (no host sync).
# The kernels record the condition in a device flag, and the flag POISONS the result: ptb_sums_finalize / ptb_region_epilogue
# turn the sums / the loss into NaN when it is set, so a bad label can never train silently.  The exception itself is raised
# WITHOUT synchronising: the flag is copied
# to pinned host memory asynchronously and examined at the next loss call (or by ``flush_label_check()``), so a bad
# label surfaces one call late -- the same asynchronous error model as a device assert, minus the dead context.
# PTB_SYNC_LABEL_CHECK=1 (or ``SYNC_LABEL_CHECK = True``) checks immediately at the price of one host sync per call;
# PTB_SKIP_LABEL_CHECK=1 disables the check.
_CHECK_LABELS = os.environ.get("PTB_SKIP_LABEL_CHECK", "0") != "1"
SYNC_LABEL_CHECK = os.environ.get("PTB_SYNC_LABEL_CHECK", "0") == "1"
_LABEL_MSG = "Class values must be smaller than num_classes."
_pending = []  # (pinned host int32 tensor, event)
ONE_LAUNCH_REGION_LOSS = os.environ.get("PTB_ONE_LAUNCH_REGION_LOSS", "1") != "0"   # (A/B switch: 0 = kernel + finalize-free epilogue launch)
_pending_lock = threading.Lock()


def _f32c(t, what):
    N.require_device(t, what)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _ptr(t):
    return t.data_ptr() if t is not None else None


def _poll(block: bool):
    global _pending
    keep, bad = [], False
    with _pending_lock:
        todo, _pending = _pending, []
    for host, ev in todo:
        if block:
            ev.synchronize()
        if block or ev.query():
            bad = bad or int(host[0]) != 0
        else:
            keep.append((host, ev))
    if keep:
        with _pending_lock:
            _pending = keep + _pending
    if bad:
        raise RuntimeError(_LABEL_MSG + " (reported asynchronously by an earlier loss call)")


def flush_label_check():
    """Wait for the outstanding label checks and raise if any loss call saw a label outside [0, C)."""
    _poll(block=True)


def check_labels(flag):
    if not _CHECK_LABELS:
        return
    if SYNC_LABEL_CHECK:
        if int(flag.item()) != 0:
            raise RuntimeError(_LABEL_MSG)
        return
    _poll(block=False)
    host = torch.empty(1, dtype=torch.int32, pin_memory=True)
    host.copy_(flag, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(flag.device))
    with _pending_lock:
        _pending.append((host, ev))


_flag_ring = None        # pinned int32 [256]: label flags the kernels of the one-launch region loss write straight into host memory
_flag_next = 0
_workspaces = {}         # (device index, stream) -> zeroed uint8 workspace of ptb_region_loss_fwd


def host_flag_slot():
    """One int32 of pinned host memory (a ring of 256) for a kernel to write its label flag into -- no device-to-host copy on the
    stream; the slot is valid once the event recorded after the launch has completed.  Slots are handed out under the lock that
    guards the pending list (two threads must never get the same one), and a slot is not reused while a pending check still reads
    it: when half the ring is in flight the backlog is drained first."""
    global _flag_ring, _flag_next
    with _pending_lock:
        if _flag_ring is None:
            _flag_ring = torch.zeros(256, dtype=torch.int32).pin_memory()
        backlog = len(_pending)
    if backlog >= 128:      # a slot must not come round again before its check has been read
        _poll(block=True)
    with _pending_lock:
        _flag_next = (_flag_next + 1) % 256
        return _flag_ring[_flag_next:_flag_next + 1]


def watch_host_flag(slot, device):
    """Register a pinned flag slot the launch just issued on the current stream will write (see ``check_labels``)."""
    if not _CHECK_LABELS:
        return
    if SYNC_LABEL_CHECK:
        torch.cuda.current_stream(device).synchronize()
        if int(slot[0]) != 0:
            raise RuntimeError(_LABEL_MSG)
        return
    _poll(block=False)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    with _pending_lock:
        _pending.append((slot, ev))


def region_workspace(device, C):
    """The persistent workspace of ``ptb_region_loss_fwd`` for the current stream of ``device``: zero when created, left zero by
    every launch (the last workgroup exchanges the slot sums, its counter and the label flag against zero)."""
    need = int(N.load().ptb_region_workspace_bytes(C))
    if need < 0:
        return None
    key = (device.index, N._raw_stream(device.index) if N._raw_stream is not None else torch.cuda.current_stream(device).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < need:
        if len(_workspaces) >= 64:            # (streams come and go: forget the oldest half; a workspace is recreated zeroed on demand)
            for k in list(_workspaces)[:32]:
                del _workspaces[k]
        ws = _workspaces[key] = torch.zeros(max(need, 1 << 16), dtype=torch.uint8, device=device)
    return ws


def new_sums(row, device, with_flag=True):
    """Uninitialised slot-sum workspace [SUM_SLOTS, row] (+ the int32 label flag): the forward entry points zero them on the launch
    stream themselves, so no framework fill kernel runs on the loss path."""
    sums = torch.empty((SUM_SLOTS, row), dtype=torch.float64, device=device)
    return sums, (torch.empty(1, dtype=torch.int32, device=device) if with_flag else None)


def finalize(sums, flag=None):
    """[row] float64 = the slots added up by ``ptb_sums_finalize``; NaN when the kernel raised the label flag (a label outside
    [0, C) that is not ignore_index can therefore never yield a finite loss, whatever the asynchronous host check does)."""
    out = torch.empty(sums.shape[1], dtype=torch.float64, device=sums.device)
    lib = N.load()
    with N.on_device(sums.device):
        rc = lib.ptb_sums_finalize(sums.data_ptr(), sums.shape[0], sums.shape[1], out.data_ptr(), _ptr(flag), N.stream_ptr(sums.device))
    N.check(rc, "ptb_sums_finalize")
    return out


class SigmoidFocalSums(torch.autograd.Function):
    """sums[0] = sum_i L_i, sums[1] = sum_i F_i (focal terms), optionally the unreduced L map.

    x: [B, C, HW] fp32 logits; labels int64 [B, HW] or dense fp32 [B, C, HW]."""

    @staticmethod
    def forward(ctx, x, labels, dense, class_weights, flags, gamma, alpha, threshold, ignore_label, ignore_value):
        B, C, HW = x.shape
        sums, flag = new_sums(2 + 3 * C, x.device)
        elem = torch.empty_like(x) if flags & SEG_ELEMWISE else None
        lib = N.load()
        with N.on_device(x.device):
            rc = lib.ptb_seg_loss_fwd(x.data_ptr(), _ptr(labels), _ptr(dense), _ptr(class_weights), sums.data_ptr(), _ptr(elem),
                                      flag.data_ptr(), B, C, HW, flags | SEG_FOCAL, PROB_SIGMOID, gamma, alpha, threshold,
                                      ignore_label, ignore_value, N.stream_ptr(x.device))
        N.bump()
        N.check(rc, "ptb_seg_loss_fwd")
        if labels is not None:
            check_labels(flag)
        ctx.save_for_backward(x, labels, dense, class_weights)
        ctx.cfg = (flags, gamma, alpha, threshold, ignore_label, ignore_value)
        if elem is None:
            elem = x.new_empty(0)
            ctx.has_elem = False
        else:
            ctx.has_elem = True
        return finalize(sums, flag if labels is not None else None)[:2], elem

    @staticmethod
    def backward(ctx, g_sums, g_elem):
        x, labels, dense, class_weights = ctx.saved_tensors
        flags, gamma, alpha, threshold, ignore_label, ignore_value = ctx.cfg
        B, C, HW = x.shape
        coef = g_sums.to(torch.float32).contiguous() if g_sums is not None else torch.zeros(2, device=x.device)
        grad_elem = None
        if ctx.has_elem and g_elem is not None:
            # grad = (g_sums[0] + g_elem) * dL + g_sums[1] * dF: fold the scalar into the map, multiplier 1
            grad_elem = (g_elem.to(torch.float32) + coef[0]).contiguous()
            coef = torch.stack([torch.ones((), device=x.device), coef[1]])
        grad = torch.empty_like(x)
        lib = N.load()
        with N.on_device(x.device):
            rc = lib.ptb_focal_bwd(x.data_ptr(), _ptr(labels), _ptr(dense), _ptr(class_weights), coef.data_ptr(), _ptr(grad_elem),
                                   grad.data_ptr(), B, C, HW, flags, gamma, alpha, threshold, ignore_label, ignore_value,
                                   N.stream_ptr(x.device))
        N.bump()
        N.check(rc, "ptb_focal_bwd")
        return grad, None, None, None, None, None, None, None, None, None


class _NotOneLaunch(Exception):
    """FocalScalar.forward: the C side does not serve this configuration in one launch and NOTHING was launched.  A private type: a
    genuine NotImplementedError out of torch / ctypes must not be mistaken for it."""


class FocalScalar(torch.autograd.Function):
    """``scale * sum_i L_i`` of the sigmoid focal loss on label maps as ONE launch (``ptb_region_loss_fwd`` with the focal sums alone:
    the streaming kernel's last workgroup adds up the slots and writes the scalar; no memset / finalize launches, no torch algebra
    on scalars).  Returns None when the configuration is not served that way (the caller takes ``SigmoidFocalSums``)."""

    @staticmethod
    def run(x, labels, flags, gamma, scale):
        if not ONE_LAUNCH_REGION_LOSS or not (x.requires_grad and torch.is_grad_enabled()):
            out = FocalScalar._launch(x, labels, flags, gamma, scale)
            return None if out is None else out[0]
        try:
            return FocalScalar.apply(x, labels, flags, gamma, scale)
        except _NotOneLaunch:             # (raised before anything was saved or launched)
            return None

    @staticmethod
    def _launch(x, labels, flags, gamma, scale):
        if not ONE_LAUNCH_REGION_LOSS:
            return None
        B, C, HW = x.shape
        ws = region_workspace(x.device, C)
        if ws is None:
            return None
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        coef = torch.empty(2 + 2 * C, dtype=torch.float32, device=x.device)
        slot = host_flag_slot() if _CHECK_LABELS else None
        with N.on_device(x.device):
            rc = N.load().ptb_region_loss_fwd(x.data_ptr(), labels.data_ptr(), None, None, ws.data_ptr(), B, C, HW, flags | SEG_FOCAL, PROB_SIGMOID,
                                              gamma, 0.0, 0.0, 0, 0.0, scale, 0.0, 0.0, 0.0, 1e-7, 0, None, C, loss.data_ptr(), coef.data_ptr(),
                                              _ptr(slot), N.stream_ptr(x.device))
        if rc == N.PTB_EUNSUPPORTED:
            return None
        N.check(rc, "ptb_region_loss_fwd (focal)")
        N.bump()
        if slot is not None:
            watch_host_flag(slot, x.device)
        return loss, coef

    @staticmethod
    def forward(ctx, x, labels, flags, gamma, scale):
        out = FocalScalar._launch(x, labels, flags, gamma, scale)
        if out is None:
            raise _NotOneLaunch
        ctx.save_for_backward(x, labels)
        ctx.cfg = (flags, gamma, scale)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        x, labels = ctx.saved_tensors
        flags, gamma, scale = ctx.cfg
        B, C, HW = x.shape
        coef = torch.stack([g.to(torch.float32) * scale, torch.zeros((), device=x.device)]).contiguous()
        grad = torch.empty_like(x)
        with N.on_device(x.device):
            rc = N.load().ptb_focal_bwd(x.data_ptr(), labels.data_ptr(), None, None, coef.data_ptr(), None, grad.data_ptr(), B, C, HW, flags, gamma,
                                        0.0, 0.0, 0, 0.0, N.stream_ptr(x.device))
        N.bump()
        N.check(rc, "ptb_focal_bwd")
        return grad, None, None, None, None


class SoftmaxActFocalSums(torch.autograd.Function):
    """``focal_loss_with_logits(activation="softmax")``: sums[0] = sum_i L_i, sums[1] = sum_i F_i, optionally the unreduced map.

    x: [B, C, HW] fp32 view whose C is the softmax dimension; labels int64 [B, HW] or dense fp32 [B, C, HW];
    cw = (class_weights | None, mode, n, div) -- see ptb_focal_softmax_fwd."""

    @staticmethod
    def forward(ctx, x, labels, dense, cw, flags, gamma, alpha, threshold, ignore_label, ignore_value):
        B, C, HW = x.shape
        weights, cw_mode, cw_n, cw_div = cw
        sums, flag = new_sums(2, x.device)
        elem = torch.empty_like(x) if flags & SEG_ELEMWISE else None
        lib = N.load()
        with N.on_device(x.device):
            rc = lib.ptb_focal_softmax_fwd(x.data_ptr(), _ptr(labels), _ptr(dense), _ptr(weights), cw_mode, cw_n, cw_div, sums.data_ptr(),
                                           _ptr(elem), flag.data_ptr(), B, C, HW, flags | SEG_FOCAL, gamma, alpha, threshold, ignore_label,
                                           ignore_value, N.stream_ptr(x.device))
        N.bump()
        N.check(rc, "ptb_focal_softmax_fwd")
        if labels is not None:
            check_labels(flag)
        ctx.save_for_backward(x, labels, dense, weights)
        ctx.cfg = (flags, gamma, alpha, threshold, ignore_label, ignore_value, cw_mode, cw_n, cw_div)
        ctx.has_elem = elem is not None
        return finalize(sums, flag if labels is not None else None), (elem if elem is not None else x.new_empty(0))

    @staticmethod
    def backward(ctx, g_sums, g_elem):
        x, labels, dense, weights = ctx.saved_tensors
        flags, gamma, alpha, threshold, ignore_label, ignore_value, cw_mode, cw_n, cw_div = ctx.cfg
        B, C, HW = x.shape
        coef = g_sums.to(torch.float32).contiguous() if g_sums is not None else torch.zeros(2, device=x.device)
        grad_elem = None
        if ctx.has_elem and g_elem is not None:
            grad_elem = (g_elem.to(torch.float32) + coef[0]).contiguous()   # (g_sums[0] + g_elem) * dL: multiplier 1
            coef = torch.stack([torch.ones((), device=x.device), coef[1]])
        grad = torch.empty_like(x)
        lib = N.load()
        with N.on_device(x.device):
            rc = lib.ptb_focal_softmax_bwd(x.data_ptr(), _ptr(labels), _ptr(dense), _ptr(weights), cw_mode, cw_n, cw_div, coef.data_ptr(),
                                           _ptr(grad_elem), grad.data_ptr(), B, C, HW, flags, gamma, alpha, threshold, ignore_label,
                                           ignore_value, N.stream_ptr(x.device))
        N.bump()
        N.check(rc, "ptb_focal_softmax_bwd")
        return grad, None, None, None, None, None, None, None, None, None


class RegionStats(torch.autograd.Function):
    """stats [3, C] float64: I_c = sum p t, P_c = sum p, T_c = sum t over batch and pixels (masked), p = activation(x)."""

    @staticmethod
    def forward(ctx, x, labels, dense, prob, has_ignore, ignore_label, ignore_value):
        B, C, HW = x.shape
        sums, flag = new_sums(2 + 3 * C, x.device)
        flags = SEG_STATS | (SEG_HAS_IGNORE if has_ignore else 0)
        lib = N.load()
        with N.on_device(x.device):
            rc = lib.ptb_seg_loss_fwd(x.data_ptr(), _ptr(labels), _ptr(dense), None, sums.data_ptr(), None, flag.data_ptr(), B, C, HW,
                                      flags, prob, 0.0, 0.0, 0.0, ignore_label, ignore_value, N.stream_ptr(x.device))
        N.bump()
        N.check(rc, "ptb_seg_loss_fwd")
        if labels is not None:
            check_labels(flag)
        ctx.save_for_backward(x, labels, dense)
        ctx.cfg = (flags, prob, ignore_label, ignore_value)
        return finalize(sums, flag if labels is not None else None)[2:].view(3, C)

    @staticmethod
    def backward(ctx, g):
        x, labels, dense = ctx.saved_tensors
        flags, prob, ignore_label, ignore_value = ctx.cfg
        B, C, HW = x.shape
        g = g.to(torch.float32).contiguous()
        grad = torch.empty_like(x)
        lib = N.load()
        with N.on_device(x.device):
            rc = lib.ptb_seg_stats_bwd(x.data_ptr(), _ptr(labels), _ptr(dense), g[0].data_ptr(), g[1].data_ptr(), grad.data_ptr(),
                                       B, C, HW, flags, prob, ignore_label, ignore_value, N.stream_ptr(x.device))
        N.bump()
        N.check(rc, "ptb_seg_stats_bwd")
        return grad, None, None, None, None, None, None


class SoftmaxFocalSums(torch.autograd.Function):
    """sums[0] = sum of per-pixel softmax-focal losses, sums[1] = sum of all focal terms; optional [B, HW] map."""

    @staticmethod
    def forward(ctx, x, labels, class_weights, reduced, gamma, threshold, ignore_label, want_map):
        B, C, HW = x.shape
        sums, flag = new_sums(2, x.device)
        pix = torch.empty((B, HW), dtype=torch.float32, device=x.device) if want_map else None
        lib = N.load()
        with N.on_device(x.device):
            rc = lib.ptb_softmax_focal_fwd(x.data_ptr(), labels.data_ptr(), _ptr(class_weights), sums.data_ptr(), _ptr(pix),
                                           flag.data_ptr(), B, C, HW, reduced, gamma, threshold, ignore_label, N.stream_ptr(x.device))
        N.bump()
        N.check(rc, "ptb_softmax_focal_fwd")
        check_labels(flag)
        ctx.save_for_backward(x, labels, class_weights)
        ctx.cfg = (reduced, gamma, threshold, ignore_label)
        ctx.has_map = want_map
        return finalize(sums, flag), (pix if want_map else x.new_empty(0))

    @staticmethod
    def backward(ctx, g_sums, g_pix):
        x, labels, class_weights = ctx.saved_tensors
        reduced, gamma, threshold, ignore_label = ctx.cfg
        B, C, HW = x.shape
        coef = g_sums.to(torch.float32).contiguous()
        grad_pix = None
        if ctx.has_map and g_pix is not None:
            grad_pix = (g_pix.to(torch.float32) + coef[0]).contiguous()
            coef = torch.stack([torch.ones((), device=x.device), coef[1]])
        grad = torch.empty_like(x)
        lib = N.load()
        with N.on_device(x.device):
            rc = lib.ptb_softmax_focal_bwd(x.data_ptr(), labels.data_ptr(), _ptr(class_weights), coef.data_ptr(), _ptr(grad_pix),
                                           grad.data_ptr(), B, C, HW, reduced, gamma, threshold, ignore_label, N.stream_ptr(x.device))
        N.bump()
        N.check(rc, "ptb_softmax_focal_bwd")
        return grad, None, None, None, None, None, None, None


def as_bchw(t):
    """View an [N, C, *] (or lower-rank) tensor as [B, C, HW]."""
    if t.dim() >= 2:
        return t.reshape(t.shape[0], t.shape[1], -1)
    return t.reshape(1, 1, -1)


class FusedSegSums(torch.autograd.Function):
    """One forward pass over logits + labels for BOTH the sigmoid focal sums [2] and the region statistics [3, C]
    (BASELINE configs[3]: BinaryFocal + SoftDice + SoftJaccard fused).  Backward = focal and region gradient kernels."""

    @staticmethod
    def forward(ctx, x, labels, dense, class_weights, flags, prob, gamma, alpha, threshold, ignore_label, ignore_value):
        B, C, HW = x.shape
        sums, flag = new_sums(2 + 3 * C, x.device)
        lib = N.load()
        with N.on_device(x.device):
            rc = lib.ptb_seg_loss_fwd(x.data_ptr(), _ptr(labels), _ptr(dense), _ptr(class_weights), sums.data_ptr(), None,
                                      flag.data_ptr(), B, C, HW, flags | SEG_FOCAL | SEG_STATS, prob, gamma, alpha, threshold,
                                      ignore_label, ignore_value, N.stream_ptr(x.device))
        N.bump()
        N.check(rc, "ptb_seg_loss_fwd")
        if labels is not None:
            check_labels(flag)
        ctx.save_for_backward(x, labels, dense, class_weights)
        ctx.cfg = (flags, prob, gamma, alpha, threshold, ignore_label, ignore_value)
        total = finalize(sums, flag if labels is not None else None)
        return total[:2], total[2:].view(3, C)

    @staticmethod
    def backward(ctx, g_focal, g_stats):
        x, labels, dense, class_weights = ctx.saved_tensors
        flags, prob, gamma, alpha, threshold, ignore_label, ignore_value = ctx.cfg
        B, C, HW = x.shape
        lib = N.load()
        grad = torch.empty_like(x)
        coef = g_focal.to(torch.float32).contiguous()
        gs = g_stats.to(torch.float32).contiguous()
        with N.on_device(x.device):
            rc = lib.ptb_seg_fused_bwd(x.data_ptr(), _ptr(labels), _ptr(dense), _ptr(class_weights), coef.data_ptr(), gs[0].data_ptr(),
                                       gs[1].data_ptr(), grad.data_ptr(), B, C, HW, flags, prob, gamma, alpha, threshold, ignore_label,
                                       ignore_value, N.stream_ptr(x.device))
        if rc == 0:
            N.bump()
            return grad, None, None, None, None, None, None, None, None, None, None
        if rc != -2:
            N.check(rc, "ptb_seg_fused_bwd")
        grad2 = torch.empty_like(x)
        with N.on_device(x.device):
            rc = lib.ptb_focal_bwd(x.data_ptr(), _ptr(labels), _ptr(dense), _ptr(class_weights), coef.data_ptr(), None, grad.data_ptr(),
                                   B, C, HW, flags, gamma, alpha, threshold, ignore_label, ignore_value, N.stream_ptr(x.device))
            N.check(rc, "ptb_focal_bwd")
            sflags = flags & SEG_HAS_IGNORE
            rc = lib.ptb_seg_stats_bwd(x.data_ptr(), _ptr(labels), _ptr(dense), gs[0].data_ptr(), gs[1].data_ptr(), grad2.data_ptr(),
                                       B, C, HW, SEG_STATS | sflags, prob, ignore_label, ignore_value, N.stream_ptr(x.device))
            N.check(rc, "ptb_seg_stats_bwd")
        N.bump()
        return grad.add_(grad2), None, None, None, None, None, None, None, None, None, None


class RegionLoss(torch.autograd.Function):
    """Dice / Jaccard / (mean sigmoid focal + Dice + Jaccard) as ONE differentiable scalar: the streaming kernel that
    produces the sums, then ``ptb_region_epilogue`` for the [C]-sized tail and its derivative (two launches forward, one
    multiply + one streaming kernel backward -- as torch ops the tail alone is ~65 launches per forward / backward pair)."""

    @staticmethod
    def forward(ctx, x, labels, dense, class_weights, flags, prob, gamma, alpha, threshold, ignore_label, ignore_value,
                with_focal, focal_scale, dice_weight, jaccard_weight, smooth, eps, log_loss, class_mask, n_selected):
        B, C, HW = x.shape
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        coef = torch.empty(2 + 2 * C, dtype=torch.float32, device=x.device)
        what = SEG_STATS | (SEG_FOCAL if with_focal else 0)
        lib = N.load()
        ctx.save_for_backward(x, labels, dense, class_weights, coef)
        ctx.cfg = (flags, prob, gamma, alpha, threshold, ignore_label, ignore_value, with_focal)
        ws = region_workspace(x.device, C) if ONE_LAUNCH_REGION_LOSS else None
        if ws is not None:
            # one launch: the streaming kernel's last workgroup adds up the slots, evaluates the tail and its derivative and leaves
            # the workspace zeroed (no memset / finalize / epilogue launches, no device-to-host copy of the label flag)
            slot = host_flag_slot() if (labels is not None and _CHECK_LABELS) else None
            with N.on_device(x.device):
                rc = lib.ptb_region_loss_fwd(x.data_ptr(), _ptr(labels), _ptr(dense), _ptr(class_weights), ws.data_ptr(), B, C, HW, flags | what,
                                             prob, gamma, alpha, threshold, ignore_label, ignore_value, focal_scale if with_focal else 0.0,
                                             dice_weight, jaccard_weight, smooth, eps, 1 if log_loss else 0, _ptr(class_mask), n_selected,
                                             loss.data_ptr(), coef.data_ptr(), _ptr(slot), N.stream_ptr(x.device))
            if rc != N.PTB_EUNSUPPORTED:
                N.check(rc, "ptb_region_loss_fwd")
                N.bump()
                if slot is not None:
                    watch_host_flag(slot, x.device)
                return loss
        sums, flag = new_sums(2 + 3 * C, x.device)
        with N.on_device(x.device):
            rc = lib.ptb_seg_loss_fwd(x.data_ptr(), _ptr(labels), _ptr(dense), _ptr(class_weights), sums.data_ptr(), None,
                                      flag.data_ptr(), B, C, HW, flags | what, prob, gamma, alpha, threshold, ignore_label, ignore_value,
                                      N.stream_ptr(x.device))
            N.check(rc, "ptb_seg_loss_fwd")
            rc = lib.ptb_region_epilogue(sums.data_ptr(), SUM_SLOTS, C, focal_scale if with_focal else 0.0, dice_weight, jaccard_weight,
                                         smooth, eps, 1 if log_loss else 0, _ptr(class_mask), n_selected, loss.data_ptr(),
                                         coef.data_ptr(), flag.data_ptr() if labels is not None else None, N.stream_ptr(x.device))
            N.check(rc, "ptb_region_epilogue")
        N.bump()
        if labels is not None:
            check_labels(flag)
        return loss

    @staticmethod
    def backward(ctx, g):
        x, labels, dense, class_weights, coef = ctx.saved_tensors
        flags, prob, gamma, alpha, threshold, ignore_label, ignore_value, with_focal = ctx.cfg
        B, C, HW = x.shape
        k = coef * g.to(torch.float32)          # upstream gradient folded into the coefficient arrays (device side)
        kf, gi, gp = k[:2], k[2:2 + C], k[2 + C:]
        grad = torch.empty_like(x)
        lib = N.load()
        none = (None,) * 19
        with N.on_device(x.device):
            if with_focal:
                rc = lib.ptb_seg_fused_bwd(x.data_ptr(), _ptr(labels), _ptr(dense), _ptr(class_weights), kf.data_ptr(), gi.data_ptr(),
                                           gp.data_ptr(), grad.data_ptr(), B, C, HW, flags, prob, gamma, alpha, threshold, ignore_label,
                                           ignore_value, N.stream_ptr(x.device))
                if rc == -2:   # the fused kernel does not apply (C > 16, unaligned, ...): two kernels and an add
                    grad2 = torch.empty_like(x)
                    rc = lib.ptb_focal_bwd(x.data_ptr(), _ptr(labels), _ptr(dense), _ptr(class_weights), kf.data_ptr(), None, grad.data_ptr(),
                                           B, C, HW, flags, gamma, alpha, threshold, ignore_label, ignore_value, N.stream_ptr(x.device))
                    N.check(rc, "ptb_focal_bwd")
                    rc = lib.ptb_seg_stats_bwd(x.data_ptr(), _ptr(labels), _ptr(dense), gi.data_ptr(), gp.data_ptr(), grad2.data_ptr(),
                                               B, C, HW, SEG_STATS | (flags & SEG_HAS_IGNORE), prob, ignore_label, ignore_value,
                                               N.stream_ptr(x.device))
                    N.check(rc, "ptb_seg_stats_bwd")
                    grad.add_(grad2)
                else:
                    N.check(rc, "ptb_seg_fused_bwd")
            else:
                rc = lib.ptb_seg_stats_bwd(x.data_ptr(), _ptr(labels), _ptr(dense), gi.data_ptr(), gp.data_ptr(), grad.data_ptr(),
                                           B, C, HW, SEG_STATS | (flags & SEG_HAS_IGNORE), prob, ignore_label, ignore_value,
                                           N.stream_ptr(x.device))
                N.check(rc, "ptb_seg_stats_bwd")
        N.bump()
        return (grad,) + none
