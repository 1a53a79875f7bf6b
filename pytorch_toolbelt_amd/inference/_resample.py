"""Resizing on the GPU for multiscale TTA: HIP kernels ptb_resize_bilinear / ptb_resize_bicubic / ptb_resize_nearest / ptb_resize_nearest_exact /
ptb_resize_area (+ their adjoints),
ptb_ms_deaug_reduce (+ adjoint) and the fused flips + multiscale pass ptb_ms_flip_deaug_reduce.

The reference calls ``torch.nn.functional.interpolate`` (inference/tta.py:599-621, 645-689) and gets gradients from autograd;
here every piece is an autograd Function over a hand-written kernel, evaluated in float32 (half / bfloat16 tensors are
converted and the result cast back, like the other TTA functions)."""
import ctypes

import torch

from .. import _native as N
from . import _host

_MODES = ("bilinear", "nearest", "bicubic", "nearest-exact", "area")      # every mode F.interpolate takes for a [B, C, H, W] tensor


def _check_mode(mode, align_corners):
    if mode not in _MODES:
        # (F.interpolate's own answer for a 4-D input: "linear" / "trilinear" want 3-D / 5-D tensors, anything else is unknown)
        raise NotImplementedError(f"Got 4D input, but mode={mode!r} is not one of F.interpolate's 4-D modes {_MODES}")
    if mode in ("nearest", "nearest-exact", "area") and align_corners is not None:
        # F.interpolate's own rule (the reference forwards both arguments, tta.py:613-615 / 683-685)
        raise ValueError("align_corners option can only be set with the interpolating modes: linear | bilinear | bicubic | trilinear")


def _f32(x, what="multiscale TTA"):
    N.require_device(x, what)
    if x.dim() != 4:
        raise NotImplementedError(f"{what}: expected a [B, C, H, W] tensor")
    return (x if x.dtype == torch.float32 else x.float()).contiguous()


class _Resize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, size, mode, align_corners):
        B, C, H, W = x.shape
        ho, wo = int(size[0]), int(size[1])
        out = torch.empty((B, C, ho, wo), device=x.device, dtype=torch.float32)
        lib = N.load()
        with N.on_device(x.device):
            if mode == "bilinear":
                rc = lib.ptb_resize_bilinear(x.data_ptr(), out.data_ptr(), B * C, H, W, ho, wo, 1 if align_corners else 0, N.stream_ptr(x.device))
            elif mode == "bicubic":
                rc = lib.ptb_resize_bicubic(x.data_ptr(), out.data_ptr(), B * C, H, W, ho, wo, 1 if align_corners else 0, 0, N.stream_ptr(x.device))
            elif mode == "nearest-exact":
                rc = lib.ptb_resize_nearest_exact(x.data_ptr(), out.data_ptr(), B * C, H, W, ho, wo, 0, N.stream_ptr(x.device))
            elif mode == "area":
                rc = lib.ptb_resize_area(x.data_ptr(), out.data_ptr(), B * C, H, W, ho, wo, 0, N.stream_ptr(x.device))
            else:
                rc = lib.ptb_resize_nearest(x.data_ptr(), out.data_ptr(), B * C, H, W, ho, wo, 0, N.stream_ptr(x.device))
        N.bump()
        N.check(rc, "ptb_resize")
        ctx.cfg = (B, C, H, W, ho, wo, mode, bool(align_corners))
        return out

    @staticmethod
    def backward(ctx, g):
        B, C, H, W, ho, wo, mode, ac = ctx.cfg
        g = g.to(torch.float32).contiguous()
        gin = torch.zeros((B, C, H, W), device=g.device, dtype=torch.float32)
        lib = N.load()
        with N.on_device(g.device):
            if mode == "bilinear":
                rc = lib.ptb_resize_bilinear_bwd(g.data_ptr(), gin.data_ptr(), B * C, H, W, ho, wo, 1 if ac else 0, N.stream_ptr(g.device))
            elif mode == "bicubic":
                rc = lib.ptb_resize_bicubic(g.data_ptr(), gin.data_ptr(), B * C, H, W, ho, wo, 1 if ac else 0, 1, N.stream_ptr(g.device))
            elif mode == "nearest-exact":
                rc = lib.ptb_resize_nearest_exact(g.data_ptr(), gin.data_ptr(), B * C, H, W, ho, wo, 1, N.stream_ptr(g.device))
            elif mode == "area":
                rc = lib.ptb_resize_area(g.data_ptr(), gin.data_ptr(), B * C, H, W, ho, wo, 1, N.stream_ptr(g.device))
            else:
                rc = lib.ptb_resize_nearest(g.data_ptr(), gin.data_ptr(), B * C, H, W, ho, wo, 1, N.stream_ptr(g.device))
        N.bump()
        N.check(rc, "ptb_resize (backward)")
        return gin, None, None, None


def resize(x: torch.Tensor, size, mode: str, align_corners) -> torch.Tensor:
    """``F.interpolate(x, size=size, mode=mode, align_corners=align_corners)`` for a [B, C, H, W] GPU tensor
    (mode "bilinear" | "bicubic" | "nearest" | "nearest-exact" | "area"), differentiable.  Host tensors go to ``F.interpolate`` itself (any mode)."""
    if not x.is_cuda:
        return _host.resize(x, size, mode, align_corners)
    _check_mode(mode, align_corners)
    y = _Resize.apply(_f32(x), (int(size[0]), int(size[1])), mode, align_corners)
    return y if x.dtype == torch.float32 else y.to(x.dtype)


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


class _MsReduce(torch.autograd.Function):
    """out = reduction_s(bilinear_s(maps[s])) in one launch; backward = one launch that adds every output pixel's gradient to the
    (<= 4) taps of every scale."""

    @staticmethod
    def forward(ctx, size, align_corners, code, *maps):
        first = maps[0]
        B, C = first.shape[0], first.shape[1]
        ho, wo = size
        out = torch.empty((B, C, ho, wo), device=first.device, dtype=torch.float32)
        hs = N.int_array([int(m.shape[2]) for m in maps])
        ws = N.int_array([int(m.shape[3]) for m in maps])
        lib = N.load()
        with N.on_device(first.device):
            # the 64 x 64-tile kernel of the fused flips + multiscale pass with a single (identity) view: 1.08x source traffic
            # instead of the 1.21x of the 64 x 16-tile kernel; shapes it does not take (odd widths, ratios above ~1.3) fall through
            rc = lib.ptb_ms_flip_deaug_reduce(_ptr_array(maps), hs, ws, len(maps), 1, N.int_array([N.IDENT]), N.RED_SUM, out.data_ptr(), B * C,
                                              ho, wo, 1 if align_corners else 0, code, N.stream_ptr(first.device))
            if rc == N.PTB_EUNSUPPORTED:
                rc = lib.ptb_ms_deaug_reduce(_ptr_array(maps), hs, ws, len(maps), out.data_ptr(), B * C, ho, wo, 1 if align_corners else 0, code,
                                             N.stream_ptr(first.device))
        N.bump()
        N.check(rc, "ptb_ms_deaug_reduce")
        ctx.save_for_backward(out, *maps)
        ctx.cfg = (B * C, ho, wo, bool(align_corners), code)
        return out

    @staticmethod
    def backward(ctx, g):
        out, *maps = ctx.saved_tensors
        planes, ho, wo, ac, code = ctx.cfg
        g = g.to(torch.float32).contiguous()
        grads = [torch.zeros_like(m) for m in maps]
        hs = N.int_array([int(m.shape[2]) for m in maps])
        ws = N.int_array([int(m.shape[3]) for m in maps])
        lib = N.load()
        with N.on_device(g.device):
            rc = lib.ptb_ms_deaug_reduce_bwd(_ptr_array(maps), hs, ws, len(maps), out.data_ptr(), g.data_ptr(), _ptr_array(grads), planes, ho, wo,
                                             1 if ac else 0, code, N.stream_ptr(g.device))
        N.bump()
        N.check(rc, "ptb_ms_deaug_reduce_bwd")
        return (None, None, None) + tuple(grads)


def ms_reduce(maps, size, align_corners: bool, code: int) -> torch.Tensor:
    """Fused multiscale de-augmentation: ``reduce_s F.interpolate(maps[s], size, 'bilinear', align_corners)`` in one HIP
    pass (maps already of the target size are read as they are).  maps: [B, C, h_s, w_s] GPU tensors, <= 8.  Differentiable."""
    first = maps[0]
    if not first.is_cuda:
        return _host.ms_reduce(maps, size, align_corners, code)
    B, C = first.shape[0], first.shape[1]
    ms = []
    for m in maps:
        m32 = _f32(m)
        if m32.shape[0] != B or m32.shape[1] != C:
            raise RuntimeError("multiscale TTA: every scale must have the same batch size and channel count")
        ms.append(m32)
    out = _MsReduce.apply((int(size[0]), int(size[1])), bool(align_corners), code, *ms)
    return out if first.dtype == torch.float32 else out.to(first.dtype)


def ms_flip_reduce(maps, views, size, align_corners: bool, inner_code: int, code: int):
    """One pass over every flip view of every scale (``ptb_ms_flip_deaug_reduce``).  maps: the model outputs of the flip-augmented
    batches, [V*B, C, h_s, w_s] chunk-major.  Returns None when the fused kernel does not take the configuration (the caller
    composes ``<group>_image_deaugment`` + ``ms_image_deaugment``).  Inference only (no autograd)."""
    first = maps[0]
    V = len(views)
    if not first.is_cuda or any(v & 1 for v in views) or not 1 <= len(maps) <= 8 or first.size(0) % V:
        return None           # (host tensors: the caller composes <group>_image_deaugment + ms_image_deaugment)
    B, C = first.shape[0] // V, first.shape[1]
    ms = []
    for m in maps:
        m32 = _f32(m)
        if m32.shape[0] != V * B or m32.shape[1] != C:
            raise RuntimeError("multiscale TTA: every scale must have the same batch size and channel count")
        ms.append(m32)
    ho, wo = int(size[0]), int(size[1])
    out = torch.empty((B, C, ho, wo), device=first.device, dtype=torch.float32)
    hs = N.int_array([int(m.shape[2]) for m in ms])
    ws = N.int_array([int(m.shape[3]) for m in ms])
    lib = N.load()
    with N.on_device(first.device):
        rc = lib.ptb_ms_flip_deaug_reduce(_ptr_array(ms), hs, ws, len(ms), V, N.int_array(list(views)), inner_code, out.data_ptr(), B * C, ho, wo,
                                          1 if align_corners else 0, code, N.stream_ptr(first.device))
    N.bump()
    if rc == N.PTB_EUNSUPPORTED:
        return None
    N.check(rc, "ptb_ms_flip_deaug_reduce")
    return out if first.dtype == torch.float32 else out.to(first.dtype)
