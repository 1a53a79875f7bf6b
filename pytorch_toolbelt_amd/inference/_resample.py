"""Bilinear resize on the GPU for multiscale TTA (HIP kernel ptb_resize_bilinear)."""
import torch

from .. import _native as N


def resize(x: torch.Tensor, size, mode: str, align_corners: bool) -> torch.Tensor:
    """``F.interpolate(x, size=size, mode="bilinear", align_corners=align_corners)`` for a float32 [B,C,H,W] GPU tensor."""
    if mode != "bilinear":
        raise NotImplementedError(f"multiscale TTA: only mode='bilinear' has a native kernel (got {mode!r})")
    N.require_device(x, "multiscale TTA")
    if x.dim() != 4 or x.dtype != torch.float32:
        raise NotImplementedError("multiscale TTA: expected a float32 [B, C, H, W] tensor")
    if x.requires_grad and torch.is_grad_enabled():
        raise NotImplementedError("multiscale TTA: backward of the native bilinear resize is not implemented")
    x = x.contiguous()
    B, C, H, W = x.shape
    ho, wo = int(size[0]), int(size[1])
    out = torch.empty((B, C, ho, wo), device=x.device, dtype=x.dtype)
    lib = N.load()
    with N.on_device(x.device):
        rc = lib.ptb_resize_bilinear(x.data_ptr(), out.data_ptr(), B * C, H, W, ho, wo, 1 if align_corners else 0, N.stream_ptr(x.device))
    N.bump()
    N.check(rc, "ptb_resize_bilinear")
    return out


def ms_reduce(maps, size, align_corners: bool, code: int) -> torch.Tensor:
    """Fused multiscale de-augmentation: ``reduce_s F.interpolate(maps[s], size, 'bilinear', align_corners)`` in one HIP
    pass (maps already of the target size are read as they are).  maps: float32 [B, C, h_s, w_s] GPU tensors, <= 8."""
    import ctypes

    first = maps[0]
    N.require_device(first, "multiscale TTA")
    B, C = first.shape[0], first.shape[1]
    ms = []
    for m in maps:
        N.require_device(m, "multiscale TTA")
        if m.dim() != 4 or m.dtype != torch.float32 or m.shape[0] != B or m.shape[1] != C:
            raise NotImplementedError("multiscale TTA: expected float32 [B, C, H, W] tensors with equal B and C")
        if m.requires_grad and torch.is_grad_enabled():
            raise NotImplementedError("multiscale TTA: backward of the native bilinear resize is not implemented")
        ms.append(m.contiguous())
    ho, wo = int(size[0]), int(size[1])
    out = torch.empty((B, C, ho, wo), device=first.device, dtype=torch.float32)
    ptrs = (ctypes.c_void_p * len(ms))(*[m.data_ptr() for m in ms])
    hs = N.int_array([int(m.shape[2]) for m in ms])
    ws = N.int_array([int(m.shape[3]) for m in ms])
    lib = N.load()
    with N.on_device(first.device):
        rc = lib.ptb_ms_deaug_reduce(ptrs, hs, ws, len(ms), out.data_ptr(), B * C, ho, wo, 1 if align_corners else 0, code,
                                     N.stream_ptr(first.device))
    N.bump()
    N.check(rc, "ptb_ms_deaug_reduce")
    return out
