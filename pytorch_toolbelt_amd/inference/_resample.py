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
