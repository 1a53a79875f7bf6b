"""Autograd-aware Python front of the view kernels (ptb_view_transform / ptb_deaug_reduce).

A *view code* is 3 bits (transpose, flip source rows, flip source cols) -- see include/ptb_hip.h.  CUDA tensors launch the
hand-written HIP kernels on the current stream (evaluated in float32); host tensors are handed to ``_host`` at the three public
entry points (``view_transform``, ``deaug_reduce``, ``stack_reduce``) and never touch the binding.
"""
from typing import Sequence

import torch

from .. import _native as N
from . import _host

# inverse of each D4 element: the two quarter turns swap, everything else is an involution
INVERSE = {
    N.IDENT: N.IDENT,
    N.TRANSPOSE: N.TRANSPOSE,
    N.FLIPUD: N.FLIPUD,
    N.FLIPLR: N.FLIPLR,
    N.ROT180: N.ROT180,
    N.ANTITRANSPOSE: N.ANTITRANSPOSE,
    N.ROT90_CW: N.ROT90_CCW,
    N.ROT90_CCW: N.ROT90_CW,
}

REDUCTION_CODES = {
    "sum": N.RED_SUM,
    "mean": N.RED_MEAN,
    "gmean": N.RED_GMEAN,
    "geometric_mean": N.RED_GMEAN,
    "hmean": N.RED_HMEAN,
    "harmonic_mean": N.RED_HMEAN,
    "harmonic1p": N.RED_HARMONIC1P,
    "logodd": N.RED_LOGODD,
    "log1p": N.RED_LOG1P,
}
REDUCTION_NAMES = {}
for _name, _code in REDUCTION_CODES.items():
    REDUCTION_NAMES.setdefault(_code, _name)


_LOW_PRECISION = (torch.float16, torch.bfloat16)
_EXACT_IN_F32 = (torch.uint8, torch.int8, torch.int16, torch.bool)


def _check_image(x, what, ints=False):
    """Validate and return (float32 tensor, dtype to cast the result back to or None).  Half / bfloat16 inputs are
    evaluated in float32 by the kernels and the result is cast back (at least as accurate as the reference's
    half-precision op chain)."""
    N.require_device(x, what)
    if x.dim() != 4:
        raise NotImplementedError(f"{what}: expected a [B, C, H, W] tensor, got shape {tuple(x.shape)}")
    if x.dtype in _LOW_PRECISION or (ints and x.dtype in _EXACT_IN_F32):
        return x.float(), x.dtype        # (uint8 / int8 / int16 / bool images: every value is exact in float32, so are the views of them)
    if x.dtype == torch.float64:
        return x.float(), x.dtype        # evaluated in float32 like everything else here (documented deviation)
    if x.dtype != torch.float32:
        raise NotImplementedError(f"{what}: the native path is float32 (half / bfloat16 / float64 and 8 / 16-bit integers are converted), got {x.dtype}")
    return x, None


def _raw_view_transform(x, views: Sequence[int], in_is_batch: bool, scale: float):
    """out[k*B + b] = scale * view_k(x[b or k*B+b]); x contiguous fp32 [*, C, H, W] on the GPU."""
    V = len(views)
    n, C, H, W = x.shape
    B = n if in_is_batch else n // V
    if any(v & 1 for v in views) and H != W:
        raise ValueError(f"Input tensor must have number of rows equal to number of cols. Got input tensor of shape {x.size()}")
    out = torch.empty((V * B, C, H, W), device=x.device, dtype=x.dtype)
    lib = N.load()
    with N.on_device(x.device):
        rc = lib.ptb_view_transform(x.data_ptr(), out.data_ptr(), V, N.int_array(views), 1 if in_is_batch else 0, float(scale),
                                    B, C, H, W, N.stream_ptr(x.device))
    N.bump()
    N.check(rc, "ptb_view_transform")
    return out


def _raw_deaug_reduce(x, views: Sequence[int], code: int):
    """out[b] = reduce_k view_k(x[k*B + b]); x contiguous [V*B, C, H, W] on the GPU, fp32 -- or fp16 / bf16, which the
    kernel widens in registers (the fp32 copy the reference's `.float()` would write never exists).  out is fp32."""
    V = len(views)
    n, C, H, W = x.shape
    B = n // V
    if any(v & 1 for v in views) and H != W:
        raise ValueError(f"Transposing views need square inputs, got {tuple(x.shape)}")
    out = torch.empty((B, C, H, W), device=x.device, dtype=torch.float32)
    lib = N.load()
    dcode = N.DTYPE_CODES[x.dtype]
    with N.on_device(x.device):
        rc = lib.ptb_deaug_reduce_t(x.data_ptr(), dcode, out.data_ptr(), V, N.int_array(views), code, B, C, H, W, N.stream_ptr(x.device))
    N.bump()
    if rc == -2 and dcode != N.F32:   # shape needs the scalar kernels
        return _raw_deaug_reduce(x.float(), views, code)
    N.check(rc, "ptb_deaug_reduce")
    return out


class _ViewTransform(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, views, in_is_batch, scale):
        ctx.views, ctx.in_is_batch, ctx.scale = tuple(views), in_is_batch, scale
        return _raw_view_transform(x.contiguous(), views, in_is_batch, scale)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        inv = [INVERSE[v] for v in ctx.views]
        if ctx.in_is_batch:  # every input element fans out to V outputs: gather them back and sum
            gx = _raw_deaug_reduce(g, inv, N.RED_SUM)
            if ctx.scale != 1.0:
                gx = gx * ctx.scale
        else:
            gx = _raw_view_transform(g, inv, False, ctx.scale)
        return gx, None, None, None


def _raw_deaug_reduce_bwd(x, out, g, views, code):
    """grad wrt x of the non-linear reductions: view_k^-1(g * post'(m) / V) * pre'(x_k), one HIP scatter launch."""
    V = len(views)
    n, C, H, W = x.shape
    grad = torch.empty_like(x)
    lib = N.load()
    with N.on_device(x.device):
        rc = lib.ptb_deaug_reduce_bwd(x.data_ptr(), out.data_ptr(), g.data_ptr(), grad.data_ptr(), V, N.int_array(views), code,
                                      n // V, C, H, W, N.stream_ptr(x.device))
    N.bump()
    N.check(rc, "ptb_deaug_reduce_bwd")
    return grad


class _DeaugReduce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, views, code):
        ctx.views, ctx.code = tuple(views), code
        x = x.contiguous()
        out = _raw_deaug_reduce(x, views, code)
        if code not in (N.RED_SUM, N.RED_MEAN):
            ctx.save_for_backward(x, out)
        return out

    @staticmethod
    def backward(ctx, g):
        if ctx.code not in (N.RED_SUM, N.RED_MEAN):
            x, out = ctx.saved_tensors
            return _raw_deaug_reduce_bwd(x, out, g.contiguous(), list(ctx.views), ctx.code), None, None
        inv = [INVERSE[v] for v in ctx.views]
        scale = 1.0 if ctx.code == N.RED_SUM else 1.0 / len(inv)
        return _raw_view_transform(g.contiguous(), inv, True, scale), None, None


def _raw_permute(x, views: Sequence[int], in_is_batch: bool):
    """``ptb_view_permute``: the views as pure data movement -- any dtype, any rank >= 4 (dims beyond the fourth ride along with their
    pixel), non-square planes when the views agree on the output shape.  ``x`` contiguous on the GPU; the result has its dtype and
    exactly its bits."""
    V = len(views)
    n, C, H, W = (int(v) for v in x.shape[:4])
    rest = tuple(int(v) for v in x.shape[4:])
    B = n if in_is_batch else n // V
    n_t = sum(v & 1 for v in views)
    if n_t and n_t != V and H != W:
        raise ValueError(f"Input tensor must have number of rows equal to number of cols. Got input tensor of shape {x.size()}")
    Ho, Wo = (W, H) if n_t else (H, W)
    out = torch.empty((V * B, C, Ho, Wo) + rest, device=x.device, dtype=x.dtype)
    if out.numel() == 0:
        return out
    payload = x.element_size()
    for r in rest:
        payload *= r
    align = x.data_ptr() | out.data_ptr()
    elem = next(e for e in (16, 8, 4, 2, 1) if payload % e == 0 and align % e == 0)
    lib = N.load()
    with N.on_device(x.device):
        rc = lib.ptb_view_permute(x.data_ptr(), out.data_ptr(), V, N.int_array(views), 1 if in_is_batch else 0, B * C, H, W, elem,
                                  payload // elem, N.stream_ptr(x.device))
    N.bump()
    N.check(rc, "ptb_view_permute")
    return out


class _Permute(torch.autograd.Function):
    """The permutation path with a gradient (float64 / half / complex tensors, tensors of more than four dims): the backward moves
    the incoming gradient back with the inverse views -- and sums the V copies of an augmented batch, in the gradient's own dtype."""

    @staticmethod
    def forward(ctx, x, views, in_is_batch):
        ctx.views, ctx.in_is_batch = tuple(views), in_is_batch
        return _raw_permute(x.contiguous(), views, in_is_batch)

    @staticmethod
    def backward(ctx, g):
        inv = [INVERSE[v] for v in ctx.views]
        back = _raw_permute(g.contiguous(), inv, False)
        if ctx.in_is_batch and len(inv) > 1:
            back = back.view(len(inv), back.shape[0] // len(inv), *back.shape[1:]).sum(dim=0)
        return back, None, None


def view_transform(x, views, in_is_batch=True, scale=1.0):
    views = list(views)
    if not x.is_cuda:      # a host tensor: the device-agnostic torch path (inference/_host.py), like the reference
        return _host.view_transform(x, views, in_is_batch, scale)
    N.require_device(x, "view transform")
    if x.dim() < 4:        # dims 2 and 3 do not exist: raise what x.flip(3) / x.rot90(dims=(2, 3)) / x.transpose(2, 3) raise (functional.py:47-132)
        return _host.view_transform(x, views, in_is_batch, scale)
    if not in_is_batch and x.shape[0] % len(views) != 0:
        raise RuntimeError(f"Input batch size ({x.size(0)}) must be divisible by {len(views)}.")
    square = x.shape[2] == x.shape[3] or not any(v & 1 for v in views)
    if x.dim() == 4 and x.dtype == torch.float32 and square:
        return _ViewTransform.apply(x, views, in_is_batch, scale)          # the fp32 kernels of the hot path
    # every other dtype / rank / a non-square plane under transposing views only: an index permutation is index work -- the elements
    # move as opaque 1 / 2 / 4 / 8 / 16-byte words and come out with the bits they went in with (float64 stays float64)
    if x.requires_grad and torch.is_grad_enabled() and (x.is_floating_point() or x.is_complex()):
        out = _Permute.apply(x, views, in_is_batch)
    else:
        out = _raw_permute(x.detach().contiguous(), views, in_is_batch)
    return out if scale == 1.0 else out * scale


def deaug_reduce(x, views, code):
    views = list(views)
    if not x.is_cuda:
        return _host.deaug_reduce(x, views, code)
    if x.dim() != 4 or x.dtype not in (torch.float32,) + _LOW_PRECISION:
        # float64 (evaluated IN float64, like the reference), integer tensors (whatever torch says to a mean of integers) and tensors
        # of more than four dims: the inverse views as a bit-exact permutation, then the reduction over the stack -- the HIP stack
        # kernel for fp32 / half, torch's own ops in the tensor's dtype otherwise
        V = len(views)
        if x.dim() >= 4 and x.shape[0] % V != 0:
            raise RuntimeError(f"Input batch size ({x.size(0)}) must be divisible by {V}.")
        stack = view_transform(x, views, in_is_batch=False)
        stack = stack.view(V, stack.shape[0] // V, *stack.shape[1:])
        if stack.dtype in (torch.float32,) + _LOW_PRECISION:
            return stack_reduce(stack, code)
        return _host.reduce_stack(stack, code)
    if x.dtype in _LOW_PRECISION and not (x.requires_grad and torch.is_grad_enabled()):
        # inference on half-precision model outputs: read them as they are
        if x.shape[0] % len(views) != 0:
            raise RuntimeError(f"Input batch size ({x.size(0)}) must be divisible by {len(views)}.")
        return _raw_deaug_reduce(x.contiguous(), views, code).to(x.dtype)
    x, back = _check_image(x, "de-augment")
    if x.shape[0] % len(views) != 0:
        raise RuntimeError(f"Input batch size ({x.size(0)}) must be divisible by {len(views)}.")
    out = _DeaugReduce.apply(x, views, code)
    return out.to(back) if back is not None else out


def _plane_shape(numel):
    """Factor a flat length into [H, W] with W a multiple of 4 when possible (vector kernels), else one long row."""
    for w in (256, 128, 64, 32, 16, 8, 4):
        if numel % w == 0:
            return numel // w, w
    return 1, numel


class _StackReduce(torch.autograd.Function):
    """reduce dim 0 of a [T, ...] tensor: identity views of a flat plane."""

    @staticmethod
    def forward(ctx, x, code):
        ctx.code, ctx.T = code, x.shape[0]
        T = x.shape[0]
        rest = x.shape[1:]
        numel = 1
        for s in rest:
            numel *= int(s)
        if numel == 0:
            return x.new_empty(rest)
        H, W = _plane_shape(numel)
        flat = x.contiguous().view(T, 1, H, W)
        out = _raw_deaug_reduce(flat, [N.IDENT] * T, code)
        if code not in (N.RED_SUM, N.RED_MEAN):
            ctx.save_for_backward(flat, out)
        return out.view(rest)

    @staticmethod
    def backward(ctx, g):
        if ctx.code == N.RED_SUM:
            return g.unsqueeze(0).expand(ctx.T, *g.shape), None
        if ctx.code == N.RED_MEAN:
            return (g / ctx.T).unsqueeze(0).expand(ctx.T, *g.shape), None
        flat, out = ctx.saved_tensors
        grad = _raw_deaug_reduce_bwd(flat, out, g.contiguous().view(out.shape), [N.IDENT] * ctx.T, ctx.code)
        return grad.view(ctx.T, *g.shape), None


class _StackReduceAny(torch.autograd.Function):
    """reduce dim 0 of a [T, ...] tensor of any length, explicit eps: ``ptb_stack_reduce`` (one pass over the T planes)."""

    @staticmethod
    def forward(ctx, x, code, eps):
        T, rest = x.shape[0], x.shape[1:]
        flat = x.contiguous().view(T, -1)
        out = torch.empty(flat.shape[1], dtype=torch.float32, device=x.device)
        with N.on_device(x.device):
            rc = N.load().ptb_stack_reduce(flat.data_ptr(), T, flat.shape[1], code, float(eps), out.data_ptr(), N.stream_ptr(x.device))
        N.bump()
        N.check(rc, "ptb_stack_reduce")
        ctx.cfg, ctx.rest = (code, float(eps), T), rest
        ctx.save_for_backward(flat, out)
        return out.view(rest)

    @staticmethod
    def backward(ctx, g):
        code, eps, T = ctx.cfg
        flat, out = ctx.saved_tensors
        g = g.contiguous().view(-1).float()
        grad = torch.empty_like(flat)
        with N.on_device(flat.device):
            rc = N.load().ptb_stack_reduce_bwd(flat.data_ptr(), out.data_ptr(), g.data_ptr(), T, flat.shape[1], code, eps, grad.data_ptr(),
                                               N.stream_ptr(flat.device))
        N.bump()
        N.check(rc, "ptb_stack_reduce_bwd")
        return grad.view((T,) + tuple(ctx.rest)), None, None


DEFAULT_EPS = 1e-6


def stack_reduce(x, code, eps=DEFAULT_EPS):
    """Reduce dim 0 of ``x [T, ...]`` (float32, GPU) with the HIP reduction ``code``.  Up to 8 planes with the default eps go through
    the unrolled view kernels (the TTA groups); longer stacks and a caller-chosen eps through ``ptb_stack_reduce``.  Host tensors
    take the torch path of ``inference/_host.py`` (any dtype, differentiable)."""
    if not x.is_cuda:
        if x.shape[0] < 1:
            raise RuntimeError("cannot reduce an empty stack")
        return _host.reduce_stack(x, code, eps)
    N.require_device(x, "TTA reduction")
    if x.dtype in _LOW_PRECISION:
        return stack_reduce(x.float(), code, eps).to(x.dtype)
    if x.dtype != torch.float32:
        # float64 is reduced IN float64 (the HIP kernels are float32: a double-precision caller wants the reference's double-precision
        # result, not speed); integer stacks get whatever torch answers to the reference's op chain on them
        if x.shape[0] < 1:
            raise RuntimeError("cannot reduce an empty stack")
        return _host.reduce_stack(x, code, eps)
    T = x.shape[0]
    if T < 1:
        raise RuntimeError("cannot reduce an empty stack")
    if T > 8 or (eps != DEFAULT_EPS and code in (N.RED_HMEAN, N.RED_LOGODD)):
        if x.numel() == 0:
            return x.new_empty(x.shape[1:])
        return _StackReduceAny.apply(x, code, eps)
    return _StackReduce.apply(x, code)
