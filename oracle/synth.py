"""Deterministic synthetic tensors that numpy (build container, CPU) and torch (GPU box, cuda) produce BIT-IDENTICALLY.

Test infrastructure only (like everything under oracle/).  The full-size parity tests need the same multi-gigabyte inputs on
both sides of the comparison without shipping them: the golden digests in tests/golden/fullsize.npz are made here, without a
GPU, by the unmodified reference; the `-m gpu` tests regenerate the inputs on the device.  A numpy / torch random generator
cannot do that, an integer hash of the element index can: all arithmetic below is exact (int64 with 32-bit masks, then an
integer < 2^24 scaled by a power of two), so there is no rounding that could differ between the two libraries.
"""
import numpy as np

_M32 = 0xFFFFFFFF


def _mix(h, xp):
    """32-bit avalanche (lowbias32-style) on int64 lanes holding values < 2^32; `xp` is numpy or torch."""
    h = h ^ (h >> 16)
    h = (h * 0x7FEB352D) & _M32
    h = h ^ (h >> 15)
    h = (h * 0x2C1B3C6D) & _M32      # (both multipliers < 2^31: h * m < 2^63, no int64 wrap-around anywhere)
    h = h ^ (h >> 16)
    return h


def _bits24_np(n, seed):
    idx = np.arange(n, dtype=np.int64)
    h = (idx * 0x9E3779B1 + (int(seed) * 0x85EBCA77 + 0x165667B1)) & _M32
    return _mix(h, np) >> 8                      # 24 random bits, int64


def _bits24_torch(n, seed, device):
    import torch

    idx = torch.arange(n, dtype=torch.int64, device=device)
    h = (idx * 0x9E3779B1 + (int(seed) * 0x85EBCA77 + 0x165667B1)) & _M32
    return _mix(h, torch) >> 8


# kind -> (a, b): value = (k + a) * 2^-24 * scale + b   with k the 24 random bits
#   "sym"  : uniform on [-2, 2)      ((k - 2^23) * 2^-22, exact)
#   "unit" : uniform on (0, 1]       ((k + 1) * 2^-24, exact; positive -> legal input of gmean / log)
def _finish(k, kind, as_float):
    if kind == "sym":
        return as_float(k - (1 << 23)) * (2.0 ** -22)
    if kind == "unit":
        return as_float(k + 1) * (2.0 ** -24)
    raise KeyError(kind)


def synth_np(shape, seed, kind="sym"):
    n = int(np.prod(shape))
    return _finish(_bits24_np(n, seed), kind, lambda a: a.astype(np.float32)).astype(np.float32).reshape(shape)


def synth_torch(shape, seed, kind="sym", device="cuda"):
    import torch

    n = int(np.prod(shape))
    return _finish(_bits24_torch(n, seed, device), kind, lambda a: a.to(torch.float32)).reshape(shape)


def labels_np(shape, seed, classes):
    n = int(np.prod(shape))
    return (_bits24_np(n, seed) % int(classes)).astype(np.int64).reshape(shape)


def labels_torch(shape, seed, classes, device="cuda"):
    n = int(np.prod(shape))
    return (_bits24_torch(n, seed, device) % int(classes)).reshape(shape)


def digest(arr, step_h=97, step_w=101):
    """What a golden fixture keeps of a big [..., H, W] float array: a strided subsample (values), and float64 sum / abs-sum."""
    a = np.asarray(arr)
    return a[..., ::step_h, ::step_w].astype(np.float32).copy(), np.array([a.sum(dtype=np.float64), np.abs(a).sum(dtype=np.float64)])
