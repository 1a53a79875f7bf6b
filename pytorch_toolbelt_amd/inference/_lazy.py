"""Lazy de-augmentation results: what lets the reference's literal loop take the fused kernels.

The reference's hot loop is two calls (README.md:215-226; inference/tta.py:442-467 feeding inference/tiles.py:321-339)::

    merger.integrate_batch(tta.d4_image_deaugment(model(tta.d4_image_augment(tiles))), crops)

Evaluated call by call, the reduced tile travels through HBM between them (one write + one read of ``[B, C, h, w]`` and a
second launch per batch).  ``*_image_deaugment`` therefore returns a ``LazyDeaugment``: a ``torch.Tensor`` subclass that has
the result's shape / dtype / device and remembers ``(source, views, reduction)`` instead of computing anything.
``TileMerger.integrate_batch`` recognises it and launches the fused de-augment + blend kernel on the source
(``ptb_deaug_accumulate`` / the planned kernel / the band plan); ANY other use -- an operator, a method, ``print``, indexing,
``.cpu()``, ``data_ptr()``, a custom kernel -- goes through ``__torch_function__`` / ``__torch_dispatch__``, which first run
the ordinary de-augment kernel (``ptb_deaug_reduce``) once and then hand the real tensor on.  The value is cached, so in-place
operations on the handle behave like in-place operations on the eager result.

What a caller could observe, and how it is bounded:

* the source is kept alive until the handle dies or is evaluated -- at most ``PTB_LAZY_DEAUG_BYTES`` (default 4 GiB) of
  sources are kept across all pending handles, older ones are evaluated when a new one would exceed it;
* a source modified in place before the handle is evaluated would change the result: the source's version counter is
  checked at evaluation and a ``RuntimeError`` says so (tensors created under ``torch.inference_mode()`` have no counter;
  there the check is skipped);
* evaluation happens on the stream that is current *then*; when that is not the stream of the call the source is
  ``record_stream``-ed.

Code that takes the storage of a tensor without passing through the dispatcher or ``__torch_function__`` -- a pybind11 function
with an ``at::Tensor`` argument, the legacy ``torch.utils.dlpack.to_dlpack`` (guarded here, see ``_guard_legacy_dlpack``) -- sees a
tensor without storage, as with every wrapper subclass (``FakeTensor``, ``DTensor``); hand it ``handle + 0`` or switch the handles off.

Tensors made under ``torch.inference_mode()`` carry no version counter; they get a handle only when the call expression is their sole
owner (``deaugment(model(x))`` written as one expression, the model keeping no reference to its output and nothing else sharing its
storage): then nobody exists who could change the source.

Only inference-shaped calls are lazy (float32 / float16 / bfloat16 CUDA source, string reduction, no autograd, no tracing / compiling);
everything else is evaluated on the spot exactly as before.  ``tta.set_lazy_deaugment(False)`` / ``PTB_LAZY_DEAUG=0``
switch it off.
"""
import os
import threading
import weakref
from collections import OrderedDict

import torch
from torch.utils._pytree import tree_map

_ENABLED = os.environ.get("PTB_LAZY_DEAUG", "1") != "0"
_BUDGET = int(os.environ.get("PTB_LAZY_DEAUG_BYTES", str(4 << 30)))
_pending = OrderedDict()      # id(handle) -> (weakref, bytes of its source), creation order
_pending_bytes = 0
_lock = threading.RLock()     # the bookkeeping is shared by every thread that de-augments (weakref callbacks run wherever a handle dies)
evaluations = 0               # handles that had to be evaluated by themselves (diagnostics / tests)
fused = 0                     # handles a merger consumed through the fused path (diagnostics / tests)


def set_enabled(flag: bool) -> bool:
    """Switch lazy de-augmentation on / off; returns the previous setting."""
    global _ENABLED
    prev, _ENABLED = _ENABLED, bool(flag)
    if not _ENABLED:
        remove_dlpack_guard()
    return prev


def enabled() -> bool:
    return _ENABLED


def _source_version(t):
    try:
        return t._version
    except RuntimeError:      # inference tensors do not track a version counter
        return None


def _forget(key):
    global _pending_bytes
    try:
        with _lock:
            ent = _pending.pop(key, None)
            if ent is not None:
                _pending_bytes -= ent[1]
    except TypeError:      # interpreter shutdown: the module's globals are already gone when the last handles die
        pass


def _admit(handle, nbytes):
    """Register a new pending handle; evaluate the oldest ones while the sources kept alive exceed the budget."""
    global _pending_bytes
    while True:
        with _lock:
            if not _pending or _pending_bytes + nbytes <= _BUDGET:
                break
            key, (ref, _n) = next(iter(_pending.items()))
        old = ref()
        if old is None:
            _forget(key)
        else:
            try:
                old._evaluate()   # (forgets itself)
            except RuntimeError:  # its source was modified: that is for ITS user to hear about, not for this unrelated call
                _forget(key)
    key = id(handle)
    with _lock:
        _pending[key] = (weakref.ref(handle, lambda _r, k=key: _forget(k)), nbytes)
        _pending_bytes += nbytes


# attribute getters / methods that only look at metadata the wrapper itself carries: answered without evaluating
_META_NAMES = ("shape", "dtype", "device", "ndim", "is_cuda", "is_cpu", "layout", "requires_grad", "grad_fn", "is_leaf", "is_sparse",
               "is_quantized", "is_meta", "names", "grad", "is_nested", "is_mkldnn", "is_xpu", "is_mps", "output_nr")
_META_FUNCS = set()
for _n in _META_NAMES:
    _p = getattr(torch.Tensor, _n, None)
    if _p is not None and hasattr(_p, "__get__"):
        _META_FUNCS.add(_p.__get__)
for _n in ("size", "dim", "ndimension", "numel", "nelement", "__len__", "element_size", "is_floating_point", "is_complex", "get_device",
           "is_contiguous", "stride", "storage_offset", "is_inference", "is_signed", "is_shared", "is_pinned", "is_same_size", "has_names"):
    _f = getattr(torch.Tensor, _n, None)
    if _f is not None:
        _META_FUNCS.add(_f)
del _n, _p, _f


_CONTAINERS = (list, tuple, dict)
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


class LazyDeaugment(torch.Tensor):
    """Result of ``<group>_image_deaugment(source, reduction)`` that has not been computed yet (see the module docstring)."""

    @staticmethod
    def __new__(cls, source, group, views, code, compute):
        n_views = len(views)
        shape = (source.shape[0] // n_views,) + tuple(source.shape[1:])
        return torch.Tensor._make_wrapper_subclass(cls, shape, dtype=source.dtype, device=source.device, requires_grad=False)

    def __init__(self, source, group, views, code, compute):
        if _dlpack_orig is None:
            _guard_legacy_dlpack()         # (first handle of the process: from here on to_dlpack(handle) must evaluate it first)
        self._len = source.shape[0] // len(views)      # (len(handle) without the __torch_function__ round trip)
        self._src = source                 # [V*B, C, H, W] contiguous float32 / float16 / bfloat16 model output (chunk-major)
        self._group = group                # "d4" | "d2" | "flips" | "fliplr" | "flipud"
        self._views = views                # inverse view codes, chunk order
        self._code = code                  # HIP reduction code
        self._compute = compute            # (source, views, code) -> tensor: the eager kernel
        self._value = None
        self._src_version = _source_version(source)
        self._stream = _raw_stream(source.device.index) if (source.is_cuda and _raw_stream is not None) else None   # raw handle: 0.3 us
        _admit(self, source.numel() * source.element_size())

    # ---------------------------------------------------------------- evaluation
    def _check_source(self):
        if self._src_version is not None and _source_version(self._src) != self._src_version:
            raise RuntimeError("the model output passed to *_image_deaugment was modified in place before the (lazily evaluated) result was "
                               "used; evaluate the result first, or switch lazy de-augmentation off (tta.set_lazy_deaugment(False))")

    def _note_stream(self):
        """Evaluated / consumed on another stream than the one the handle was made on (budget eviction from another thread's call, a
        caller that switched streams): that stream first waits for everything queued on the creating stream -- the producer of the
        source among it -- and the source is ``record_stream``-ed so that its memory outlives this read."""
        src = self._src
        if self._stream is not None and _raw_stream(src.device.index) != self._stream:
            cur = torch.cuda.current_stream(src.device)
            cur.wait_stream(torch.cuda.ExternalStream(self._stream, device=src.device))
            src.record_stream(cur)

    def _evaluate(self):
        """The real tensor (computed once)."""
        value = self._value
        if value is None:
            global evaluations
            with _lock:                      # two threads using one handle must end up with ONE value (in-place operations act on it)
                value = self._value
                if value is None:
                    self._check_source()
                    self._note_stream()
                    with torch._C.DisableTorchFunctionSubclass():
                        value = self._compute(self._src, self._views, self._code)
                    self._value = value
                    self._src = None
                    _forget(id(self))
                    evaluations += 1
        return value

    def _take_source(self):
        """For a merger: ``(source, group, views, code)`` when the fused path may still be taken, else None.  The handle stays
        valid (a later use evaluates it from the same source)."""
        if self._value is not None:
            return None
        self._check_source()
        self._note_stream()
        return self._src, self._group, self._views, self._code

    # ---------------------------------------------------------------- protocol
    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in _META_FUNCS and args and type(args[0]) is LazyDeaugment and args[0]._value is None:
            # (only while nothing was computed: afterwards requires_grad / grad / grad_fn / is_leaf are those of the value)
            with torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)
        # (flat argument lists -- almost every call -- are unwrapped by hand: tree_map costs ~10 us per call)
        args = tuple(a._evaluate() if type(a) is LazyDeaugment else (tree_map(_unwrap, a) if isinstance(a, _CONTAINERS) else a) for a in args)
        if kwargs:
            kwargs = {k: (v._evaluate() if type(v) is LazyDeaugment else (tree_map(_unwrap, v) if isinstance(v, _CONTAINERS) else v)) for k, v in kwargs.items()}
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **kwargs)

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):   # safety net: code that reaches ATen without __torch_function__
        kwargs = kwargs or {}
        return func(*tree_map(_unwrap, args), **tree_map(_unwrap, kwargs))

    def __repr__(self):   # noqa: D105
        return repr(self._evaluate())


def _unwrap(x):
    return x._evaluate() if type(x) is LazyDeaugment else x


_dlpack_orig = None        # torch.utils.dlpack.to_dlpack as it was before the guard went in (None: no guard installed)


def _guard_legacy_dlpack():
    """``torch.utils.dlpack.to_dlpack`` is a bare C function: it reads the storage of what it is given without passing
    ``__torch_function__``, and a handle has none (the capsule would carry a null pointer).  The module attribute is replaced by a
    wrapper that evaluates a handle first; ``torch.from_dlpack(handle)`` / ``handle.__dlpack__()`` never needed it.  Installed when
    the FIRST handle is created (importing the package changes nothing in torch), taken out again by ``remove_dlpack_guard()`` /
    ``set_enabled(False)``."""
    global _dlpack_orig
    import torch.utils.dlpack as D

    orig = D.to_dlpack
    if getattr(orig, "_ptb_lazy_guard", False):
        return

    def to_dlpack(tensor):
        return orig(tensor._evaluate() if type(tensor) is LazyDeaugment else tensor)

    to_dlpack.__doc__ = getattr(orig, "__doc__", None)
    to_dlpack._ptb_lazy_guard = True
    _dlpack_orig = orig
    D.to_dlpack = to_dlpack
    if getattr(torch, "to_dlpack", None) is orig:
        torch.to_dlpack = to_dlpack


def remove_dlpack_guard():
    """Put ``torch.utils.dlpack.to_dlpack`` back (handles still alive must then not be handed to it)."""
    global _dlpack_orig
    import torch.utils.dlpack as D

    if _dlpack_orig is not None and getattr(D.to_dlpack, "_ptb_lazy_guard", False):
        if getattr(torch, "to_dlpack", None) is D.to_dlpack:
            torch.to_dlpack = _dlpack_orig
        D.to_dlpack = _dlpack_orig
    _dlpack_orig = None


def _storage_owners(t):
    return torch._C._storage_Use_Count(t.untyped_storage()._cdata)


try:
    _ONE_OWNER = _storage_owners(torch.empty(1))      # what the count reads for a tensor that is its storage's only owner
except Exception:  # noqa: BLE001  (a torch without the private counter: no handles for tensors without a version counter)
    _ONE_OWNER = None


def _sole_owner(t):
    """Nobody but ``t`` can reach its memory: no view of it, no base tensor, no second tensor made from its storage is alive."""
    try:
        return _ONE_OWNER is not None and _storage_owners(t) <= _ONE_OWNER
    except Exception:  # noqa: BLE001
        return False


# float32, and (round 6) the half-precision outputs of a model under torch.autocast: the handle of a half source stands for the HALF
# tensor the eager call returns (the fp32 reduction rounded once to the source dtype), and a merger that fuses it rounds the reduced value
# the same way in registers (PTB_ROUND_SRC) -- fused and evaluated results are bit-identical
_LAZY_DTYPES = (torch.float32, torch.float16, torch.bfloat16)


def maybe_lazy(source, group, views, code, compute, owned=False):
    """A ``LazyDeaugment`` for this call when it is inference-shaped, else None (the caller evaluates eagerly).  ``owned``: the public
    function saw its argument referenced by nobody but the call expression (``tta._TEMP_REFS``)."""
    if not _ENABLED or type(source) is not torch.Tensor or not source.is_cuda or source.dtype not in _LAZY_DTYPES or source.dim() != 4:
        return None
    if source.requires_grad and torch.is_grad_enabled():
        return None
    n_views = len(views)
    if source.shape[0] % n_views != 0 or source.numel() == 0 or not source.is_contiguous():
        return None
    if any(v & 1 for v in views) and source.shape[2] != source.shape[3]:
        return None
    if torch.jit.is_tracing() or torch.jit.is_scripting() or torch.compiler.is_compiling():
        return None
    # A handle reads its source LATER.  That is only safe while an in-place change of the source in between can be noticed: tensors
    # made under torch.inference_mode() carry no version counter, and a stream that is being captured into a HIP graph replays into
    # the same (static) output buffers -- both are evaluated here and now, exactly like the reference.
    # Round 5: without a version counter a handle is still safe when NOBODY else can reach the source -- a temporary of the call
    # expression (`integrate_batch(d4_image_deaugment(model(x)), crops)` under torch.inference_mode(), the recommended context of an
    # inference loop) that is the only owner of its storage: there is no one to change it before the handle is consumed.
    if torch.cuda.is_current_stream_capturing():
        return None
    if _source_version(source) is None and not (owned and _sole_owner(source)):
        return None
    return LazyDeaugment(source, group, views, code, compute)
