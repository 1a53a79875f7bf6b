"""Model ensembling with the reference's API (``pytorch_toolbelt/inference/ensembling.py``), MI355X-native.

``Ensembler`` averages the outputs of several models; ``ApplySigmoidTo`` / ``ApplySoftmaxTo`` wrap a model and activate
selected outputs.  The reference stacks the T outputs (one full copy) and runs ``_deaugment_averaging`` over the stack,
after each wrapper has made its own elementwise passes (ensembling.py:38-42, 62-66, 89-123).  Here one HIP launch
(``ptb_ensemble_reduce``) reads the T raw outputs in place, applies the wrappers' activation in registers, reduces in
list order and writes the ensemble once.  Tensors that take part in autograd use the differentiable stack path instead.
"""
import collections
from typing import Dict, Iterable, List, Optional, Tuple, Union

import ctypes
import torch
from torch import Tensor, nn

from .. import _native as N
from .tta import _deaugment_averaging, _reduction_code

__all__ = ["ApplySoftmaxTo", "ApplySigmoidTo", "Ensembler", "PickModelOutput", "SelectByIndex", "average_checkpoints"]

_ACT_NONE, _ACT_SIGMOID, _ACT_SOFTMAX = 0, 1, 2
_MAX_MODELS = 16


def _native_ok(t: Tensor) -> bool:
    return (torch.is_tensor(t) and t.is_cuda and t.dtype in (torch.float32, torch.float16, torch.bfloat16)
            and not (t.requires_grad and torch.is_grad_enabled()))


def _as_bchw(shape, act, dim):
    """(B, C, HW) factorisation of a tensor shape for the kernel; None if the softmax dim cannot be expressed."""
    numel = 1
    for s in shape:
        numel *= int(s)
    if act != _ACT_SOFTMAX:
        return 1, 1, numel
    nd = len(shape)
    if nd == 0:
        return None
    d = dim % nd
    B = 1
    for s in shape[:d]:
        B *= int(s)
    HW = 1
    for s in shape[d + 1:]:
        HW *= int(s)
    return B, int(shape[d]), HW


def _ensemble_native(tensors: List[Tensor], code: int, act: int, temperature: float, dim: int = 1) -> Tensor:
    """reduce_t act(tensors[t]) as one HIP launch; tensors share a shape and live on one GPU."""
    first = tensors[0]
    back = first.dtype if first.dtype != torch.float32 else None
    xs = [t.detach().float().contiguous() for t in tensors]
    out = torch.empty_like(xs[0])
    fact = _as_bchw(first.shape, act, dim)
    if out.numel() == 0:
        return out.to(back) if back is not None else out
    B, C, HW = fact
    ptrs = (ctypes.c_void_p * len(xs))(*[x.data_ptr() for x in xs])
    lib = N.load()
    with N.on_device(first.device):
        rc = lib.ptb_ensemble_reduce(ptrs, len(xs), code, act, float(temperature), B, C, HW, out.data_ptr(), N.stream_ptr(first.device))
    N.bump()
    N.check(rc, "Ensembler")
    return out.to(back) if back is not None else out


def _activate(x: Tensor, act: int, temperature: float, dim: int) -> Tensor:
    """The wrapper's activation on one tensor: HIP (a T = 1 ensemble) when it does not need autograd."""
    if act == _ACT_NONE:
        return x
    if _native_ok(x) and _as_bchw(x.shape, act, dim) is not None:
        return _ensemble_native([x], N.RED_SUM, act, temperature, dim)
    scaled = x.mul(temperature)
    return scaled.sigmoid_() if act == _ACT_SIGMOID else scaled.softmax(dim=dim)


def _keys(output_key) -> Tuple:
    # a set prevents double activation for output_key=["logits", "logits"]
    return (output_key,) if isinstance(output_key, (str, int)) else tuple(set(output_key))


class ApplySoftmaxTo(nn.Module):
    """Apply ``softmax(output * temperature, dim)`` to the chosen output(s) of ``model`` (a dict or list of tensors)."""

    output_keys: Tuple
    temperature: float
    dim: int
    _act = _ACT_SOFTMAX

    def __init__(self, model: nn.Module, output_key: Union[str, int, Iterable[str]] = "logits", dim: int = 1, temperature: float = 1):
        super().__init__()
        self.output_keys = _keys(output_key)
        self.model = model
        self.dim = dim
        self.temperature = temperature

    def forward(self, *input, **kwargs):
        output = self.model(*input, **kwargs)
        for key in self.output_keys:
            output[key] = _activate(output[key], _ACT_SOFTMAX, self.temperature, self.dim)
        return output


class ApplySigmoidTo(nn.Module):
    """Apply ``sigmoid(output * temperature)`` to the chosen output(s) of ``model`` (a dict or list of tensors)."""

    output_keys: Tuple
    temperature: float
    _act = _ACT_SIGMOID
    dim = 1

    def __init__(self, model: nn.Module, output_key: Union[str, int, Iterable[str]] = "logits", temperature=1):
        super().__init__()
        self.output_keys = _keys(output_key)
        self.model = model
        self.temperature = temperature

    def forward(self, *input, **kwargs):  # skipcq: PYL-W0221
        output = self.model(*input, **kwargs)
        for key in self.output_keys:
            output[key] = _activate(output[key], _ACT_SIGMOID, self.temperature, 1)
        return output


class Ensembler(nn.Module):
    """Sum / average (``reduction``: 'mean', 'sum', 'gmean', 'hmean', ..., a callable, or None) the outputs of several
    models.  ``outputs``: names of the model outputs to reduce and return (default: all outputs of the first model)."""

    __slots__ = ["outputs", "reduction", "return_some_outputs"]

    def __init__(self, models: List[nn.Module], reduction: str = "mean", outputs: Optional[Iterable[str]] = None):
        super().__init__()
        self.return_some_outputs = outputs is not None
        self.outputs = tuple(outputs) if outputs else tuple()
        self.models = nn.ModuleList(models)
        self.reduction = reduction

    # -- one model: its output and the activations still owed to it (wrapper peeled so the kernel can fuse them)
    def _run(self, model, input, kwargs):
        if type(model) in (ApplySigmoidTo, ApplySoftmaxTo):
            raw = model.model(*input, **kwargs)
            owed = {k: (model._act, float(model.temperature), int(model.dim)) for k in model.output_keys}
            if isinstance(raw, (dict, list)):
                return raw, owed
            for key, a in owed.items():   # tensor / tuple output: item assignment fails exactly like the reference's wrapper
                raw[key] = _activate(raw[key], *a)
            return raw, {}
        return model(*input, **kwargs), {}

    def _reduce(self, preds: List[Tensor], acts: List[Tuple[int, float, int]]) -> Tensor:
        code = _reduction_code(self.reduction)
        same_shape = all(torch.is_tensor(p) and p.shape == preds[0].shape and p.device == preds[0].device for p in preds)
        if code is not None and same_shape and len(preds) <= _MAX_MODELS and all(_native_ok(p) for p in preds):
            uniform = all(a == acts[0] for a in acts)
            act, temperature, dim = acts[0]
            if uniform and _as_bchw(preds[0].shape, act, dim) is not None:
                return _ensemble_native(preds, code, act, temperature, dim)
            preds = [_activate(p, *a) for p, a in zip(preds, acts)]
            return _ensemble_native(preds, code, _ACT_NONE, 1.0)
        preds = [_activate(p, *a) for p, a in zip(preds, acts)]
        return _deaugment_averaging(torch.stack(preds), self.reduction)

    def _selected(self, first):
        """What of the models' outputs is reduced, decided by the first model's output like the reference (ensembling.py:94-107):
        (keys, builder of the result container); keys None = the output is one tensor."""
        as_dict = isinstance(first, dict)
        as_seq = isinstance(first, (list, tuple))
        if self.return_some_outputs:
            return self.outputs, as_dict
        if as_dict:
            return first.keys(), True
        if as_seq:
            return range(len(first)), False
        if torch.is_tensor(first):
            return None, False
        raise RuntimeError()

    def forward(self, *input, **kwargs):  # skipcq: PYL-W0221
        outputs, owed = zip(*(self._run(model, input, kwargs) for model in self.models))
        plain = (_ACT_NONE, 1.0, 1)
        keys, as_dict = self._selected(outputs[0])
        if keys is None:
            return self._reduce(list(outputs), [plain] * len(outputs))
        reduced = [(key, self._reduce([out[key] for out in outputs], [o.get(key, plain) for o in owed])) for key in keys]
        return dict(reduced) if as_dict else [value for _key, value in reduced]


class PickModelOutput(nn.Module):
    """Wrap a model that returns a dict or list and return only one element of it."""

    __slots__ = ["target_key"]

    def __init__(self, model: nn.Module, key: Union[str, int]):
        super().__init__()
        self.model = model
        self.target_key = key

    def forward(self, *input, **kwargs) -> Tensor:
        return self.model(*input, **kwargs)[self.target_key]


class SelectByIndex(nn.Module):
    """Select a single tensor from a dict or list of output tensors (for use inside ``nn.Sequential``)."""

    __slots__ = ["target_key"]

    def __init__(self, key: Union[str, int]):
        super().__init__()
        self.target_key = key

    def forward(self, outputs: Dict[str, Tensor]) -> Tensor:
        return outputs[self.target_key]


def average_checkpoints(inputs: List[str]) -> collections.OrderedDict:
    """Average the ``model_state_dict`` of several checkpoint files (host-side utility; the first checkpoint supplies
    every other field).  Floating-point parameters are divided, integer buffers floor-divided, by the file count;
    a parameter-name mismatch raises ``KeyError``."""
    total = collections.OrderedDict()
    names, state0 = None, None
    for path in inputs:
        with open(path, "rb") as f:
            state = torch.load(f, map_location="cpu")
        if state0 is None:
            state0 = state
        params = state["model_state_dict"]
        if names is None:
            names = list(params.keys())
        elif names != list(params.keys()):
            raise KeyError(f"For checkpoint {path}, expected list of params: {names}, but found: {list(params.keys())}")
        for k in names:
            p = params[k]
            if p.dtype == torch.float16:
                p = p.float()
            if k in total:
                total[k] += p
            else:
                total[k] = p.clone()  # (shared parameters must not be accumulated into twice)
    n = len(inputs)
    for k, v in total.items():
        if v.is_floating_point():
            v.div_(n)
        else:
            v //= n
    state0["model_state_dict"] = total
    return state0
