"""CPU: a lazy de-augmentation handle (inference/_lazy.py) against the eager tensor over EVERY public attribute / method of
torch.Tensor and every callable of the torch namespace that takes a tensor: same value, or the same exception type.  A handle is a
storage-less wrapper subclass; anything that is not routed through __torch_function__ / __torch_dispatch__ would show up here."""
import inspect
import signal
import warnings

import torch

from pytorch_toolbelt_amd.inference import _lazy as L


def _mean_views(src, views, code):
    V = len(views)
    return src.view(V, src.shape[0] // V, *src.shape[1:]).mean(0)


def _same(a, b):
    if isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor):
        if a.shape != b.shape or a.dtype != b.dtype:
            return False
        if a.layout != torch.strided or b.layout != torch.strided:
            return a.layout == b.layout
        if a.is_quantized or b.is_quantized:
            return a.is_quantized == b.is_quantized
        try:
            return torch.equal(torch.nan_to_num(a.detach().float()), torch.nan_to_num(b.detach().float()))
        except Exception:  # noqa: BLE001
            return True
    if isinstance(a, (tuple, list)) and isinstance(b, (tuple, list)):
        return len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    if callable(a) and callable(b):
        return True
    try:
        return bool(a == b) or (a != a and b != b)
    except Exception:  # noqa: BLE001
        return type(a) is type(b)


# not comparable by value / act on process state / need arguments a sweep cannot guess
_SKIP_ATTR = {"share_memory_", "pin_memory", "cuda", "xpu", "ipu", "hpu", "mtia", "to_mkldnn", "backward", "register_hook",
              "register_post_accumulate_grad_hook", "retain_grad", "rename_", "refine_names", "align_as", "align_to", "storage",
              "untyped_storage", "data_ptr", "_typed_storage", "random_", "normal_", "uniform_", "cauchy_", "exponential_", "geometric_",
              "log_normal_", "bernoulli_", "bernoulli", "multinomial", "set_", "resize_", "resize_as_", "requires_grad_", "detach_", "zero_",
              "volatile", "is_shared", "_version", "grad", "data", "names", "T", "H", "mH", "mT", "real", "imag", "_base", "_grad", "_cdata",
              "itemsize", "nbytes", "module_load", "apply_", "map_", "map2_", "stride", "new", "type", "tolist", "item",
              "_python_dispatch", "_reduce_ex_internal"}
_SKIP_FN = {"save", "load", "manual_seed", "seed", "compile", "export", "typename", "is_storage", "from_dlpack", "to_dlpack", "rand_like",
            "randn_like", "randint_like", "empty_like", "bernoulli", "multinomial", "normal", "poisson", "dropout", "alpha_dropout",
            "feature_dropout", "feature_alpha_dropout", "rrelu", "native_dropout", "binomial", "result_type", "can_cast", "promote_types",
            "get_device_module", "cond", "while_loop", "vmap", "autocast", "enable_grad", "no_grad", "inference_mode", "unravel_index",
            "init_num_threads", "fork", "wait", "prepare_multiprocessing_environment", "use_deterministic_algorithms",
            "classproperty", "fbgemm_pack_gemm_matrix_fp16", "lobpcg", "pca_lowrank", "svd_lowrank"}     # (the last three draw random numbers)


class _Timeout(Exception):
    pass


def _alarm(*_a):
    raise _Timeout()


def _outcome(call):
    old = signal.signal(signal.SIGALRM, _alarm)
    signal.alarm(10)
    try:
        return ("ok", call())
    except _Timeout:
        return ("err", "timeout")
    except Exception as e:  # noqa: BLE001
        return ("err", type(e).__name__)
    finally:
        signal.alarm(0)
        signal.signal(signal.SIGALRM, old)


def _compare(run_on_handle, run_on_eager):
    ra, rb = _outcome(run_on_handle), _outcome(run_on_eager)
    return ra[0] == rb[0] and (_same(ra[1], rb[1]) if ra[0] == "ok" else ra[1] == rb[1]), ra, rb


def test_every_tensor_attribute_and_torch_function_sees_the_eager_value():
    warnings.filterwarnings("ignore")
    src = torch.rand(8, 2, 4, 4) + 0.1
    want = _mean_views(src, (0, 4), 1)
    bad, swept = [], 0

    def fresh():
        return L.LazyDeaugment(src.clone(), "fliplr", (0, 4), 1, _mean_views), want.clone()

    for name in [n for n in dir(torch.Tensor) if not n.startswith("__") and n not in _SKIP_ATTR]:
        h, w = fresh()

        def use(obj, name=name):
            a = getattr(obj, name)
            if not callable(a):
                return a
            try:
                return a()
            except TypeError:
                return a(obj)                  # binary form, with itself

        ok, ra, rb = _compare(lambda: use(h), lambda: use(w))
        swept += 1
        if not ok:
            bad.append((f"Tensor.{name}", ra[0], str(ra[1])[:60], rb[0], str(rb[1])[:60]))
    for name in [n for n in dir(torch) if not n.startswith("_") and n not in _SKIP_FN and not n.startswith("set_") and not n.startswith("sym_")]:
        f = getattr(torch, name)
        if not callable(f) or inspect.isclass(f) or inspect.ismodule(f):
            continue
        h, w = fresh()

        def use(x, f=f):
            try:
                return f(x)
            except TypeError:
                return f(x, x)

        ok, ra, rb = _compare(lambda: use(h), lambda: use(w))
        swept += 1
        if not ok:
            bad.append((f"torch.{name}", ra[0], str(ra[1])[:60], rb[0], str(rb[1])[:60]))
    assert swept > 1000 and not bad, bad


def test_torch_nn_functional_sees_the_eager_value():
    """The same sweep over torch.nn.functional (activations, pooling, interpolate, normalisation ...: what usually follows a TTA merge)."""
    import torch.nn.functional as F

    warnings.filterwarnings("ignore")
    src = torch.rand(8, 2, 4, 4) + 0.1
    want = _mean_views(src, (0, 4), 1)
    random_or_introspective = {"dropout", "dropout1d", "dropout2d", "dropout3d", "alpha_dropout", "feature_alpha_dropout", "rrelu", "gumbel_softmax",
                               "fractional_max_pool2d", "fractional_max_pool3d", "fractional_max_pool2d_with_indices",
                               "fractional_max_pool3d_with_indices", "has_torch_function_unary", "has_torch_function_variadic", "has_torch_function"}
    bad, succeeded = [], 0
    for name in dir(F):
        f = getattr(F, name)
        if name.startswith("_") or name in random_or_introspective or not callable(f) or inspect.isclass(f) or inspect.ismodule(f):
            continue
        for args in ((), (2,), ((2, 2),), (1,)):           # the first argument form the eager tensor accepts
            h, w = L.LazyDeaugment(src.clone(), "fliplr", (0, 4), 1, _mean_views), want.clone()
            ok, ra, rb = _compare(lambda: f(h, *args), lambda: f(w, *args))
            if rb[0] == "ok":
                succeeded += 1
                if not ok:
                    bad.append((name, args, ra[0], str(ra[1])[:60]))
                break
    assert succeeded >= 50 and not bad, bad
