"""Deprecation decorator (drop-in for ``pytorch_toolbelt.utils.support.pytorch_toolbelt_deprecated``)."""
import functools
import inspect
import warnings

__all__ = ["pytorch_toolbelt_deprecated"]


def _wrap(obj, reason):
    kind = "class" if inspect.isclass(obj) else "function"
    text = f"Call to deprecated {kind} {obj.__name__}" + (f" ({reason})." if reason else ".")

    @functools.wraps(obj)
    def guarded(*args, **kwargs):
        with warnings.catch_warnings():
            warnings.simplefilter("always", DeprecationWarning)
            warnings.warn(text, category=DeprecationWarning, stacklevel=2)
        return obj(*args, **kwargs)

    return guarded


def pytorch_toolbelt_deprecated(reason):
    """``@pytorch_toolbelt_deprecated("why")`` or bare ``@pytorch_toolbelt_deprecated``: warn on every call."""
    if isinstance(reason, (str, bytes)):
        return lambda obj: _wrap(obj, reason)
    if inspect.isclass(reason) or inspect.isfunction(reason):
        return _wrap(reason, None)
    raise TypeError(repr(type(reason)))
