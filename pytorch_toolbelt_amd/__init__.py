"""pytorch_toolbelt_amd -- MI355X-native (gfx950) implementation of pytorch-toolbelt's large-image inference hot path.

Same Python surface as ``pytorch_toolbelt.inference.{tiles,tta,functional}`` and ``pytorch_toolbelt.losses``; the
tensor math runs as hand-written HIP kernels from ``lib/libptb_hip.so`` (C ABI in ``include/ptb_hip.h``).
"""
__version__ = "0.1.0"


_before_strict = None      # (lazy de-augmentation, self-planning, reference accumulators) as they were when strict mode was switched on


def set_strict_dropin(flag: bool = True):
    """Strict drop-in mode (extension, env-free): every call is evaluated where and when the reference evaluates it --
    ``*_image_deaugment`` return real tensors (no lazy handles: ``inference/_lazy.py``) and a ``TileMerger`` without ``crops=`` never
    plans itself from the previous image (``inference/tiles.py``).  Results are the same either way; what changes is that nothing
    is fused across the two calls of ``merger.integrate_batch(tta.d4_image_deaugment(y), crops)``, and no model output is read
    after the ``integrate_batch`` call that was handed it.  Strict mode also keeps float16 / bfloat16 accumulators of a CUDA
    ``TileMerger(dtype=...)`` in that dtype, like the reference (``tiles.set_reference_accumulators``: torch ops on the device instead of
    the HIP merger's float32 sums -- the one place where the defaults are closer to the exact result than to the reference's bits).
    ``flag=False`` puts back what was in force when strict mode was switched on (``PTB_LAZY_DEAUG`` / ``PTB_AUTO_PLAN`` /
    ``PTB_REFERENCE_ACCUMULATORS`` and earlier ``set_*`` calls included -- not hard-coded defaults; without a preceding
    ``set_strict_dropin(True)`` it changes nothing).  Returns the previous ``(lazy de-augmentation, self-planning, reference
    accumulators)`` settings."""
    global _before_strict
    from .inference import _lazy, tiles

    if flag:
        prev = (_lazy.set_enabled(False), tiles.set_auto_plan(False), tiles.set_reference_accumulators(True))
        if _before_strict is None:
            _before_strict = prev
        return prev
    prev = (_lazy.enabled(), tiles._AUTO_PLAN, tiles._REFERENCE_ACCUMULATORS)
    if _before_strict is not None:
        lazy, plan, ref_acc = _before_strict
        _before_strict = None
        _lazy.set_enabled(lazy)
        tiles.set_auto_plan(plan)
        tiles.set_reference_accumulators(ref_acc)
    return prev
