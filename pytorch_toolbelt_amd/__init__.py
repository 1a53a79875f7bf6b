"""pytorch_toolbelt_amd -- MI355X-native (gfx950) implementation of pytorch-toolbelt's large-image inference hot path.

Same Python surface as ``pytorch_toolbelt.inference.{tiles,tta,functional}`` and ``pytorch_toolbelt.losses``; the
tensor math runs as hand-written HIP kernels from ``lib/libptb_hip.so`` (C ABI in ``include/ptb_hip.h``).
"""
__version__ = "0.1.0"
