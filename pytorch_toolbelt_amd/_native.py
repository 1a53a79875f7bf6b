"""ctypes binding of libptb_hip.so (the C ABI declared in include/ptb_hip.h).

There is deliberately NO fallback behind this binding: if the shared library is missing, or a tensor that reaches an entry point
is not on the MI355X, the call fails loudly.  (Host tensors never reach it: the public functions route them -- by their device, and
by nothing else -- to the separate torch-op modules ``inference/_host.py`` / ``losses/_host.py``, like the device-agnostic
reference.)  PyTorch is used only for device memory, the current HIP stream and autograd glue.
"""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# PTB_HIP_LIB: load another build of the library (A/B experiments, an integrator's own build location)
LIB_PATH = os.environ.get("PTB_HIP_LIB") or os.path.join(_HERE, "lib", "libptb_hip.so")

# view codes (bit0 transpose, bit1 flip source rows, bit2 flip source cols) -- include/ptb_hip.h
IDENT, TRANSPOSE, FLIPUD, ROT90_CW, FLIPLR, ROT90_CCW, ROT180, ANTITRANSPOSE = range(8)

RED_SUM, RED_MEAN, RED_GMEAN, RED_HMEAN, RED_HARMONIC1P, RED_LOGODD, RED_LOG1P = range(7)
F32, F16, BF16 = range(3)   # PTB_F32 / PTB_F16 / PTB_BF16: element type of the model outputs a `_t` entry point reads
DTYPE_CODES = {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}
ROUND_SRC = 0x100           # PTB_ROUND_SRC, or-ed into a dtype code: round the reduced value to the (half) source type before blending

EFRESH = -5
PTB_EHELD = -6
PTB_EUNSUPPORTED = -2
_ERR = {-1: "invalid argument", -2: "unsupported configuration", -3: "HIP launch failed", -4: "tile rectangle outside the accumulator"}

_c_int = ctypes.c_int
_c_f = ctypes.c_float
_c_d = ctypes.c_double
_c_i64 = ctypes.c_int64
_vp = ctypes.c_void_p
_i64p = ctypes.POINTER(ctypes.c_int64)
_ip = ctypes.POINTER(ctypes.c_int)
_fp = ctypes.POINTER(ctypes.c_float)
_vpp = ctypes.POINTER(ctypes.c_void_p)

# name -> (restype, argtypes); must list every symbol of include/ptb_hip.h (checked by tests/test_abi.py)
SIGNATURES = {
    "ptb_version": (_c_int, []),
    "ptb_last_hip_error": (ctypes.c_char_p, []),
    "ptb_set_tunable": (_c_int, [_c_int, _c_int]),
    "ptb_read_probe": (_c_int, [_vp, _c_i64, _vp, _vp]),
    "ptb_read_probe_multi": (_c_i64, [_vpp, _i64p, _c_int, _vp, _c_int, _vp]),
    "ptb_sums_finalize": (_c_int, [_vp, _c_int, _c_int, _vp, _vp, _vp]),
    "ptb_tile_accumulate": (_c_int, [_vp, _vp, _vp, _vp, _i64p, _i64p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _c_int, _vp]),
    "ptb_deaug_reduce_t": (_c_int, [_vp, _c_int, _vp, _c_int, _ip, _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ptb_deaug_accumulate_t": (_c_int, [_vp, _vp, _vp, _vp, _c_int, _c_int, _ip, _c_int, _i64p, _i64p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _c_int, _vp]),
    "ptb_accumulate_planned": (_c_int, [_vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _ip, _c_int, _i64p, _i64p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                                        _vp, _c_int, _vp, _vp, _vp]),
    "ptb_accumulate_planned2": (_c_int, [_vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _ip, _c_int, _i64p, _i64p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                                         _vp, _c_int, _vp, _vp, _c_int, _vp]),
    "ptb_merge_div_masked": (_c_int, [_vp, _vp, _vp, _c_int, _c_int, _c_int, _vp, _c_int, _vp]),
    "ptb_norm_accumulate": (_c_int, [_vp, _vp, _i64p, _i64p, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _c_int, _vp]),
    "ptb_debug_plan": (_c_int, [_i64p, _i64p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _c_int, _ip, _c_int]),
    "ptb_merge_div": (_c_int, [_vp, _vp, _vp, _c_int, _c_i64, _vp]),
    "ptb_merge_band": (_c_int, [_vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _vp, _c_int, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ptb_band_plan_create": (_c_i64, [_i64p, _i64p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _i64p, _c_int, _vpp]),
    "ptb_band_plan_upload": (_c_int, [_vp, _vp, _vp]),
    "ptb_band_plan_info": (_c_int, [_vp, _ip, _ip, _i64p, _i64p, _i64p]),
    "ptb_band_plan_reset": (_c_int, [_vp]),
    "ptb_band_plan_state": (_c_int, [_vp, _ip, _ip]),
    "ptb_band_plan_submit": (_c_int, [_vp, _c_int, _c_int, _vp, _c_i64, _c_i64, _c_int, _c_int, _ip, _c_int, _vp, _vp, _vp, _vp]),
    "ptb_band_plan_submit_next": (_c_int, [_vp, _vp, _c_int, _vp]),
    "ptb_band_plan_destroy": (None, [_vp]),
    "ptb_rccl_available": (_c_int, []),
    "ptb_rccl_unique_id": (_c_int, [_vp]),
    "ptb_rccl_comm_init": (_c_int, [_vp, _c_int, _c_int, _vpp]),
    "ptb_rccl_comm_destroy": (_c_int, [_vp]),
    "ptb_halo_exchange": (_c_int, [_vp, _c_int, _vpp, _i64p, _ip, _c_int, _vpp, _i64p, _ip, _vp]),
    "ptb_band_plan_create2": (_c_i64, [_i64p, _i64p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _i64p, _c_int, _i64p, _c_int, _vpp]),
    "ptb_band_plan_create3": (_c_i64, [_i64p, _i64p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _i64p, _c_int, _i64p, _c_int, _c_int,
                                       _vpp]),
    "ptb_band_plan_rows_launched": (_c_int, [_vp, _c_int, _c_int]),
    "ptb_halo_pack": (_c_int, [_vp, _c_i64, _c_i64, _c_int, _c_int, _c_int, _vp, _vp]),
    "ptb_band_plan_finish_rank": (_c_int, [_vp, _vp, _vp, _c_int, _i64p, _vpp, _c_int, _i64p, _vp]),
    "ptb_band_plan_submit_rank": (_c_int, [_vp, _c_int, _c_int, _vp, _c_i64, _c_i64, _c_int, _c_int, _ip, _c_int, _vp, _vp, _vp, _c_int, _i64p, _vpp, _ip, _vp,
                                           _ip, _vp]),
    "ptb_rect_add": (_c_int, [_vp, _vp, _c_int, _c_int, _c_int, _c_i64, _c_i64, _vp]),
    "ptb_merge_div_ex": (_c_int, [_vp, _vp, _vp, _c_int, _c_i64, _c_i64, _c_i64, _vp, _c_i64, _c_i64, _vp]),
    "ptb_deaug_reduce": (_c_int, [_vp, _vp, _c_int, _ip, _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ptb_deaug_reduce_bwd": (_c_int, [_vp, _vp, _vp, _vp, _c_int, _ip, _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ptb_view_transform": (_c_int, [_vp, _vp, _c_int, _ip, _c_int, _c_f, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ptb_view_permute": (_c_int, [_vp, _vp, _c_int, _ip, _c_int, _c_i64, _c_int, _c_int, _c_int, _c_i64, _vp]),
    "ptb_resize_bilinear": (_c_int, [_vp, _vp, _c_i64, _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ptb_ms_deaug_reduce": (_c_int, [_vp, _ip, _ip, _c_int, _vp, _c_i64, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ptb_ms_deaug_reduce_strip": (_c_int, [_vp, _ip, _ip, _ip, _ip, _c_int, _vp, _c_i64, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ptb_resize_bilinear_bwd": (_c_int, [_vp, _vp, _c_i64, _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ptb_ms_deaug_reduce_bwd": (_c_int, [_vp, _ip, _ip, _c_int, _vp, _vp, _vp, _c_i64, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ptb_resize_bicubic": (_c_int, [_vp, _vp, _c_i64, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ptb_bitempered_rows": (_c_int, [_vp, _vp, _vp, _vp, _c_i64, _c_int, _c_f, _c_f, _c_f, _c_int, _c_int, _vp]),
    "ptb_stack_reduce": (_c_int, [_vp, _c_int, _c_i64, _c_int, _c_d, _vp, _vp]),
    "ptb_stack_reduce_bwd": (_c_int, [_vp, _vp, _vp, _c_int, _c_i64, _c_int, _c_d, _vp, _vp]),
    "ptb_resize_nearest": (_c_int, [_vp, _vp, _c_i64, _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ptb_resize_nearest_exact": (_c_int, [_vp, _vp, _c_i64, _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ptb_resize_area": (_c_int, [_vp, _vp, _c_i64, _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ptb_ms_flip_deaug_reduce_strip": (_c_int, [_vp, _ip, _ip, _ip, _ip, _c_int, _c_int, _ip, _c_int, _vp, _c_i64, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ptb_ms_flip_deaug_reduce": (_c_int, [_vp, _ip, _ip, _c_int, _c_int, _ip, _c_int, _vp, _c_i64, _c_int, _c_int, _c_int, _c_int, _vp]),
    "ptb_seg_loss_fwd": (_c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_i64, _c_int, _c_int, _c_f, _c_f, _c_f, _c_i64, _c_f, _vp]),
    "ptb_focal_bwd": (_c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_i64, _c_int, _c_f, _c_f, _c_f, _c_i64, _c_f, _vp]),
    "ptb_focal_softmax_fwd": (_c_int, [_vp, _vp, _vp, _vp, _c_int, _c_int, _c_i64, _vp, _vp, _vp, _c_int, _c_int, _c_i64, _c_int, _c_f, _c_f, _c_f, _c_i64, _c_f, _vp]),
    "ptb_focal_softmax_bwd": (_c_int, [_vp, _vp, _vp, _vp, _c_int, _c_int, _c_i64, _vp, _vp, _vp, _c_int, _c_int, _c_i64, _c_int, _c_f, _c_f, _c_f, _c_i64, _c_f, _vp]),
    "ptb_seg_stats_bwd": (_c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_i64, _c_int, _c_int, _c_i64, _c_f, _vp]),
    "ptb_region_workspace_bytes": (_c_i64, [_c_int]),
    "ptb_region_loss_fwd": (_c_int, [_vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_i64, _c_int, _c_int, _c_f, _c_f, _c_f, _c_i64, _c_f, _c_f, _c_f, _c_f, _c_f,
                                     _c_f, _c_int, _vp, _c_int, _vp, _vp, _vp, _vp]),
    "ptb_region_epilogue": (_c_int, [_vp, _c_int, _c_int, _c_f, _c_f, _c_f, _c_f, _c_f, _c_int, _vp, _c_int, _vp, _vp, _vp, _vp]),
    "ptb_seg_fused_bwd": (_c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_i64, _c_int, _c_int, _c_f, _c_f, _c_f, _c_i64, _c_f, _vp]),
    "ptb_softmax_focal_fwd": (_c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_i64, _c_int, _c_f, _c_f, _c_i64, _vp]),
    "ptb_softmax_focal_bwd": (_c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_i64, _c_int, _c_f, _c_f, _c_i64, _vp]),
    "ptb_lovasz_temp_bytes": (_c_i64, [_c_i64, _c_int]),
    "ptb_lovasz_fwd": (_c_int, [_vp, _vp, _vp, _c_int, _c_int, _c_i64, _c_int, _c_int, _c_int, _c_i64, _c_f] + [_vp] * 9 + [_c_i64, _vp]),
    "ptb_lovasz_fwd_keys": (_c_int, [_vp, _vp, _vp, _c_int, _c_int, _c_i64, _c_int, _c_int, _c_int, _c_i64, _c_f] + [_vp] * 6 + [_c_i64, _vp]),
    "ptb_lovasz_fwd_binned": (_c_int, [_vp, _vp, _vp, _c_int, _c_int, _c_i64, _c_int, _c_int, _c_int, _c_i64, _c_f] + [_vp] * 9 + [_c_i64, _vp]),
    "ptb_lovasz_bwd_binned": (_c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_i64, _c_int, _c_int, _c_int, _c_i64, _c_f, _c_int, _vp]),
    "ptb_lovasz_bwd_binned2": (_c_int, [_vp] * 8 + [_c_int, _c_int, _c_i64, _c_int, _c_int, _c_int, _c_i64, _c_f, _c_int, _vp]),
    "ptb_lovasz_reduce": (_c_int, [_vp, _vp, _c_int, _c_int, _c_int, _vp, _vp, _vp]),
    "ptb_lovasz_bwd": (_c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_i64, _c_int, _c_int, _c_int, _c_i64, _c_f, _vp]),
    "ptb_deaug_accumulate": (_c_int, [_vp, _vp, _vp, _vp, _c_int, _ip, _c_int, _i64p, _i64p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _c_int, _vp]),
    "ptb_split_tiles_u8": (_c_int, [_vp, _c_int, _c_int, _c_int, _i64p, _i64p, _c_int, _c_int, _c_int, _c_int, _ip, _fp, _fp, _c_int, _vp, _vp]),
    "ptb_pointwise_loss_fwd": (_c_int, [_c_int, _vp, _vp, _vp, _vp, _vp, _vp, _c_i64, _c_int, _c_i64, _c_int, _c_f, _c_f, _c_f, _c_f, _vp]),
    "ptb_pointwise_loss_apply": (_c_int, [_c_int, _c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _c_i64, _c_int, _c_i64, _c_int, _c_f, _c_f, _c_f, _c_f, _vp]),
    "ptb_bitempered_binary_fwd": (_c_int, [_vp, _vp, _vp, _vp, _c_i64, _c_f, _c_f, _c_f, _c_int, _c_int, _c_f, _vp]),
    "ptb_bitempered_binary_bwd": (_c_int, [_vp, _vp, _vp, _vp, _vp, _c_i64, _c_f, _c_f, _c_f, _c_int, _c_int, _c_f, _vp]),
    "ptb_soft_ce_fwd": (_c_int, [_vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_i64, _c_f, _c_int, _c_i64, _vp]),
    "ptb_soft_ce_bwd": (_c_int, [_vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_i64, _c_f, _c_int, _c_i64, _vp]),
    "ptb_volume_accumulate": (_c_int, [_vp, _vp, _vp, _vp, _i64p, _i64p, _i64p] + [_c_int] * 8 + [_vp]),
    "ptb_ensemble_reduce": (_c_int, [_vpp, _c_int, _c_int, _c_int, _c_f, _c_int, _c_int, _c_i64, _vp, _vp]),
    "ptb_merge_crop": (_c_int, [_vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp]),
}

_lib = None
_lock = threading.Lock()
fresh_fallbacks = 0  # times a first-touch bitmap could not be honoured (diagnostics)
calls = 0  # number of native entry-point invocations (tests assert the HIP path really ran)


def load():
    """Load libptb_hip.so (built by ``__graft_entry__.build()``); raises ImportError when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise ImportError(
                    f"{LIB_PATH} is missing: build the HIP extension first "
                    "(python -c 'import __graft_entry__ as g; g.build()'). There is no non-HIP fallback."
                )
            lib = ctypes.CDLL(LIB_PATH)  # torch (imported above) has already loaded libamdhip64.so.7
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def check(rc, what):
    if rc == 0:
        return
    msg = _ERR.get(rc, f"error {rc}")
    if rc == -3:
        msg += ": " + load().ptb_last_hip_error().decode()
    if rc == -2:
        raise NotImplementedError(f"{what}: {msg}")
    raise RuntimeError(f"{what}: {msg}")


def require_device(t, what):
    """The native path runs on the GPU only; a CPU tensor that gets this far (entry points without a host form: split_device, the
    sharded mergers' deferred plan, the strip kernels) is refused instead of silently computed elsewhere."""
    if not t.is_cuda:
        raise RuntimeError(
            f"{what}: tensor is on '{t.device}', but pytorch_toolbelt_amd runs its hot path as HIP kernels on the "
            "MI355X only (no CPU fallback). Move the tensor to a 'cuda' device."
        )


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr(device):
    """The current HIP stream of ``device`` as a ``void*`` (the raw-handle query: 0.3 us instead of 4 us for a Stream object)."""
    if _raw_stream is not None:
        idx = device.index
        return _vp(_raw_stream(idx if idx is not None else torch.cuda.current_device()))
    return _vp(torch.cuda.current_stream(device).cuda_stream)


class on_device:
    """Make ``device`` current for the duration of a launch (kernels launch on the HIP current device)."""

    __slots__ = ("dev", "prev")

    def __init__(self, device):
        self.dev = device.index if device.index is not None else torch.cuda.current_device()
        self.prev = None

    def __enter__(self):
        cur = torch.cuda.current_device()
        if cur != self.dev:
            self.prev = cur
            torch.cuda.set_device(self.dev)
        return self

    def __exit__(self, *exc):
        if self.prev is not None:
            torch.cuda.set_device(self.prev)
        return False


def int_array(values):
    return (ctypes.c_int * len(values))(*values)


def i64_array(values):
    return (ctypes.c_int64 * len(values))(*values)


def bump():
    global calls
    calls += 1
