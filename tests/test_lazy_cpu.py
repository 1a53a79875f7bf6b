"""CPU: the protocol of the lazy de-augmentation handle (inference/_lazy.py) and the held-batch guard of the deferred mergers
(inference/tiles.py:_check_held) -- the parts of the drop-in path that are plain Python and need no GPU.  The kernels behind them are
covered in tests/test_dropin_gpu.py."""
import copy
import gc
import pickle

import numpy as np
import pytest
import torch

from pytorch_toolbelt_amd.inference import _lazy as L


def _mean_views(src, views, code):          # stands in for ptb_deaug_reduce: chunk-major [V*B, ...] -> mean over the views
    V = len(views)
    return src.view(V, src.shape[0] // V, *src.shape[1:]).mean(0)


def _handle(src=None):
    src = torch.randn(8, 2, 4, 4) if src is None else src
    return L.LazyDeaugment(src, "fliplr", (0, 4), 1, _mean_views), src


def test_metadata_does_not_evaluate_and_everything_else_does():
    h, src = _handle()
    before = L.evaluations
    assert isinstance(h, torch.Tensor) and type(h) is L.LazyDeaugment
    assert h.shape == (4, 2, 4, 4) and h.dtype == torch.float32 and h.device.type == "cpu" and len(h) == 4 and h.dim() == 4
    assert h.numel() == 128 and h.size(1) == 2 and not h.requires_grad and h.is_contiguous() and not h.is_cuda
    assert L.evaluations == before and h._value is None
    want = _mean_views(src, (0, 4), 1)
    assert torch.equal(h + 1, want + 1) and L.evaluations == before + 1        # evaluated once ...
    assert torch.equal(h * 2, want * 2) and torch.equal(h[1], want[1]) and L.evaluations == before + 1   # ... and cached
    assert torch.equal(torch.cat([h, h]), torch.cat([want, want])) and torch.equal(torch.stack((h,)), torch.stack((want,)))
    assert torch.equal(torch.add(input=h, other=h), want * 2)
    assert np.array_equal(np.asarray(h), want.numpy()) and np.array_equal(h.numpy(), want.numpy())
    assert h.to(torch.float64).dtype == torch.float64 and float(h.sum()) == pytest.approx(float(want.sum()))
    assert torch.equal(torch.nn.functional.interpolate(h, scale_factor=2), torch.nn.functional.interpolate(want, scale_factor=2))
    assert "tensor(" in repr(h) and h.data_ptr() == h._evaluate().data_ptr()
    assert type(copy.deepcopy(h)) is torch.Tensor and torch.equal(pickle.loads(pickle.dumps(h)), want)


def test_in_place_operations_act_on_the_cached_value():
    h, src = _handle()
    want = _mean_views(src, (0, 4), 1)
    h.mul_(3.0)
    h.add_(1.0)
    assert torch.equal(h, want * 3 + 1)
    h[0] = 0.0
    assert float(h[0].abs().max()) == 0.0


def test_take_source_is_what_a_merger_fuses_and_the_handle_stays_usable():
    h, src = _handle()
    taken = h._take_source()
    assert taken is not None and taken[0] is src and taken[1] == "fliplr" and taken[2] == (0, 4) and taken[3] == 1
    assert torch.equal(h, _mean_views(src, (0, 4), 1))      # evaluated later from the same source
    assert h._take_source() is None                         # ... after which there is nothing left to fuse


def test_an_edited_source_is_reported_not_used():
    h, src = _handle()
    src.add_(1.0)
    with pytest.raises(RuntimeError, match="modified in place"):
        h + 0
    with pytest.raises(RuntimeError, match="modified in place"):
        h._take_source()
    with torch.inference_mode():                            # tensors without a version counter are taken as they are
        s2 = torch.randn(8, 2, 4, 4)
        h2 = L.LazyDeaugment(s2, "fliplr", (0, 4), 1, _mean_views)
        assert torch.equal(h2 * 1, _mean_views(s2, (0, 4), 1))


def test_pending_sources_are_bounded():
    old = L._BUDGET
    try:
        src = torch.randn(8, 2, 4, 4)
        L._BUDGET = L._pending_bytes + 3 * src.numel() * 4
        hs = [L.LazyDeaugment(src, "fliplr", (0, 4), 1, _mean_views) for _ in range(6)]
        assert [x._value is not None for x in hs] == [True, True, True, False, False, False]   # the oldest were evaluated to make room
        pending = L._pending_bytes
        del hs
        gc.collect()
        assert L._pending_bytes == pending - 3 * src.numel() * 4
    finally:
        L._BUDGET = old


def test_maybe_lazy_only_takes_inference_shaped_gpu_calls():
    x = torch.randn(8, 2, 4, 4)
    assert L.maybe_lazy(x, "fliplr", (0, 4), 1, _mean_views) is None            # CPU tensor: evaluated on the spot (and refused downstream)
    prev = L.set_enabled(False)
    try:
        assert not L.enabled()
    finally:
        L.set_enabled(prev)


def test_held_batch_guard():
    from pytorch_toolbelt_amd.inference.tiles import _check_held, _held_entry

    pool = torch.zeros(6, 2, 4, 4)
    a, b = pool[0:4], pool[2:6]
    held = [(a, "coords", None, 0, 0) + _held_entry(a)]
    with pytest.raises(RuntimeError, match="occupies memory of an earlier batch"):
        _check_held(held, b, _held_entry(b), False, "TileMerger(defer=True)")
    with pytest.raises(RuntimeError, match="occupies memory"):               # the same tensor again: a refilled static buffer looks like this
        _check_held(held, a, _held_entry(a), False, "TileMerger(defer=True)")
    c = torch.zeros(4, 2, 4, 4)
    _check_held(held, c, _held_entry(c), True, "TileMerger(defer=True)")      # disjoint memory, nothing edited: fine
    a.mul_(2.0)
    _check_held(held, c, _held_entry(c), False, "TileMerger(defer=True)")     # versions are only compared when a launch is due ...
    with pytest.raises(RuntimeError, match="held batch 0 .* modified in place"):
        _check_held(held, c, _held_entry(c), True, "TileMerger(defer=True)")
    with torch.inference_mode():
        d = torch.zeros(4, 2, 4, 4)
        entry = _held_entry(d)
        assert entry[2] is None                                               # no version counter: accepted, never compared
        d.add_(1.0)
        _check_held([(d,) + entry], c, _held_entry(c), True, "x")


def test_held_batches_custody_is_one_class_for_both_deferred_mergers():
    """``HeldBatches`` (inference/_merge_modes.py) is the custody list of ``TileMerger(defer=True)`` and of the sharded merger's
    deferred band: admit = the contract check, keep / release_before / take_all = the bookkeeping both used to do by hand."""
    from pytorch_toolbelt_amd.inference._merge_modes import HeldBatches

    h = HeldBatches("X(defer=True)")
    pool = torch.zeros(6, 2, 4, 4)
    a, b, c = pool[0:2], pool[2:4], pool[1:3]
    h.keep(a, h.admit(a, False), ("crops a", None, 0), last_group=0)
    h.keep(b, h.admit(b, True), ("crops b", None, 0), last_group=1)
    assert len(h) == 2 and bool(h) and [r[0] is t for r, t in zip(h, (a, b))] == [True, True]
    with pytest.raises(RuntimeError, match=r"X\(defer=True\): this batch occupies memory"):
        h.admit(c, False)
    h.release_before(1)                       # group 0 is out: a goes, b (read by group 1) stays
    assert len(h) == 1 and next(iter(h))[0] is b and next(iter(h))[1] == ("crops b", None, 0)
    b.add_(1.0)
    with pytest.raises(RuntimeError, match="modified in place"):
        h.admit(torch.zeros(1), True)
    rows = h.take_all()
    assert len(rows) == 1 and len(h) == 0 and not h
    h.keep(a, h.admit(a, True))
    h.clear()
    assert len(h) == 0


def test_autograd_on_an_evaluated_handle_and_the_legacy_dlpack_guard():
    """Once evaluated, requires_grad / grad / is_leaf are those of the value (they were answered by the storage-less wrapper:
    `y.requires_grad_(); ...backward(); y.grad` gave None); torch.utils.dlpack.to_dlpack -- a bare C function that would read the
    wrapper's null storage -- evaluates a handle first and is untouched for ordinary tensors."""
    import threading

    import torch.utils.dlpack as D

    h, src = _handle()
    want = _mean_views(src, (0, 4), 1)
    assert not h.requires_grad and h.grad is None and h._value is None          # metadata of an unevaluated handle: no evaluation
    h.requires_grad_(True)
    (h * 2).sum().backward()
    assert h.requires_grad and h.is_leaf and torch.equal(h.grad, torch.full_like(want, 2.0))
    assert getattr(D.to_dlpack, "_ptb_lazy_guard", False)
    h2, src2 = _handle()
    assert torch.equal(D.from_dlpack(D.to_dlpack(h2)), _mean_views(src2, (0, 4), 1))
    plain = torch.arange(6.0)
    assert torch.equal(D.from_dlpack(D.to_dlpack(plain)), plain)
    assert torch.equal(torch.from_dlpack(_handle(src2)[0]), _mean_views(src2, (0, 4), 1))
    # two threads using one handle end up with ONE value
    h3, _ = _handle()
    got = []
    ts = [threading.Thread(target=lambda: got.append(h3._evaluate())) for _ in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert all(g is got[0] for g in got)


def test_importing_the_package_leaves_torch_alone_and_the_guard_comes_and_goes():
    """The to_dlpack guard is installed by the FIRST handle, not by `import pytorch_toolbelt_amd.inference`; switching the handles off
    (tta.set_lazy_deaugment(False) / pytorch_toolbelt_amd.set_strict_dropin()) takes it out again; strict drop-in mode also stops
    mergers from planning themselves and routes half-precision accumulators of CUDA mergers to the torch-op merger."""
    import subprocess
    import sys

    code = ("import torch, torch.utils.dlpack as D; orig = D.to_dlpack\n"
            "import pytorch_toolbelt_amd, pytorch_toolbelt_amd.inference, pytorch_toolbelt_amd.losses\n"
            "assert D.to_dlpack is orig and torch.to_dlpack is getattr(torch, 'to_dlpack')\n"
            "from pytorch_toolbelt_amd.inference import _lazy as L, tiles as T\n"
            "h = L.LazyDeaugment(torch.randn(4, 1, 2, 2), 'fliplr', (0, 4), 1, lambda s, v, c: s.view(2, 2, 1, 2, 2).mean(0))\n"
            "assert D.to_dlpack is not orig and D.to_dlpack._ptb_lazy_guard\n"
            "prev = pytorch_toolbelt_amd.set_strict_dropin(True)\n"
            "assert prev == (True, True, False) and D.to_dlpack is orig and not L.enabled() and not T._AUTO_PLAN\n"
            "T.set_auto_plan(True); assert pytorch_toolbelt_amd.set_strict_dropin(True) == (False, True, True) and not T._AUTO_PLAN\n"
            "assert T._REFERENCE_ACCUMULATORS and type(T.TileMerger.__new__(T.TileMerger, (8, 8), 1, None, 'cuda', dtype=torch.float16)) is T.HostBackedTileMerger\n"
            "assert type(T.TileMerger.__new__(T.TileMerger, (8, 8), 1, None, 'cuda')) is T.TileMerger\n"
            "assert pytorch_toolbelt_amd.set_strict_dropin(False) == (False, False, True) and L.enabled() and T._AUTO_PLAN\n"
            "assert not T._REFERENCE_ACCUMULATORS and type(T.TileMerger.__new__(T.TileMerger, (8, 8), 1, None, 'cuda', dtype=torch.float16)) is T.TileMerger\n"
            # flag=False restores what was in force BEFORE strict mode, not hard-coded defaults, and is a no-op without a preceding True
            "T.set_auto_plan(False); T.set_reference_accumulators(True)\n"
            "assert pytorch_toolbelt_amd.set_strict_dropin(False) == (True, False, True) and not T._AUTO_PLAN and T._REFERENCE_ACCUMULATORS\n"
            "pytorch_toolbelt_amd.set_strict_dropin(True); pytorch_toolbelt_amd.set_strict_dropin(False)\n"
            "assert L.enabled() and not T._AUTO_PLAN and T._REFERENCE_ACCUMULATORS\n"
            "print('OK')\n")
    import os

    env = dict(os.environ)
    env.pop("PTB_AUTO_PLAN", None)
    env.pop("PTB_LAZY_DEAUG", None)
    env.pop("PTB_REFERENCE_ACCUMULATORS", None)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr


def test_refcount_ownership_probe_checks_itself_at_import():
    """`owned` (a de-augment argument that is a temporary of the call expression) rests on sys.getrefcount; the module
    verifies at import that a temporary reads as owned and a name-bound / list-held / keyword-passed object does not, and switches the test off
    (nothing is ever 'owned': version-less tensors are evaluated on the spot) when that fails."""
    from pytorch_toolbelt_amd.inference import tta

    assert tta._refcount_probe_works() and tta._TEMP_REFS > 0
    named = object()
    temp, bound = tta._probe_owned(object()), tta._probe_owned(named)       # (not inside an `assert`: pytest's rewriting keeps intermediate values alive)
    assert temp and not bound
    prev = tta._TEMP_REFS
    try:
        tta._TEMP_REFS = -1          # what a failed self-check leaves behind
        temp = tta._probe_owned(object())
        assert not temp
    finally:
        tta._TEMP_REFS = prev
