#!/usr/bin/env python3
"""Seeded differential fuzzing on an MI355X (beyond the fixed seeds of tests/):

    python tools/fuzz_gpu.py [first_seed] [count]

1. TileMerger in its three modes (lazy / kernel-maintained normaliser / planned) over random slicer geometries, bit-exact
   against the oracle (the test function of tests/test_tiles_gpu.py with more seeds);
2. fused de-augment merges over random groups x reductions x input dtypes x planned-or-not, 2e-5 relative;
3. the elementwise losses and soft cross entropy over random (odd) shapes and options against the float64 oracle;
4. deferred band merging vs the incremental merger (bit-exact), 5. Dice / Jaccard / fused region losses vs the fp64 oracle;
6. Lovasz losses (value vs the fp64 oracle, gradient vs torch-CPU autograd through the reference's op chain, binned vs scattered
   gradient); 7. the reference's literal calls (lazy handle + self-planning merger) vs the eager unplanned path, bit for bit.
   A third argument selects fuzzers: modes,fused,losses,deferred,region,lovasz,literal.
Prints one line per failure and a summary; exit code 1 on any failure."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pointwise_oracle as PO  # noqa: E402
from oracle import tiles_oracle as TO  # noqa: E402
from oracle import tta_oracle as AO  # noqa: E402
from pytorch_toolbelt_amd import _native as N  # noqa: E402
from pytorch_toolbelt_amd import losses as L  # noqa: E402
from pytorch_toolbelt_amd.inference.tiles import TileMerger  # noqa: E402

dev = torch.device("cuda:0")


def fuzz_merger_modes(seeds):
    import test_tiles_gpu as T

    bad = 0
    for seed in seeds:
        try:
            T.test_random_geometries_all_merger_modes(seed, dev, N)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print("merger modes FAIL seed", seed, repr(e)[:300])
    return bad


def fuzz_fused(seeds):
    bad = 0
    for seed in seeds:
        rng = np.random.default_rng(seed)
        group = ["d4", "d2", "flips", "fliplr", "flipud"][seed % 5]
        red = ["mean", "sum", "gmean", "hmean", "logodd", "log1p", "harmonic1p"][seed % 7]
        V = {"d4": 8, "d2": 4, "flips": 3, "fliplr": 2, "flipud": 2}[group]
        if seed % 3:
            th = int(rng.choice([64, 128, 192]))
            tw = th if group == "d4" else int(rng.choice([64, 128, 256]))
            step = (min(int(rng.choice([32, 64, th])), th), min(int(rng.choice([64, tw])), tw))
        else:
            th = int(rng.integers(8, 60))
            tw = th if group == "d4" else int(rng.integers(8, 60))
            step = (int(rng.integers(max(1, th // 2), th + 1)), int(rng.integers(max(1, tw // 2), tw + 1)))
        shape = (int(rng.integers(th, 3 * th + 30)), int(rng.integers(tw, 3 * tw + 30)))
        C, batch = int(rng.integers(1, 4)), int(rng.integers(1, 9))
        geom = TO.slicer_geometry(shape, (th, tw), step)
        crops, n = geom["crops"], len(geom["crops"])
        w = TO.pyramid_window(th, tw)[0]
        positive = red not in ("mean", "sum")
        x = (rng.random((n, V, C, th, tw)) * 0.9 + 0.05 if positive else rng.standard_normal((n, V, C, th, tw))).astype(np.float32)
        dt = [torch.float32, torch.float16, torch.bfloat16][(seed // 5) % 3]
        xt = torch.from_numpy(x).to(dev).to(dt)
        xq = xt.float().cpu().numpy()
        st = TO.merger_new(geom["target_shape"], C, w)
        m = TileMerger(geom["target_shape"], C, w, device=dev, crops=crops if seed % 2 else None)
        try:
            for b0 in range(0, n, batch):
                sel = slice(b0, min(n, b0 + batch))
                nb = sel.stop - sel.start
                xb = np.ascontiguousarray(np.moveaxis(xq[sel], 1, 0)).reshape(V * nb, C, th, tw)
                TO.merger_integrate(st, AO.image_deaugment(xb, group, red), crops[sel])
                m.integrate_batch_deaugment(xt[sel].transpose(0, 1).reshape(V * nb, C, th, tw).contiguous(), crops[sel], group=group, reduction=red)
            got, want = m.merge().cpu().numpy(), TO.merger_merge(st)
            err = np.nanmax(np.abs(got - want) / (1 + np.abs(want)))
            if not (err <= 2e-5) or (np.isnan(got) != np.isnan(want)).any():
                bad += 1
                print("fused FAIL", seed, group, red, dt, shape, (th, tw), step, C, batch, err)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print("fused ERROR", seed, group, red, dt, repr(e)[:200])
    return bad


def fuzz_losses(seeds):
    def close(a, b, tol=3e-5):
        a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
        return a.shape == b.shape and bool(np.all(np.abs(a - b) <= tol * (1 + np.abs(b))))

    bad = 0
    for seed in seeds:
        rng = np.random.default_rng(seed)
        B, C, H, W = int(rng.integers(1, 4)), int(rng.integers(1, 22)), int(rng.integers(1, 40)), int(rng.integers(1, 40))
        if seed % 4 == 0:
            W = 4 * int(rng.integers(1, 12))
        x = (rng.standard_normal((B, C, H, W)) * 2).astype(np.float32)
        t = (rng.random((B, C, H, W)) < 0.4).astype(np.float32)
        ti = t.copy()
        ti[rng.random((B, C, H, W)) < 0.1] = -100
        labi = rng.integers(0, C, (B, H, W))
        labi[rng.random((B, H, W)) < 0.15] = -100
        xd, td, tid = (torch.from_numpy(v).to(dev) for v in (x, t, ti))
        red = ["mean", "sum", "none"][seed % 3]
        try:
            wv, pwv = (rng.random(C) + 0.5).astype(np.float32), (rng.random(C) * 2 + 0.5).astype(np.float32)
            sf = [None, 0.1][seed % 2]
            ok = close(L.SoftBCEWithLogitsLoss(weight=torch.from_numpy(wv).view(C, 1, 1).to(dev), pos_weight=torch.from_numpy(pwv).view(C, 1, 1).to(dev),
                                               reduction=red, smooth_factor=sf)(xd, tid).cpu().numpy(),
                       PO.soft_bce(x, ti, wv.reshape(C, 1, 1), pwv.reshape(C, 1, 1), -100, red, sf))
            g = 1.0 + seed % 3 * 0.5
            ok &= close(L.balanced_binary_cross_entropy_with_logits(xd, tid, gamma=g, ignore_index=-100, reduction=red).cpu().numpy(),
                        PO.balanced_bce(x, ti, g, -100, red))
            qred, beta = ["mean", "sum", "none", "normalized"][seed % 4], [2.0, 1.0, 1.5][seed % 3]
            ok &= close(L.QualityFocalLoss(beta=beta, reduction=qred)(xd, td).cpu().numpy(), PO.quality_focal(x, t, beta, qred))
            ok &= close(L.functional.wing_loss(xd, td, 2.0, 0.7, red).cpu().numpy(), PO.wing(x, t, 2.0, 0.7, red))
            ok &= close(float(L.functional.log_cosh_loss(xd, td)), PO.log_cosh(x, t))
            eps = 0.1 * (seed % 3)
            ok &= close(L.SoftCrossEntropyLoss(reduction=red, smooth_factor=eps)(xd, torch.from_numpy(labi).to(dev)).cpu().numpy(),
                        PO.soft_ce(x, labi, eps, -100, red))
            if not ok:
                bad += 1
                print("losses FAIL", seed, (B, C, H, W), red)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print("losses ERROR", seed, (B, C, H, W), repr(e)[:300])
    from pytorch_toolbelt_amd.losses import _kernels as K

    K.flush_label_check()
    return bad


def fuzz_deferred(seeds):
    """Deferred band merging vs the incremental merger, bit for bit, over random block-aligned geometries."""
    import test_tiles_gpu as T

    bad = 0
    for seed in seeds:
        rng = np.random.default_rng(seed)
        th = int(rng.choice([64, 128, 192, 256]))
        tw = th if seed % 2 else int(rng.choice([64, 128, 256]))
        sy = int(rng.choice([v for v in (32, 64, 96, 128, 192, 256) if v <= th]))
        sx = int(rng.choice([v for v in (64, 128, 192, 256) if v <= tw]))
        shape = (int(rng.integers(th, 4 * th)), int(rng.integers(tw, 4 * tw)))
        groups = ["d4", "d2", "flips", "fliplr", "flipud"] if th == tw else ["d2", "flips", "fliplr", "flipud"]
        group = groups[seed % len(groups)]
        red = ["mean", "sum", "gmean", "hmean"][seed % 4]
        dtype = [torch.float32, torch.float32, torch.float16, torch.bfloat16][(seed // 3) % 4]
        try:
            T._deferred_case(dev, shape, (th, tw), (sy, sx), int(rng.integers(1, 5)), group, red, dtype, int(rng.integers(1, 10)), images=2, seed=seed)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print("deferred FAIL seed", seed, shape, (th, tw), (sy, sx), group, red, dtype, repr(e)[:300])
    return bad


def fuzz_region_losses(seeds):
    """Dice / Jaccard / fused focal+Dice+Jaccard (straight-line and generic kernels, epilogue kernel) vs the fp64 oracle."""
    from oracle import losses_oracle as LO

    bad = 0
    for seed in seeds:
        rng = np.random.default_rng(seed)
        C = int(rng.integers(2, 20))
        B = int(rng.integers(1, 4))
        H, W = (int(rng.choice([16, 32, 48])), int(rng.choice([16, 32, 64]))) if seed % 3 else (int(rng.integers(5, 40)), int(rng.integers(5, 40)))
        x = (rng.standard_normal((B, C, H, W)) * rng.choice([1.0, 3.0, 8.0])).astype(np.float32)
        lab = rng.integers(0, C, (B, H, W))
        ign = [None, 0, 255][seed % 3]
        if ign == 255:
            lab[0, : H // 3] = 255
        kw = dict(log_loss=bool(seed % 2), smooth=float(rng.choice([0.0, 1.0])))
        xt, lt = torch.from_numpy(x).to(dev), torch.from_numpy(lab).to(dev)
        try:
            got = float(L.DiceLoss("multiclass", ignore_index=ign, **kw)(xt, lt))
            want = LO.dice_loss(x, lab, "multiclass", ignore_index=ign, **kw)
            assert abs(got - want) <= 2e-5 * (1 + abs(want)), ("dice", got, want)
            if ign is None:
                got = float(L.JaccardLoss("multiclass", **kw)(xt, lt))
                want = LO.jaccard_loss(x, lab, "multiclass", **kw)
                assert abs(got - want) <= 2e-5 * (1 + abs(want)), ("jaccard", got, want)
                xg = xt.clone().requires_grad_(True)
                fused = L.FocalDiceJaccardLoss("multiclass", **kw)(xg, lt)
                want = LO.binary_focal_loss(x, lab) + LO.dice_loss(x, lab, "multiclass", **kw) + LO.jaccard_loss(x, lab, "multiclass", **kw)
                assert abs(float(fused) - want) <= 2e-5 * (1 + abs(want)), ("fused", float(fused), want)
                fused.backward()
                x2 = xt.clone().requires_grad_(True)
                parts = L.BinaryFocalLoss()(x2, lt) + L.DiceLoss("multiclass", **kw)(x2, lt) + L.JaccardLoss("multiclass", **kw)(x2, lt)
                parts.backward()
                assert torch.allclose(xg.grad, x2.grad, rtol=2e-4, atol=1e-8), ("fused grad", float((xg.grad - x2.grad).abs().max()))
            # dense (multilabel) targets: streaming statistics kernels forward and backward vs the oracle and the generic kernels
            tdense = (rng.random((B, C, H, W)) < 0.3).astype(np.float32)
            ignd = [None, 255][seed % 2]
            if ignd is not None:
                tdense[0, :, : H // 4] = 255.0
            tt = torch.from_numpy(tdense).to(dev)
            got = float(L.DiceLoss("multilabel", ignore_index=ignd, **kw)(xt, tt))
            want = LO.dice_loss(x, tdense, "multilabel", ignore_index=ignd, **kw)
            assert abs(got - want) <= 2e-5 * (1 + abs(want)), ("dice multilabel", got, want)
            grads = []
            for scalar in (0, 1):
                N.load().ptb_set_tunable(1, scalar)
                try:
                    xg = xt.clone().requires_grad_(True)
                    (L.JaccardLoss("multilabel", **kw)(xg, tt) + L.FocalDiceJaccardLoss("multilabel", ignore_index=ignd)(xg, tt)).backward()
                    grads.append(xg.grad)
                finally:
                    N.load().ptb_set_tunable(1, 0)
            assert torch.allclose(grads[0], grads[1], rtol=2e-4, atol=1e-8), ("multilabel grad", float((grads[0] - grads[1]).abs().max()))
        except Exception as e:  # noqa: BLE001
            bad += 1
            print("region loss FAIL seed", seed, (B, C, H, W), ign, kw, repr(e)[:300])
    return bad


def fuzz_lovasz(seeds):
    """LovaszLoss / BinaryLovaszLoss over random shapes and options: the value against the fp64 oracle, the gradient against torch-CPU
    autograd through the reference's op chain (oracle/torch_chain.py), binned against scattered gradient bit for bit."""
    from oracle import losses_oracle as LO
    from oracle import torch_chain as TC
    from pytorch_toolbelt_amd.losses import lovasz as LV

    bad = 0
    for seed in seeds:
        rng = np.random.default_rng(seed)
        B, C = int(rng.integers(1, 4)), int(rng.integers(2, 9))
        H, W = (int(rng.integers(1, 30)), int(rng.integers(1, 30))) if seed % 3 else (int(rng.integers(60, 100)), int(rng.integers(60, 100)))
        per_image, ign = bool(seed % 2), [None, 255][(seed // 2) % 2]
        classes = ["present", "all"][(seed // 4) % 2]
        logits = (rng.standard_normal((B, C, H, W)) * rng.choice([1.0, 4.0])).astype(np.float32)
        probs = torch.softmax(torch.from_numpy(logits), 1)
        lab = rng.integers(0, C, (B, H, W))
        lab[lab == C - 1] = 0
        if ign is not None:
            lab[rng.random((B, H, W)) < 0.1] = 255
        labt = torch.from_numpy(lab)
        try:
            want = LO.lovasz_softmax(probs.numpy(), lab, classes=classes, per_image=per_image, ignore_index=ign)
            grads = {}
            for binned in (True, False):
                prev, LV.BINNED_GRADIENT = LV.BINNED_GRADIENT, binned
                try:
                    xg = probs.to(dev).requires_grad_(True)
                    got = LV._lovasz_softmax(xg, labt.to(dev), classes=classes, per_image=per_image, ignore_index=ign)
                    got.backward()
                    grads[binned] = xg.grad
                finally:
                    LV.BINNED_GRADIENT = prev
            assert abs(float(got.detach()) - float(want)) <= 1e-5 * (1 + abs(float(want))), ("value", float(got.detach()), float(want))
            assert torch.equal(grads[True], grads[False]), "binned gradient differs from the scattered one"
            if classes == "present":      # (the op chain of the reference skips absent classes)
                xc = probs.clone().requires_grad_(True)
                groups = [(xc[b:b + 1], labt[b:b + 1]) for b in range(B)] if per_image else [(xc, labt)]
                total, ties = 0.0, False
                for pg, lg in groups:
                    flat = pg.movedim(1, -1).reshape(-1, C)
                    ll = lg.reshape(-1)
                    if ign is not None:
                        keep = ll != ign
                        flat, ll = flat[keep], ll[keep]
                    if flat.numel():
                        total = total + TC.lovasz_softmax(flat.t().reshape(1, C, -1, 1), ll.reshape(1, -1, 1))
                        with torch.no_grad():      # equal errors inside a class: which of the tied pixels gets which gradient is the sort's choice
                            for c in range(C):
                                err = ((ll == c).float() - flat[:, c]).abs()
                                ties = ties or err.unique().numel() < err.numel()
                if torch.is_tensor(total) and not ties:
                    (total / len(groups)).backward()
                    # (a gradient element is J_k - J_{k-1} of two fp32 Jaccard values near 0.5: a few ulps of those, up to ~6e-7, in absolute terms)
                    assert torch.allclose(grads[True].cpu(), xc.grad, rtol=1e-3, atol=1e-6), ("gradient", float((grads[True].cpu() - xc.grad).abs().max()))
            x = (rng.standard_normal((B, H, W)) * 2).astype(np.float32)
            y = (rng.random((B, H, W)) < 0.4).astype(np.float32)
            if ign is not None:
                y[rng.random((B, H, W)) < 0.1] = 255.0
            wantb = LO.lovasz_hinge(x, y, per_image=per_image, ignore_index=ign)
            gotb = LV._lovasz_hinge(torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev), per_image=per_image, ignore_index=ign)
            assert abs(float(gotb) - float(wantb)) <= 1e-5 * (1 + abs(float(wantb))), ("hinge", float(gotb), float(wantb))
        except Exception as e:  # noqa: BLE001
            bad += 1
            print("lovasz FAIL seed", seed, (B, C, H, W), per_image, ign, classes, repr(e)[:300])
    return bad


def fuzz_literal(seeds):
    """The reference's literal calls (lazy de-augment handle into a self-planning TileMerger, a new merger per image) over random
    geometries: bit for bit against the eager unplanned path, three images per geometry (the plan is used from the second on)."""
    import test_dropin_gpu as D
    from pytorch_toolbelt_amd.inference import _lazy, tiles

    bad = 0
    prev_l, prev_a = _lazy.set_enabled(True), tiles.set_auto_plan(True)
    try:
        for seed in seeds:
            rng = np.random.default_rng(seed)
            group = ["d4", "d2", "flips", "fliplr", "flipud"][seed % 5]
            red = ["mean", "sum", "gmean", "hmean"][seed % 4]
            th = int(rng.choice([32, 64, 96, 128]))
            tw = th if group == "d4" else int(rng.choice([32, 64, 128]))
            step = (int(rng.choice([v for v in (16, 32, 48, 64, 128) if v <= th])), int(rng.choice([v for v in (16, 32, 64, 128) if v <= tw])))
            shape = (int(rng.integers(th, 3 * th + 20)), int(rng.integers(tw, 3 * tw + 20)))
            C, batch = int(rng.integers(1, 4)), int(rng.integers(1, 9))
            V = D.GROUPS[group]
            geom = TO.slicer_geometry(shape, (th, tw), step)
            crops, n = geom["crops"], len(geom["crops"])
            w = TO.pyramid_window(th, tw)[0]
            try:
                for image in range(3):
                    g = torch.Generator(device="cpu").manual_seed(seed * 10 + image)
                    outputs = (torch.rand((V * n, C, th, tw), generator=g) * 0.9 + 0.05).to(dev)
                    _lazy.set_enabled(True)
                    tiles.set_auto_plan(True)
                    got = D._run_image(TileMerger(geom["target_shape"], C, w, device=dev), outputs, crops, batch, True, group, red)
                    _lazy.set_enabled(False)
                    tiles.set_auto_plan(False)
                    want = D._run_image(TileMerger(geom["target_shape"], C, w, device=dev), outputs, crops, batch, True, group, red)
                    same = torch.equal(torch.nan_to_num(got, nan=-7.0), torch.nan_to_num(want, nan=-7.0))
                    assert same, ("image", image, float((got - want).abs().nan_to_num().max()))
            except Exception as e:  # noqa: BLE001
                bad += 1
                print("literal FAIL seed", seed, group, red, shape, (th, tw), step, C, batch, repr(e)[:300])
    finally:
        _lazy.set_enabled(prev_l)
        tiles.set_auto_plan(prev_a)
    return bad


def fuzz_views(seeds):
    """ptb_view_permute / ptb_view_transform through inference/_views.view_transform against torch's own flip / transpose chains
    (inference/_host.py on the same CUDA tensor): random dtype, rank 4-6, plane sizes around the kernel's tile edges, random view
    lists (all transposing or none on non-square planes), augment and chunk-wise forms, sliced (offset) inputs.  torch.equal."""
    from pytorch_toolbelt_amd.inference import _host
    from pytorch_toolbelt_amd.inference import _views as V

    dts = [torch.uint8, torch.int8, torch.int16, torch.int32, torch.int64, torch.bool, torch.float16, torch.bfloat16, torch.float32, torch.float64,
           torch.complex64, torch.complex128]
    bad = 0
    for seed in seeds:
        rng = np.random.default_rng(seed)
        dt = dts[int(rng.integers(0, len(dts)))]
        H = int(rng.choice([1, 3, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 200, 257]))
        square = rng.random() < 0.5
        W = H if square else int(rng.choice([1, 4, 17, 64, 96, 128, 130, 255, 300]))
        rest = tuple(int(v) for v in rng.integers(1, 4, size=int(rng.integers(0, 3))))
        nv = int(rng.integers(1, 9))
        if H == W:
            views = [int(v) for v in rng.integers(0, 8, size=nv)]
        else:
            t = int(rng.integers(0, 2))
            views = [int(v) * 2 + t for v in rng.integers(0, 4, size=nv)]
        in_is_batch = bool(rng.integers(0, 2))
        B, C = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        n = B if in_is_batch else B * nv
        try:
            g = torch.Generator().manual_seed(int(seed))
            shape = (n + 1, C, H, W) + rest
            if dt == torch.bool:
                x = torch.rand(shape, generator=g) < 0.5
            elif dt.is_complex:
                x = torch.complex(torch.randn(shape, generator=g, dtype=torch.float64), torch.randn(shape, generator=g, dtype=torch.float64)).to(dt)
            elif dt.is_floating_point:
                x = torch.randn(shape, generator=g, dtype=torch.float64).to(dt)
            else:
                info = torch.iinfo(dt)
                x = torch.randint(max(info.min, -2**40), min(info.max, 2**40), shape, generator=g, dtype=torch.int64).to(dt)
            x = x.to(dev)[1:]                      # a storage offset: not every base is 16-byte aligned
            got = V.view_transform(x, views, in_is_batch=in_is_batch)
            want = _host.view_transform(x, views, in_is_batch)
            assert got.dtype == want.dtype and got.shape == want.shape and torch.equal(got, want.contiguous()), "mismatch"
        except Exception as e:  # noqa: BLE001
            bad += 1
            print("views FAIL seed", seed, dt, (n, C, H, W) + rest, views, in_is_batch, repr(e)[:300])
    return bad


def fuzz_self_deferred(seeds):
    """Self-planning mergers under callers that do NOT repeat themselves (round 5: mergers without crops= plan themselves into deferred
    bands by default).  Per seed a random geometry and a stream of images through new TileMerger(shape, C, weight) objects, every image
    with a random twist -- none, tiles skipped, tiles in another order from a random point on, an extra tile at the end, merge() early,
    a read of .image / .norm_mask in the middle, another batch size, a static output buffer, fp16 outputs -- against the plain merger
    (auto_plan=False) fed the same tiles in the same order: bit-identical while nothing forces a self-deferred merger to degrade
    mid-image, within 1e-6 relative + the same NaN pattern when it does; never an exception except the one refusal (a model that
    switches to a static buffer after its geometry was learnt with fresh outputs)."""
    import warnings

    import test_dropin_gpu as D
    from pytorch_toolbelt_amd.inference import _lazy, tiles, tta

    bad = 0
    stats = {"images": 0, "started deferred": 0, "started planned": 0, "degraded (not bit-identical)": 0, "refused static": 0}
    prev_l, prev_a = _lazy.set_enabled(True), tiles.set_auto_plan(True)
    os.environ["PTB_DEFER_ROWS"] = "128"
    try:
        for seed in seeds:
            rng = np.random.default_rng(seed)
            tiles._auto.clear()
            th = int(rng.choice([64, 128]))
            step = int(rng.choice([v for v in (32, 64, 128) if v <= th]))
            shape = (int(rng.integers(2 * th, 6 * th)), int(rng.integers(th, 4 * th)))
            C, batch = int(rng.integers(1, 4)), int(rng.integers(1, 9))
            geom = TO.slicer_geometry(shape, (th, th), (step, step))
            crops, n = geom["crops"], len(geom["crops"])
            w = TO.pyramid_window(th, th)[0]
            static = torch.empty((8 * 8, C, th, th), device=dev)
            try:
                for image in range(10):
                    g = torch.Generator(device="cpu").manual_seed(seed * 16 + image)
                    outputs = (torch.rand((8 * n, C, th, th), generator=g) * 0.9 + 0.05).to(dev)
                    # even images repeat the regular loop (the geometry is forgotten first: a twisted image makes it ask for more evidence),
                    # odd images bring a twist to a merger that -- mostly -- starts in deferred bands
                    if image % 2 == 0:
                        tiles._auto.clear()
                    twist = ["skip", "reorder", "extra", "early", "peek", "batch", "static", "half", "late-reorder", "late-early"][int(rng.integers(0, 10))] if image % 2 else "none"
                    order = np.arange(n)
                    b = batch
                    if twist == "skip":
                        order = order[rng.random(n) > 0.15]
                    elif twist == "reorder":
                        cut = int(rng.integers(0, n))
                        order = np.concatenate([order[:cut], rng.permutation(order[cut:])])
                    elif twist == "extra":
                        order = np.concatenate([order, order[int(rng.integers(0, n)):][:1]])
                    elif twist == "early":
                        order = order[:int(rng.integers(1, n + 1))]
                    elif twist == "late-reorder":
                        cut = int(rng.integers(n // 2, n))
                        order = np.concatenate([order[:cut], order[cut:][::-1]])
                    elif twist == "late-early":
                        order = order[:int(rng.integers(n // 2, n + 1))]
                    elif twist == "batch":
                        b = int(rng.integers(1, 9))
                    peek_at = int(rng.integers(0, len(order))) if twist == "peek" else -1
                    mergers = {"self": tiles.TileMerger(geom["target_shape"], C, w, device=dev), "plain": tiles.TileMerger(geom["target_shape"], C, w, device=dev, auto_plan=False)}
                    started = mergers["self"].mode
                    stats["images"] += 1
                    stats["started deferred"] += started == "deferred bands"
                    stats["started planned"] += started == "planned"
                    peeked, refused = {}, False
                    with warnings.catch_warnings():
                        warnings.simplefilter("ignore")
                        for name, m in mergers.items():
                            for b0 in range(0, len(order), b):
                                sel = order[b0:b0 + b]
                                y = torch.cat([outputs[k * n + sel] for k in range(8)])
                                if twist == "half":
                                    y = y.half()
                                if twist == "static":
                                    buf = static[:8 * len(sel)]
                                    buf.copy_(y)
                                    y = buf
                                try:
                                    m.integrate_batch(tta.d4_image_deaugment(y), crops[sel])
                                except RuntimeError as e:
                                    if name == "self" and twist == "static" and started == "deferred bands" and "occupies memory" in str(e):
                                        refused = True
                                        break
                                    raise
                                if b0 <= peek_at < b0 + b:
                                    peeked[name] = (m.image.clone(), m.norm_mask.clone())
                            if refused:
                                break
                        if refused:
                            stats["refused static"] += 1
                            continue
                        got, want = mergers["self"].merge(), mergers["plain"].merge()
                    gn, wn = got.float().cpu().numpy(), want.float().cpu().numpy()
                    assert np.array_equal(np.isnan(gn), np.isnan(wn)), ("nan pattern", image, twist, started)
                    exact = np.array_equal(np.nan_to_num(gn, nan=-7.0), np.nan_to_num(wn, nan=-7.0))
                    if not exact:
                        stats["degraded (not bit-identical)"] += 1
                        assert started == "deferred bands" and twist in ("skip", "reorder", "extra", "early", "peek", "late-reorder", "late-early"), ("not bit-identical", image, twist, started)
                        np.testing.assert_allclose(np.nan_to_num(gn), np.nan_to_num(wn), rtol=1e-6, atol=1e-6)
                    if peeked:
                        np.testing.assert_allclose(peeked["self"][0].cpu().numpy(), peeked["plain"][0].cpu().numpy(), rtol=1e-6, atol=1e-6)
                        assert torch.equal(peeked["self"][1], peeked["plain"][1])
            except Exception as e:  # noqa: BLE001
                bad += 1
                print("self-deferred FAIL seed", seed, shape, th, step, C, batch, repr(e)[:400])
    finally:
        os.environ.pop("PTB_DEFER_ROWS", None)
        _lazy.set_enabled(prev_l)
        tiles.set_auto_plan(prev_a)
        tiles._auto.clear()
    print("self-deferred:", stats)
    return bad


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    seeds = range(first, first + count)
    only = set(sys.argv[3].split(",")) if len(sys.argv) > 3 else None
    fuzzers = {"modes": fuzz_merger_modes, "fused": fuzz_fused, "losses": fuzz_losses, "deferred": fuzz_deferred, "region": fuzz_region_losses,
               "lovasz": fuzz_lovasz, "literal": fuzz_literal, "selfdeferred": fuzz_self_deferred, "views": fuzz_views}
    run = [f for k, f in fuzzers.items() if only is None or k in only]
    total = sum(f(seeds) for f in run)
    print(f"fuzz: {len(run) * count} cases, {total} failures")
    sys.exit(1 if total else 0)
