#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the UNMODIFIED reference (/root/reference) on seeded inputs.

Test infrastructure only.  Run in the build container (the reference does not exist on the GPU box):

    python oracle/make_golden.py

The reference is imported behind oracle/ref_shim (a cv2 / torchvision stand-in: the reference's
utils/__init__ imports both, neither wheel is installed here).  No reference source is copied;
only its inputs/outputs are stored.  Every fixture is a flat npz: arrays plus a ``__cases__`` JSON string
listing (name, function, kwargs, input keys, output key) so the tests can replay each case.
"""
import hashlib
import json
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("PTB_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(HERE, "ref_shim"))
sys.path.insert(0, REF)
warnings.filterwarnings("ignore")

import torch  # noqa: E402

from pytorch_toolbelt.inference import tiles as rt  # noqa: E402
from pytorch_toolbelt.inference import tta as rtta  # noqa: E402
from pytorch_toolbelt.inference import functional as rfn  # noqa: E402
from pytorch_toolbelt import losses as rl  # noqa: E402
from pytorch_toolbelt.losses import functional as rlf  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def save(name, arrays, cases):
    arrays = dict(arrays)
    arrays["__cases__"] = np.array(json.dumps(cases))
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print(f"{name}: {len(cases)} cases, {os.path.getsize(path) / 1024:.1f} KiB")


def t2n(t):
    return t.detach().cpu().numpy()


# ----------------------------------------------------------------------------------------------- tiles
def gen_tiles():
    A, cases = {}, []
    geoms = [
        dict(image_shape=[1024, 1024, 3], tile_size=256, tile_step=128),            # BASELINE cfg1
        dict(image_shape=[5000, 5000, 3], tile_size=[512, 512], tile_step=[256, 256]),  # cfg2/3
        dict(image_shape=[500, 500, 3], tile_size=51, tile_step=26),                # ref tests/test_tiles.py:13-18
        dict(image_shape=[563, 512, 3], tile_size=[128, 128], tile_step=[128, 128]),  # ref tests/test_tiles.py:21-26
        dict(image_shape=[5632, 5120, 3], tile_size=[1280, 1280], tile_step=[1280, 1280]),
        dict(image_shape=[75, 83, 3], tile_size=[32, 24], tile_step=[16, 12]),
        dict(image_shape=[40, 40], tile_size=64, tile_step=32),                     # image smaller than a tile
        dict(image_shape=[100, 130, 1], tile_size=[48, 64], tile_step=[48, 32], image_margin=8),
        dict(image_shape=[100, 130, 1], tile_size=[48, 64], tile_step=[24, 64], image_margin=[3, 5, 7, 9]),
        dict(image_shape=[4096, 4096, 3], tile_size=512, tile_step=256),
        dict(image_shape=[97, 61, 2], tile_size=[17, 13], tile_step=[5, 13]),
    ]
    for k, g in enumerate(geoms):
        s = rt.ImageSlicer(g["image_shape"], g["tile_size"], g["tile_step"], image_margin=g.get("image_margin", 0))
        A[f"geom{k}_crops"] = s.crops.astype(np.int64)
        A[f"geom{k}_bbox"] = s.bbox_crops.astype(np.int64)
        A[f"geom{k}_meta"] = np.array(
            [s.margin_left, s.margin_right, s.margin_top, s.margin_bottom, *s.target_shape, *s.tile_size, *s.tile_step],
            dtype=np.int64,
        )
        cases.append(dict(name=f"geom{k}", fn="geometry", kwargs=g))

    # pyramid windows: small ones in full, the 512x512 one by digest + probes
    for (w, h) in [(32, 24), (51, 51), (17, 13), (64, 64)]:
        W, Dc, De = rt.compute_pyramid_patch_weight_loss(w, h)
        A[f"pyr_{w}x{h}_W"], A[f"pyr_{w}x{h}_Dc"], A[f"pyr_{w}x{h}_De"] = W, Dc, De
        cases.append(dict(name=f"pyr_{w}x{h}", fn="pyramid", kwargs=dict(width=w, height=h)))
    for (w, h) in [(512, 512), (256, 256), (1280, 1280)]:
        W, _, _ = rt.compute_pyramid_patch_weight_loss(w, h)
        A[f"pyr_{w}x{h}_sha256"] = np.array(hashlib.sha256(np.ascontiguousarray(W).tobytes()).hexdigest())
        A[f"pyr_{w}x{h}_stats"] = np.array([W.min(), W.max(), W.sum(), W[3, 5], W[h // 2, w // 3]])
        cases.append(dict(name=f"pyr_{w}x{h}", fn="pyramid_digest", kwargs=dict(width=w, height=h)))

    # split / cut_patch / iter_split / merge (fp64 numpy) on a non-zero image
    rng = np.random.default_rng(0)
    for k, (shape, ts, st, wname) in enumerate(
        [((75, 83, 3), (32, 24), (16, 12), "pyramid"), ((64, 50, 2), (32, 32), (16, 16), "mean"), ((50, 47, 1), (16, 16), (16, 16), "mean")]
    ):
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        s = rt.ImageSlicer(img.shape, ts, st, weight=wname)
        tl = s.split(img)
        A[f"split{k}_image"] = img
        A[f"split{k}_tiles"] = np.stack(tl)
        A[f"split{k}_cut"] = np.stack([s.cut_patch(img, i) for i in range(len(s.crops))])
        it = list(s.iter_split(img))
        A[f"split{k}_iter_tiles"] = np.stack([t for t, _ in it])
        A[f"split{k}_iter_coords"] = np.stack([c for _, c in it]).astype(np.int64)
        A[f"split{k}_merge_f32"] = s.merge(tl, dtype=np.float32)
        A[f"split{k}_merge_u8"] = s.merge(tl, dtype=np.uint8)
        ftiles = [rng.standard_normal(t.shape) for t in tl]
        A[f"split{k}_ftiles"] = np.stack(ftiles)
        A[f"split{k}_fmerge"] = s.merge(ftiles, dtype=np.float32)
        cases.append(dict(name=f"split{k}", fn="split_merge", kwargs=dict(image_shape=list(shape), tile_size=list(ts), tile_step=list(st), weight=wname)))

    # TileMerger (torch CPU fp32): integrate in batches, then merge
    g = torch.Generator().manual_seed(0)
    for k, (shape, ts, st, wname, C, bs) in enumerate(
        [((96, 80, 3), (32, 32), (16, 16), "pyramid", 3, 4), ((75, 83, 3), (32, 24), (16, 12), "pyramid", 2, 5),
         ((64, 64, 3), (32, 32), (32, 32), "mean", 1, 3), ((50, 70, 3), (20, 28), (7, 9), "pyramid", 2, 8)]
    ):
        s = rt.ImageSlicer(shape, ts, st, weight=wname)
        n = len(s.crops)
        pred = torch.randn((n, C, ts[0], ts[1]), generator=g)
        m = rt.TileMerger(s.target_shape, C, s.weight)
        for b0 in range(0, n, bs):
            m.integrate_batch(pred[b0:b0 + bs], s.crops[b0:b0 + bs])
        A[f"merger{k}_pred"] = t2n(pred)
        A[f"merger{k}_image"] = t2n(m.image)
        A[f"merger{k}_norm"] = t2n(m.norm_mask)
        A[f"merger{k}_merged"] = t2n(m.merge())
        cases.append(dict(name=f"merger{k}", fn="tile_merger", kwargs=dict(image_shape=list(shape), tile_size=list(ts), tile_step=list(st), weight=wname, channels=C, batch=bs)))
    save("tiles.npz", A, cases)


# ----------------------------------------------------------------------------------------------- tta
def gen_tta():
    A, cases = {}, []
    g = torch.Generator().manual_seed(1)
    x = torch.rand((2, 3, 16, 16), generator=g)
    xr = torch.rand((2, 3, 12, 20), generator=g)           # non-square for flip / d2 / flips
    A["x_sq"], A["x_rect"] = t2n(x), t2n(xr)
    groups = dict(fliplr=2, flipud=2, flips=3, d2=4, d4=8)
    for grp, nv in groups.items():
        for tag, src in (("sq", x), ("rect", xr)):
            if grp == "d4" and tag == "rect":
                continue
            aug = getattr(rtta, f"{grp}_image_augment")(src)
            A[f"aug_{grp}_{tag}"] = t2n(aug)
            cases.append(dict(name=f"aug_{grp}_{tag}", fn="image_augment", kwargs=dict(group=grp), inputs=[f"x_{tag}"], output=f"aug_{grp}_{tag}"))
        for tag, shape in (("sq", (nv * 2, 3, 16, 16)), ("rect", (nv * 2, 3, 12, 20))):
            if grp == "d4" and tag == "rect":
                continue
            y = torch.rand(shape, generator=g) * 0.98 + 0.01
            A[f"y_{grp}_{tag}"] = t2n(y)
            for red in ["mean", "sum", "gmean", "hmean", "harmonic1p", "logodd", "log1p", None]:
                out = getattr(rtta, f"{grp}_image_deaugment")(y, reduction=red)
                key = f"deaug_{grp}_{tag}_{red}"
                A[key] = t2n(out)
                cases.append(dict(name=key, fn="image_deaugment", kwargs=dict(group=grp, reduction=red), inputs=[f"y_{grp}_{tag}"], output=key))
    # gmean on zeros / negatives (reference: 0 -> 0, negative -> NaN), hmean clamp
    y = torch.rand((16, 1, 8, 8), generator=g)
    y[0, 0, 0, 0] = 0.0
    y[3, 0, 1, 1] = -0.5
    y[5, 0, 2, 2] = 1e-9
    A["y_d4_edge"] = t2n(y)
    for red in ["gmean", "hmean", "logodd", "log1p", "harmonic1p"]:
        key = f"deaug_d4_edge_{red}"
        A[key] = t2n(rtta.d4_image_deaugment(y, reduction=red))
        cases.append(dict(name=key, fn="image_deaugment", kwargs=dict(group="d4", reduction=red), inputs=["y_d4_edge"], output=key))

    # label variants (incl. the b7,b7 quirk of d4_labels_deaugment)
    for grp, nv in dict(groups, fivecrop=5).items():
        lg = torch.rand((nv * 3, 5), generator=g)
        A[f"labels_{grp}"] = t2n(lg)
        fn = rtta.fivecrop_label_deaugment if grp == "fivecrop" else getattr(rtta, f"{grp}_labels_deaugment")
        for red in ["mean", "sum", "gmean", None]:
            key = f"labdeaug_{grp}_{red}"
            A[key] = t2n(fn(lg, reduction=red))
            cases.append(dict(name=key, fn="labels_deaugment", kwargs=dict(group=grp, reduction=red), inputs=[f"labels_{grp}"], output=key))
    A["fivecrop_aug"] = t2n(rtta.fivecrop_image_augment(x, (8, 10)))
    cases.append(dict(name="fivecrop_aug", fn="fivecrop_image_augment", kwargs=dict(crop_size=[8, 10]), inputs=["x_sq"], output="fivecrop_aug"))

    # multiscale
    xm = torch.rand((1, 2, 24, 20), generator=g)
    A["x_ms"] = t2n(xm)
    offsets = [-8, 0, 6, [4, -4]]
    for ac in (False, True):
        augs = rtta.ms_image_augment(xm, offsets, mode="bilinear", align_corners=ac)
        for i, a in enumerate(augs):
            A[f"ms_aug_ac{int(ac)}_{i}"] = t2n(a)
        cases.append(dict(name=f"ms_aug_ac{int(ac)}", fn="ms_image_augment", kwargs=dict(size_offsets=offsets, align_corners=ac), inputs=["x_ms"], output=[f"ms_aug_ac{int(ac)}_{i}" for i in range(len(offsets))]))
    fmaps = [torch.rand((1, 2, 24 + (o[0] if isinstance(o, list) else o), 20 + (o[1] if isinstance(o, list) else o)), generator=g) * 0.9 + 0.05 for o in offsets]
    for i, f in enumerate(fmaps):
        A[f"ms_fm_{i}"] = t2n(f)
    for ac in (False, True):
        for red in ("mean", "gmean"):
            key = f"ms_deaug_ac{int(ac)}_{red}"
            A[key] = t2n(rtta.ms_image_deaugment(fmaps, offsets, reduction=red, mode="bilinear", align_corners=ac))
            cases.append(dict(name=key, fn="ms_image_deaugment", kwargs=dict(size_offsets=offsets, reduction=red, align_corners=ac, stride=1), inputs=[f"ms_fm_{i}" for i in range(len(offsets))], output=key))
    # stride 2 with negative non-multiple offsets (floor-division quirk Q3)
    offs2 = [-5, 0, 7]
    fm2 = [torch.rand((1, 2, 12 + o // 2 + (0 if o % 2 == 0 or o > 0 else 0), 10 + o // 2), generator=g) for o in offs2]
    # feature maps of a stride-2 model: size = (input + offset) // 2 is the caller's business; any sizes are legal inputs
    for i, f in enumerate(fm2):
        A[f"ms2_fm_{i}"] = t2n(f)
    key = "ms_deaug_stride2"
    A[key] = t2n(rtta.ms_image_deaugment(fm2, offs2, reduction="mean", mode="bilinear", align_corners=True, stride=2))
    cases.append(dict(name=key, fn="ms_image_deaugment", kwargs=dict(size_offsets=offs2, reduction="mean", align_corners=True, stride=2), inputs=[f"ms2_fm_{i}" for i in range(3)], output=key))

    # reductions on a [T, ...] stack directly (inference/functional.py:250-333)
    st = torch.rand((5, 7, 9), generator=g)
    A["red_stack"] = t2n(st)
    for nm, fn in [("geometric_mean", rfn.geometric_mean), ("harmonic_mean", rfn.harmonic_mean), ("harmonic1p_mean", rfn.harmonic1p_mean), ("logodd_mean", rfn.logodd_mean), ("log1p_mean", rfn.log1p_mean)]:
        A[f"red_{nm}"] = t2n(fn(st, dim=0))
        cases.append(dict(name=f"red_{nm}", fn="reduction", kwargs=dict(which=nm), inputs=["red_stack"], output=f"red_{nm}"))
    save("tta.npz", A, cases)


# ----------------------------------------------------------------------------------------------- losses
def gen_losses():
    A, cases = {}, []
    g = torch.Generator().manual_seed(2)
    B, C, H, W = 3, 5, 12, 10
    logits = torch.randn((B, C, H, W), generator=g) * 3.0
    # non-iid across images so per_image Lovasz differs from the batch version
    logits[1] += 1.5
    logits[2] *= 0.3
    labels = torch.randint(0, C, (B, H, W), generator=g)
    labels[0, :3] = 2
    labels_ign = labels.clone()
    labels_ign[torch.rand((B, H, W), generator=g) < 0.15] = 255
    onehot = torch.nn.functional.one_hot(labels, C).permute(0, 3, 1, 2).float()
    multilabel = (torch.rand((B, C, H, W), generator=g) < 0.3).float()
    multilabel[:, 4] = 0  # an empty class -> its loss term is zeroed
    multilabel_ign = multilabel.clone()
    multilabel_ign[torch.rand((B, C, H, W), generator=g) < 0.1] = 255
    cw = torch.tensor([0.5, 1.0, 2.0, 1.5, 0.25])
    bin_logits = torch.randn((B, 1, H, W), generator=g) * 2.0
    bin_logits[1] -= 1.0
    bin_t = (torch.rand((B, 1, H, W), generator=g) < 0.4).float()
    bin_t_ign = bin_t.clone()
    bin_t_ign[torch.rand((B, 1, H, W), generator=g) < 0.1] = 255
    for k, v in dict(logits=logits, labels=labels, labels_ign=labels_ign, onehot=onehot, multilabel=multilabel,
                     multilabel_ign=multilabel_ign, class_weights=cw, bin_logits=bin_logits, bin_t=bin_t, bin_t_ign=bin_t_ign).items():
        A[k] = t2n(v)

    def add(name, fn, kwargs, inputs, value):
        A[name] = t2n(value) if torch.is_tensor(value) else np.asarray(value)
        cases.append(dict(name=name, fn=fn, kwargs=kwargs, inputs=inputs, output=name))

    # focal_loss_with_logits option matrix
    focal_opts = [
        dict(), dict(alpha=None), dict(gamma=1.5, alpha=0.6), dict(reduction="sum"), dict(reduction="batchwise_mean"),
        dict(reduction="none"), dict(normalized=True), dict(reduced_threshold=0.5), dict(reduced_threshold=0.3, normalized=True),
        dict(gamma=0.0, alpha=None), dict(gamma=3.0),
    ]
    for i, kw in enumerate(focal_opts):
        add(f"focal_fn_{i}", "focal_loss_with_logits", kw, ["logits", "onehot"], rlf.focal_loss_with_logits(logits, onehot, **kw))
    add("focal_fn_cw", "focal_loss_with_logits", dict(class_weights=True), ["logits", "onehot"], rlf.focal_loss_with_logits(logits, onehot, class_weights=cw))
    add("focal_fn_softmax", "focal_loss_with_logits", dict(activation="softmax", softmax_dim=1), ["logits", "onehot"], rlf.focal_loss_with_logits(logits, onehot, activation="softmax", softmax_dim=1))
    add("focal_fn_soft_targets", "focal_loss_with_logits", dict(alpha=0.3), ["logits", "multilabel"], rlf.focal_loss_with_logits(logits, multilabel * 0.7 + 0.1, alpha=0.3))
    A["soft_targets"] = t2n(multilabel * 0.7 + 0.1)
    cases[-1]["inputs"] = ["logits", "soft_targets"]
    add("focal_fn_ign", "focal_loss_with_logits", dict(ignore_index=255, normalized=True), ["logits", "multilabel_ign"], rlf.focal_loss_with_logits(logits, multilabel_ign, ignore_index=255, normalized=True))

    # BinaryFocalLoss module (label targets -> one-hot, ignore handling)
    bf_opts = [dict(), dict(alpha=0.25), dict(ignore_index=255), dict(ignore_index=255, normalized=True, alpha=0.4),
               dict(reduction="sum", gamma=1.0), dict(reduced_threshold=0.5, reduction="sum"), dict(class_weights=True)]
    for i, kw in enumerate(bf_opts):
        k2 = dict(kw)
        if k2.pop("class_weights", None):
            k2["class_weights"] = cw
        lab = labels_ign if kw.get("ignore_index") is not None else labels
        add(f"binary_focal_{i}", "binary_focal_loss", kw, ["logits", "labels_ign" if kw.get("ignore_index") is not None else "labels"], rl.BinaryFocalLoss(**k2)(logits, lab))
    add("binary_focal_same_shape", "binary_focal_loss", dict(alpha=0.5), ["bin_logits", "bin_t"], rl.BinaryFocalLoss(alpha=0.5)(bin_logits, bin_t))

    # CrossEntropyFocalLoss / softmax_focal_loss_with_logits
    sf_opts = [dict(), dict(gamma=1.0), dict(reduction="sum"), dict(reduction="batchwise_mean"), dict(reduction="none"),
               dict(normalized=True), dict(reduced_threshold=0.5), dict(ignore_index=255), dict(ignore_index=255, class_weights=True, gamma=1.5)]
    for i, kw in enumerate(sf_opts):
        k2 = dict(kw)
        if k2.pop("class_weights", None):
            k2["class_weights"] = cw
        lab = labels_ign if kw.get("ignore_index") is not None else labels
        add(f"softmax_focal_{i}", "softmax_focal_loss_with_logits", kw, ["logits", "labels_ign" if kw.get("ignore_index") is not None else "labels"], rlf.softmax_focal_loss_with_logits(logits, lab, **k2))

    # soft scores
    probs = torch.softmax(logits, dim=1)
    A["probs"] = t2n(probs)
    for nm, fn in (("soft_dice_score", rlf.soft_dice_score), ("soft_jaccard_score", rlf.soft_jaccard_score)):
        add(f"{nm}_all", nm, dict(), ["probs", "onehot"], fn(probs, onehot))
        add(f"{nm}_dims", nm, dict(smooth=1.0, dims=[0, 2, 3]), ["probs", "onehot"], fn(probs, onehot, smooth=1.0, dims=(0, 2, 3)))

    # Dice / Jaccard modules
    for cls_name, cls in (("dice_loss", rl.DiceLoss), ("jaccard_loss", rl.JaccardLoss)):
        opts = [
            (dict(mode="multiclass"), "logits", "labels"),
            (dict(mode="multiclass", log_loss=True, smooth=1.0), "logits", "labels"),
            (dict(mode="multiclass", from_logits=False), "probs", "labels"),
            (dict(mode="multilabel"), "logits", "multilabel"),
            (dict(mode="multilabel", smooth=0.5, log_loss=True), "logits", "multilabel"),
            (dict(mode="binary"), "bin_logits", "bin_t"),
            (dict(mode="binary", log_loss=True), "bin_logits", "bin_t"),
            (dict(mode="multiclass", classes=[0, 2, 3]), "logits", "labels"),
        ]
        if cls_name == "dice_loss":
            opts += [
                (dict(mode="multiclass", ignore_index=255), "logits", "labels_ign"),
                (dict(mode="multilabel", ignore_index=255), "logits", "multilabel_ign"),
                (dict(mode="binary", ignore_index=255), "bin_logits", "bin_t_ign"),
            ]
        for i, (kw, a, b) in enumerate(opts):
            k2 = dict(kw)
            if "classes" in k2:
                # the reference's list handling is broken (np.ndarray(list) -> NaN); a tensor works
                k2["classes"] = torch.tensor(k2["classes"])
            val = cls(**k2)(torch.from_numpy(A[a]), torch.from_numpy(A[b]))
            add(f"{cls_name}_{i}", cls_name, kw, [a, b], val)

    # Lovasz (the reference has no test for it at all)
    for i, kw in enumerate([dict(), dict(per_image=True), dict(ignore=255), dict(per_image=True, ignore=255)]):
        lab = labels_ign if "ignore" in kw else labels
        add(f"lovasz_softmax_{i}", "lovasz_softmax", kw, ["probs", "labels_ign" if "ignore" in kw else "labels"], rl.LovaszLoss(**kw)(probs, lab))
    bl = bin_logits[:, 0].contiguous()
    bt = bin_t[:, 0].contiguous()
    bti = bin_t_ign[:, 0].contiguous()
    A["bl"], A["bt"], A["bti"] = t2n(bl), t2n(bt), t2n(bti)
    for i, kw in enumerate([dict(), dict(per_image=True), dict(ignore_index=255), dict(per_image=True, ignore_index=255)]):
        add(f"lovasz_hinge_{i}", "lovasz_hinge", kw, ["bl", "bti" if "ignore_index" in kw else "bt"], rl.BinaryLovaszLoss(**kw)(bl, bti if "ignore_index" in kw else bt))

    # gradients of the training losses w.r.t. logits (autograd of the reference)
    def grad_of(fn):
        x = logits.clone().requires_grad_(True)
        fn(x).backward()
        return x.grad

    add("grad_binary_focal", "grad_binary_focal", dict(alpha=0.25), ["logits", "labels"], grad_of(lambda x: rl.BinaryFocalLoss(alpha=0.25)(x, labels)))
    add("grad_binary_focal_norm_ign", "grad_binary_focal", dict(ignore_index=255, normalized=True), ["logits", "labels_ign"], grad_of(lambda x: rl.BinaryFocalLoss(ignore_index=255, normalized=True)(x, labels_ign)))
    add("grad_softmax_focal", "grad_softmax_focal", dict(), ["logits", "labels"], grad_of(lambda x: rl.CrossEntropyFocalLoss()(x, labels)))
    add("grad_dice_mc", "grad_dice", dict(mode="multiclass"), ["logits", "labels"], grad_of(lambda x: rl.DiceLoss("multiclass")(x, labels)))
    add("grad_dice_ml_log", "grad_dice", dict(mode="multilabel", log_loss=True, smooth=1.0), ["logits", "multilabel"], grad_of(lambda x: rl.DiceLoss("multilabel", log_loss=True, smooth=1.0)(x, multilabel)))
    add("grad_jaccard_mc", "grad_jaccard", dict(mode="multiclass"), ["logits", "labels"], grad_of(lambda x: rl.JaccardLoss("multiclass")(x, labels)))
    add("grad_jaccard_ml", "grad_jaccard", dict(mode="multilabel"), ["logits", "multilabel"], grad_of(lambda x: rl.JaccardLoss("multilabel")(x, multilabel)))
    save("losses.npz", A, cases)


# ----------------------------------------------------------------------------------------------- loop edges (8f-1)
def gen_edges():
    """The compositions at both ends of the README loop (README.md:196-227), run with the reference's own functions."""
    from pytorch_toolbelt.utils import torch_utils as rtu

    A, cases = {}, []
    rng = np.random.default_rng(7)
    aug_fns = {"d4": rtta.d4_image_augment, "d2": rtta.d2_image_augment, "fliplr": rtta.fliplr_image_augment,
               "flipud": rtta.flipud_image_augment, "flips": rtta.flips_image_augment, None: lambda t: t}
    front = [
        dict(image_shape=[75, 83, 3], tile_size=[32, 32], tile_step=[16, 16], augment="d4"),
        dict(image_shape=[64, 50, 2], tile_size=[32, 24], tile_step=[16, 12], augment="d2", affine=True),
        dict(image_shape=[50, 47], tile_size=[16, 16], tile_step=[16, 16], augment="fliplr"),
        dict(image_shape=[100, 130, 1], tile_size=[48, 64], tile_step=[24, 64], image_margin=[3, 5, 7, 9], augment="flips", affine=True, value=7),
        dict(image_shape=[40, 40, 3], tile_size=64, tile_step=32, augment="d4"),
        dict(image_shape=[97, 61, 4], tile_size=[17, 13], tile_step=[5, 13], augment="flipud", indices=[3, 4, 5, 40, 2]),
        dict(image_shape=[97, 61, 4], tile_size=[17, 13], tile_step=[5, 13], augment=None),
        dict(image_shape=[45, 45, 3], tile_size=[15, 15], tile_step=[10, 10], augment="d4", affine=True),
    ]
    for k, g in enumerate(front):
        img = rng.integers(0, 256, g["image_shape"], dtype=np.uint8)
        s = rt.ImageSlicer(img.shape, g["tile_size"], g["tile_step"], image_margin=g.get("image_margin", 0))
        tiles = [rtu.tensor_from_rgb_image(t) for t in s.split(img, value=g.get("value", 0))]      # README.md:209
        idx = g.get("indices") or list(range(len(tiles)))
        batch = torch.stack([tiles[i] for i in idx]).float()                                       # README.md:216
        C = batch.shape[1]
        if g.get("affine"):
            scale = (1.0 / (255.0 * rng.uniform(0.2, 0.3, C))).astype(np.float32)
            bias = (-rng.uniform(0.4, 0.5, C) / rng.uniform(0.2, 0.3, C)).astype(np.float32)
            batch = batch * torch.from_numpy(scale).view(1, C, 1, 1) + torch.from_numpy(bias).view(1, C, 1, 1)
            A[f"front{k}_scale"], A[f"front{k}_bias"] = scale, bias
        out = aug_fns[g["augment"]](batch)
        A[f"front{k}_image"] = img
        A[f"front{k}_out"] = t2n(out)
        cases.append(dict(name=f"front{k}", fn="tiles_to_batch", kwargs=g))

    gt = torch.Generator().manual_seed(11)
    back = [
        dict(image_shape=[96, 80, 3], tile_size=[32, 32], tile_step=[16, 16], weight="pyramid", channels=3, batch=4),
        dict(image_shape=[75, 83, 3], tile_size=[32, 24], tile_step=[16, 12], weight="pyramid", channels=2, batch=5),
        dict(image_shape=[50, 70, 3], tile_size=[20, 28], tile_step=[7, 9], weight="pyramid", channels=4, batch=8),
        dict(image_shape=[61, 33, 3], tile_size=[16, 16], tile_step=[16, 16], weight="mean", channels=1, batch=3),
    ]
    for k, g in enumerate(back):
        s = rt.ImageSlicer(g["image_shape"], g["tile_size"], g["tile_step"], weight=g["weight"])
        n, C = len(s.crops), g["channels"]
        pred = torch.rand((n, C, *s.tile_size), generator=gt) * 255.0
        m = rt.TileMerger(s.target_shape, C, s.weight)
        for b0 in range(0, n, g["batch"]):
            m.integrate_batch(pred[b0:b0 + g["batch"]], s.crops[b0:b0 + g["batch"]])
        merged = m.merge()
        hwc = np.moveaxis(rtu.to_numpy(merged), 0, -1)                                             # README.md:225
        A[f"back{k}_pred"] = t2n(pred)
        A[f"back{k}_hwc_f32"] = s.crop_to_orignal_size(hwc)
        A[f"back{k}_hwc_u8"] = s.crop_to_orignal_size(hwc.astype(np.uint8))                        # README.md:225-226
        A[f"back{k}_argmax"] = s.crop_to_orignal_size(rtu.to_numpy(merged.argmax(dim=0)))
        cases.append(dict(name=f"back{k}", fn="merge_crop", kwargs=g))
    save("edges.npz", A, cases)


# ----------------------------------------------------------------------------------------------- ensembling (8f-2)
class _Affine(torch.nn.Module):
    """Deterministic stand-in model: every output is ``x * k + b`` (so the tests can rebuild it without weights)."""

    def __init__(self, k, b, kind):
        super().__init__()
        self.k, self.b, self.kind = k, b, kind

    def forward(self, x):
        y = x * self.k + self.b
        if self.kind == "tensor":
            return y
        if self.kind == "list":
            return [y, y * 0.5 - 0.25]
        return {"logits": y, "aux": y * 0.5 - 0.25}


def gen_ensembling():
    from pytorch_toolbelt.inference import ensembling as re_

    A, cases = {}, []
    g = torch.Generator().manual_seed(5)
    x = torch.randn((2, 5, 12, 16), generator=g)
    xpos = torch.rand((2, 5, 12, 16), generator=g) * 0.9 + 0.05
    A["x"], A["xpos"] = t2n(x), t2n(xpos)
    coeffs = [(1.0, 0.0), (0.7, 0.3), (1.3, -0.2), (0.5, 0.1), (0.9, -0.4)]
    k = 0
    for kind in ("tensor", "dict", "list"):
        for wrap in (None, "sigmoid", "softmax"):
            if wrap is not None and kind == "tensor":
                continue
            for reduction in ("mean", "sum", "gmean", "hmean", "harmonic1p", "logodd", "log1p"):
                nm = 3 if reduction in ("sum", "logodd") else 5
                positive = wrap is None and reduction not in ("mean", "sum")
                inp = xpos if positive else x
                cs = [(1.0, 0.0), (0.9, 0.02), (0.8, 0.05), (0.95, 0.01), (0.85, 0.03)] if positive else coeffs
                models = [_Affine(a, b, kind) for a, b in cs[:nm]]
                temp = 1.0 if k % 2 else 0.5
                if wrap == "sigmoid":
                    models = [re_.ApplySigmoidTo(m, output_key=("logits" if kind == "dict" else 0), temperature=temp) for m in models]
                elif wrap == "softmax":
                    models = [re_.ApplySoftmaxTo(m, output_key=("logits" if kind == "dict" else 0), dim=1, temperature=temp) for m in models]
                outputs = ["logits"] if (kind == "dict" and k % 3 == 0) else None
                ens = re_.Ensembler(models, reduction=reduction, outputs=outputs)
                out = ens(inp)
                name = f"ens{k}"
                if kind == "tensor":
                    A[f"{name}_out"] = t2n(out)
                    keys = None
                elif kind == "dict":
                    keys = list(out.keys())
                    for kk in keys:
                        A[f"{name}_out_{kk}"] = t2n(out[kk])
                else:
                    keys = list(range(len(out)))
                    for kk in keys:
                        A[f"{name}_out_{kk}"] = t2n(out[kk])
                cases.append(dict(name=name, fn="ensembler", kwargs=dict(kind=kind, wrap=wrap, reduction=reduction, coeffs=cs[:nm], temperature=temp,
                                                                         input="xpos" if positive else "x", outputs=outputs, keys=keys)))
                k += 1
    save("ensembling.npz", A, cases)


# ----------------------------------------------------------------------------------------------- more losses (8f-3)
def gen_losses2():
    """SoftBCE / balanced BCE / QualityFocal / wing / log-cosh / SoftCrossEntropy: values and input gradients."""
    A, cases = {}, []
    g = torch.Generator().manual_seed(9)
    B, C, H, W = 3, 5, 12, 10
    logits = torch.randn((B, C, H, W), generator=g) * 3.0
    logits[1] += 1.0
    hard = (torch.rand((B, C, H, W), generator=g) < 0.3).float()
    hard_ign = hard.clone()
    hard_ign[torch.rand((B, C, H, W), generator=g) < 0.1] = -100.0
    soft = torch.rand((B, C, H, W), generator=g)
    labels = torch.randint(0, C, (B, H, W), generator=g)
    labels_ign = labels.clone()
    labels_ign[torch.rand((B, H, W), generator=g) < 0.15] = -100
    reg_pred = torch.randn((4, 6, 7), generator=g) * 4.0
    reg_true = torch.randn((4, 6, 7), generator=g) * 4.0
    wvec = torch.tensor([0.5, 1.0, 2.0, 1.5, 0.25])
    pwvec = torch.tensor([1.0, 3.0, 0.5, 2.0, 1.0])
    odd_logits = torch.randn((2, 3, 7, 5), generator=g) * 2.0          # HW % 4 != 0 -> scalar kernels
    odd_hard = (torch.rand((2, 3, 7, 5), generator=g) < 0.4).float()
    odd_labels = torch.randint(0, 3, (2, 7, 5), generator=g)
    big_logits = torch.randn((2, 20, 6, 6), generator=g) * 2.0         # C > 16 -> generic soft-CE kernel
    big_labels = torch.randint(0, 20, (2, 6, 6), generator=g)
    flat_logits = torch.randn((9, 7), generator=g) * 2.0               # [N, C] classification input
    flat_labels = torch.randint(0, 7, (9,), generator=g)
    for k, v in dict(logits=logits, hard=hard, hard_ign=hard_ign, soft=soft, labels=labels, labels_ign=labels_ign, reg_pred=reg_pred,
                     reg_true=reg_true, wvec=wvec, pwvec=pwvec, odd_logits=odd_logits, odd_hard=odd_hard, odd_labels=odd_labels,
                     big_logits=big_logits, big_labels=big_labels, flat_logits=flat_logits, flat_labels=flat_labels).items():
        A[k] = t2n(v)

    def add(name, fn, kwargs, inputs, make):
        """value and d value.sum() / d first input, both from the reference's autograd"""
        x = torch.from_numpy(A[inputs[0]]).clone().requires_grad_(True)
        val = make(x, *[torch.from_numpy(A[i]) for i in inputs[1:]])
        A[name] = t2n(val)
        val.sum().backward()
        A[name + "_grad"] = t2n(x.grad)
        cases.append(dict(name=name, fn=fn, kwargs=kwargs, inputs=inputs, output=name))

    def tensors(kw):
        k2 = dict(kw)
        for key, vec in (("weight", wvec), ("pos_weight", pwvec)):
            if k2.get(key) == "chan":
                k2[key] = vec.view(C, 1, 1)
            elif k2.get(key) == "scalar":
                k2[key] = torch.tensor(1.7)
        return k2

    bce_opts = [
        (dict(ignore_index=None), "logits", "hard"), (dict(), "logits", "hard_ign"), (dict(smooth_factor=0.1), "logits", "hard_ign"),
        (dict(reduction="sum", smooth_factor=0.2, ignore_index=None), "logits", "soft"), (dict(reduction="none"), "logits", "hard_ign"),
        (dict(weight="chan", ignore_index=None), "logits", "hard"), (dict(pos_weight="chan", smooth_factor=0.05), "logits", "hard_ign"),
        (dict(weight="chan", pos_weight="chan", reduction="sum"), "logits", "hard_ign"), (dict(weight="scalar", ignore_index=None), "logits", "soft"),
        (dict(ignore_index=None), "odd_logits", "odd_hard"),
    ]
    for i, (kw, a, b) in enumerate(bce_opts):
        add(f"soft_bce_{i}", "soft_bce", kw, [a, b], lambda x, t, kw=kw: rl.SoftBCEWithLogitsLoss(**tensors(kw))(x, t))
    bal_opts = [(dict(), "logits", "hard"), (dict(gamma=2.0), "logits", "hard"), (dict(ignore_index=-100), "logits", "hard_ign"),
                (dict(reduction="sum", gamma=0.5), "logits", "hard"), (dict(reduction="none", ignore_index=-100), "logits", "hard_ign"),
                (dict(), "odd_logits", "odd_hard")]
    for i, (kw, a, b) in enumerate(bal_opts):
        add(f"balanced_bce_{i}", "balanced_bce", kw, [a, b], lambda x, t, kw=kw: rl.balanced_binary_cross_entropy_with_logits(x, t, **kw))
    qfl_opts = [(dict(), "logits", "soft"), (dict(beta=1.0), "logits", "soft"), (dict(beta=3.0, reduction="sum"), "logits", "hard"),
                (dict(reduction="normalized"), "logits", "soft"), (dict(reduction="none", beta=1.5), "logits", "soft"), (dict(), "odd_logits", "odd_hard")]
    for i, (kw, a, b) in enumerate(qfl_opts):
        add(f"qfl_{i}", "qfl", kw, [a, b], lambda x, t, kw=kw: rl.QualityFocalLoss(**kw)(x, t))
    for i, kw in enumerate([dict(), dict(width=2.0, curvature=1.0), dict(reduction="sum"), dict(reduction="none", width=3.0)]):
        add(f"wing_{i}", "wing", kw, ["reg_pred", "reg_true"], lambda x, t, kw=kw: rlf.wing_loss(x, t, **kw))
    add("logcosh_0", "logcosh", dict(), ["reg_pred", "reg_true"], lambda x, t: rlf.log_cosh_loss(x, t))
    add("logcosh_1", "logcosh", dict(), ["logits", "soft"], lambda x, t: rl.LogCoshLoss()(x, t))
    sce_opts = [(dict(), "logits", "labels"), (dict(smooth_factor=0.1), "logits", "labels_ign"), (dict(smooth_factor=0.2, reduction="sum"), "logits", "labels_ign"),
                (dict(reduction="none", smooth_factor=0.1), "logits", "labels_ign"), (dict(reduction="none", ignore_index=None, smooth_factor=0.3), "logits", "labels"),
                (dict(ignore_index=None), "odd_logits", "odd_labels"), (dict(smooth_factor=0.1), "big_logits", "big_labels"),
                (dict(smooth_factor=0.1), "flat_logits", "flat_labels"), (dict(smooth_factor=1.0), "logits", "labels")]
    for i, (kw, a, b) in enumerate(sce_opts):
        add(f"soft_ce_{i}", "soft_ce", kw, [a, b], lambda x, t, kw=kw: rl.SoftCrossEntropyLoss(**kw)(x, t))
    save("losses2.npz", A, cases)


# ----------------------------------------------------------------------------------------------- classification-shaped losses
def gen_losses3():
    """Bi-tempered, soft-F1 and focal-cosine losses (device-agnostic torch algebra in both implementations)."""
    A, cases = {}, []
    g = torch.Generator().manual_seed(21)
    x = torch.randn((12, 6), generator=g) * 2.0
    y = torch.randint(0, 6, (12,), generator=g)
    xb = torch.randn((3, 1, 6, 5), generator=g) * 2.0
    tb = (torch.rand((3, 1, 6, 5), generator=g) < 0.4).float()
    tbi = tb.clone()
    tbi[torch.rand((3, 1, 6, 5), generator=g) < 0.15] = 255
    for k, v in dict(x=x, y=y, xb=xb, tb=tb, tbi=tbi).items():
        A[k] = t2n(v)

    def add(name, fn, kwargs, inputs, make):
        xin = torch.from_numpy(A[inputs[0]]).clone().requires_grad_(True)
        val = make(xin, torch.from_numpy(A[inputs[1]]))
        A[name] = t2n(val)
        val.sum().backward()
        A[name + "_grad"] = t2n(xin.grad)
        cases.append(dict(name=name, fn=fn, kwargs=kwargs, inputs=inputs, output=name))

    for i, kw in enumerate([dict(t1=0.8, t2=1.2), dict(t1=0.7, t2=0.6), dict(t1=1.0, t2=1.0), dict(t1=0.9, t2=2.0, smoothing=0.1),
                            dict(t1=0.5, t2=1.5, reduction="sum"), dict(t1=0.8, t2=1.4, reduction="none")]):
        add(f"bitempered_{i}", "bitempered", kw, ["x", "y"], lambda a, b, kw=kw: rl.BiTemperedLogisticLoss(**kw)(a, b))
    for i, (kw, t) in enumerate([(dict(t1=0.8, t2=1.3), "tb"), (dict(t1=0.8, t2=1.3, smoothing=0.05, ignore_index=255), "tbi"),
                                 (dict(t1=0.6, t2=0.8, reduction="none"), "tb"), (dict(t1=1.0, t2=1.0), "tb"), (dict(t1=0.9, t2=1.0, smoothing=0.1), "tb"),
                                 (dict(t1=1.0, t2=1.5, reduction="sum"), "tb"), (dict(t1=0.5, t2=4.0, reduction="sum", ignore_index=255), "tbi"),
                                 (dict(t1=0.2, t2=0.5, smoothing=0.02), "tb"), (dict(t1=0.7, t2=2.0, reduction="none", ignore_index=255), "tbi")]):
        add(f"binary_bitempered_{i}", "binary_bitempered", kw, ["xb", t], lambda a, b, kw=kw: rl.BinaryBiTemperedLogisticLoss(**kw)(a, b))
    add("binary_soft_f1_0", "binary_soft_f1", dict(), ["xb", "tb"], lambda a, b: rl.BinarySoftF1Loss()(a, b))
    add("binary_soft_f1_1", "binary_soft_f1", dict(ignore_index=255), ["xb", "tbi"], lambda a, b: rl.BinarySoftF1Loss(ignore_index=255)(a, b))
    add("soft_f1_0", "soft_f1", dict(), ["x", "y"], lambda a, b: rl.SoftF1Loss()(a, b))
    add("focal_cosine_0", "focal_cosine", dict(), ["x", "y"], lambda a, b: rl.FocalCosineLoss()(a, b))
    add("focal_cosine_1", "focal_cosine", dict(alpha=0.5, gamma=1.5, xent=0.3), ["x", "y"], lambda a, b: rl.FocalCosineLoss(alpha=0.5, gamma=1.5, xent=0.3)(a, b))
    save("losses3.npz", A, cases)


# ----------------------------------------------------------------------------------------------- multiscale: nearest + gradients
def gen_tta2():
    """ms_image_augment / ms_image_deaugment of the unmodified reference with mode="nearest", and autograd gradients through the
    bilinear and nearest paths (inference/tta.py:599-621, 645-689; "TTA respects gradient flow", tta.py:3-4)."""
    A, cases = {}, []
    g = torch.Generator().manual_seed(41)
    x = torch.rand((2, 3, 24, 36), generator=g) * 0.9 + 0.05
    A["x"] = t2n(x)
    offsets = [-8, 0, 12, (4, -4)]
    for mode, ac in (("bilinear", False), ("bilinear", True), ("nearest", None)):
        xin = x.clone().requires_grad_(True)
        outs = rtta.ms_image_augment(xin, offsets, mode=mode, align_corners=ac)
        tot = sum((o * (torch.arange(o.numel(), dtype=torch.float32).reshape(o.shape) % 5 + 1.0)).sum() for o in outs)
        tot.backward()
        key = f"aug_{mode}_{ac}"
        for i, o in enumerate(outs):
            A[f"{key}_{i}"] = t2n(o)
        A[f"{key}_grad"] = t2n(xin.grad)
        cases.append(dict(name=key, fn="ms_image_augment_grad", kwargs=dict(size_offsets=[list(o) if isinstance(o, tuple) else o for o in offsets], mode=mode, align_corners=ac)))
    # de-augment: maps of the augmented sizes back to 24 x 36
    fmaps = [torch.rand((2, 3, 24 + (o[0] if isinstance(o, tuple) else o), 36 + (o[1] if isinstance(o, tuple) else o)), generator=g) * 0.9 + 0.05 for o in offsets]
    for i, f in enumerate(fmaps):
        A[f"fm_{i}"] = t2n(f)
    for mode, ac in (("bilinear", False), ("bilinear", True), ("nearest", None)):
        for red in ("mean", "sum", "gmean", "hmean", "harmonic1p", "logodd", "log1p"):
            ins = [f.clone().requires_grad_(True) for f in fmaps]
            out = rtta.ms_image_deaugment(ins, offsets, reduction=red, mode=mode, align_corners=ac)
            (out * (torch.arange(out.numel(), dtype=torch.float32).reshape(out.shape) % 7 + 1.0)).sum().backward()
            key = f"deaug_{mode}_{ac}_{red}"
            A[key] = t2n(out)
            for i, t in enumerate(ins):
                A[f"{key}_grad_{i}"] = t2n(t.grad)
            cases.append(dict(name=key, fn="ms_image_deaugment_grad", kwargs=dict(size_offsets=[list(o) if isinstance(o, tuple) else o for o in offsets], mode=mode, align_corners=ac, reduction=red)))
    # flips inside every scale (what ms_flips_image_deaugment fuses): fliplr / d2 / flips groups, inner and outer reductions
    for group, V in (("fliplr", 2), ("flipud", 2), ("flips", 3), ("d2", 4)):
        ys = [torch.rand((V * 2, 3, f.shape[2], f.shape[3]), generator=g) * 0.9 + 0.05 for f in fmaps]
        for i, y in enumerate(ys):
            A[f"fz_{group}_y{i}"] = t2n(y)
        deaug = getattr(rtta, f"{group}_image_deaugment")
        for inner, outer, ac in (("mean", "mean", True), ("gmean", "gmean", False), ("mean", "gmean", True), ("hmean", "sum", False)):
            out = rtta.ms_image_deaugment([deaug(y, reduction=inner) for y in ys], offsets, reduction=outer, mode="bilinear", align_corners=ac)
            key = f"fz_{group}_{inner}_{outer}_{int(ac)}"
            A[key] = t2n(out)
            cases.append(dict(name=key, fn="ms_flips_image_deaugment", kwargs=dict(group=group, inner_reduction=inner, reduction=outer, align_corners=ac,
                                                                                     size_offsets=[list(o) if isinstance(o, tuple) else o for o in offsets])))
    save("tta2.npz", A, cases)


def gen_tta3():
    """`_deaugment_averaging` over stacks longer than the 8 planes of the TTA groups (tencrop: 10; larger ensembles) and the
    reductions called with their eps argument (inference/tta.py:63-95, inference/functional.py:264-318) -- values and autograd
    gradients of the unmodified reference."""
    A, cases = {}, []
    g = torch.Generator().manual_seed(43)
    for T in (10, 12, 17):
        x = torch.rand((T, 2, 3, 9, 11), generator=g) * 0.98 + 0.01
        x[0, 0, 0, 0, :3] = torch.tensor([0.0, 1.0, 1e-9])            # the clamps of hmean / logodd
        A[f"stack_{T}"] = t2n(x)
        for red in ("mean", "sum", "gmean", "hmean", "harmonic1p", "logodd", "log1p"):
            xin = x.clone().requires_grad_(True)
            out = rtta._deaugment_averaging(xin, red)
            (out * (torch.arange(out.numel(), dtype=torch.float32).reshape(out.shape) % 5 + 1.0)).sum().backward()
            key = f"avg_{T}_{red}"
            A[key], A[key + "_grad"] = t2n(out), t2n(xin.grad)
            cases.append(dict(name=key, fn="deaugment_averaging", kwargs=dict(reduction=red), inputs=[f"stack_{T}"], output=key))
    x = torch.rand((5, 4, 33), generator=g)
    x[0, 0, :4] = torch.tensor([0.0, 1.0, 5e-4, 0.9995])
    A["eps_x"] = t2n(x)
    for fn in ("harmonic_mean", "logodd_mean"):
        for eps in (1e-3, 0.05):
            for dim in (0, 1):
                xin = x.clone().requires_grad_(True)
                out = getattr(rfn, fn)(xin, dim=dim, eps=eps)
                (out * (torch.arange(out.numel(), dtype=torch.float32).reshape(out.shape) % 3 + 1.0)).sum().backward()
                key = f"{fn}_{eps}_{dim}"
                A[key], A[key + "_grad"] = t2n(out), t2n(xin.grad)
                cases.append(dict(name=key, fn=fn, kwargs=dict(dim=dim, eps=eps), inputs=["eps_x"], output=key))
    save("tta3.npz", A, cases)


def gen_tta4():
    """ms_image_augment / ms_image_deaugment with mode="bicubic" (F.interpolate's 4 x 4-tap cubic convolution, A = -0.75): values
    and autograd gradients of the unmodified reference (inference/tta.py:599-621, 645-689)."""
    A, cases = {}, []
    g = torch.Generator().manual_seed(47)
    x = torch.rand((2, 3, 24, 36), generator=g) * 0.9 + 0.05
    A["x"] = t2n(x)
    offsets = [-8, 0, 12, (4, -4)]
    offs_json = [list(o) if isinstance(o, tuple) else o for o in offsets]
    for ac in (False, True):
        xin = x.clone().requires_grad_(True)
        outs = rtta.ms_image_augment(xin, offsets, mode="bicubic", align_corners=ac)
        tot = sum((o * (torch.arange(o.numel(), dtype=torch.float32).reshape(o.shape) % 5 + 1.0)).sum() for o in outs)
        tot.backward()
        key = f"aug_bicubic_{ac}"
        for i, o in enumerate(outs):
            A[f"{key}_{i}"] = t2n(o)
        A[f"{key}_grad"] = t2n(xin.grad)
        cases.append(dict(name=key, fn="ms_image_augment_grad", kwargs=dict(size_offsets=offs_json, mode="bicubic", align_corners=ac)))
    fmaps = [torch.rand((2, 3, 24 + (o[0] if isinstance(o, tuple) else o), 36 + (o[1] if isinstance(o, tuple) else o)), generator=g) * 0.4 + 0.3 for o in offsets]   # (cubic overshoot must stay inside (0, 1) for gmean / logodd)
    for i, f in enumerate(fmaps):
        A[f"fm_{i}"] = t2n(f)
    for ac in (False, True):
        for red in ("mean", "sum", "gmean", "logodd"):
            ins = [f.clone().requires_grad_(True) for f in fmaps]
            out = rtta.ms_image_deaugment(ins, offsets, reduction=red, mode="bicubic", align_corners=ac)
            (out * (torch.arange(out.numel(), dtype=torch.float32).reshape(out.shape) % 7 + 1.0)).sum().backward()
            key = f"deaug_bicubic_{ac}_{red}"
            A[key] = t2n(out)
            for i, t in enumerate(ins):
                A[f"{key}_grad_{i}"] = t2n(t.grad)
            cases.append(dict(name=key, fn="ms_image_deaugment_grad", kwargs=dict(size_offsets=offs_json, mode="bicubic", align_corners=ac, reduction=red)))
    save("tta4.npz", A, cases)


def gen_tta6():
    """ms_image_augment / ms_image_deaugment with the two remaining 4-D modes of F.interpolate, "nearest-exact" and "area" (the
    reference forwards any mode, inference/tta.py:599-621, 645-689; align_corners=None is the only value these modes take): values and
    autograd gradients of the unmodified reference, up- and down-scaling, unequal offsets per axis."""
    A, cases = {}, []
    g = torch.Generator().manual_seed(53)
    x = torch.rand((2, 3, 24, 36), generator=g) * 0.9 + 0.05
    A["x"] = t2n(x)
    offsets = [-9, 0, 13, (5, -7)]
    offs_json = [list(o) if isinstance(o, tuple) else o for o in offsets]
    for mode in ("nearest-exact", "area"):
        xin = x.clone().requires_grad_(True)
        outs = rtta.ms_image_augment(xin, offsets, mode=mode, align_corners=None)
        tot = sum((o * (torch.arange(o.numel(), dtype=torch.float32).reshape(o.shape) % 5 + 1.0)).sum() for o in outs)
        tot.backward()
        key = f"aug_{mode}"
        for i, o in enumerate(outs):
            A[f"{key}_{i}"] = t2n(o)
        A[f"{key}_grad"] = t2n(xin.grad)
        cases.append(dict(name=key, fn="ms_image_augment_grad", kwargs=dict(size_offsets=offs_json, mode=mode, align_corners=None)))
    fmaps = [torch.rand((2, 3, 24 + (o[0] if isinstance(o, tuple) else o), 36 + (o[1] if isinstance(o, tuple) else o)), generator=g) * 0.9 + 0.05 for o in offsets]
    for i, f in enumerate(fmaps):
        A[f"fm_{i}"] = t2n(f)
    for mode in ("nearest-exact", "area"):
        for red in ("mean", "gmean", "logodd"):
            ins = [f.clone().requires_grad_(True) for f in fmaps]
            out = rtta.ms_image_deaugment(ins, offsets, reduction=red, mode=mode, align_corners=None)
            (out * (torch.arange(out.numel(), dtype=torch.float32).reshape(out.shape) % 7 + 1.0)).sum().backward()
            key = f"deaug_{mode}_{red}"
            A[key] = t2n(out)
            for i, t in enumerate(ins):
                A[f"{key}_grad_{i}"] = t2n(t.grad)
            cases.append(dict(name=key, fn="ms_image_deaugment_grad", kwargs=dict(size_offsets=offs_json, mode=mode, align_corners=None, reduction=red)))
    save("tta6.npz", A, cases)


def gen_tta5():
    """The view ops as index permutations of ANY dtype and rank >= 4 (inference/functional.py:47-132: x.flip(3), x.rot90(k, dims=(2, 3)),
    x.transpose(2, 3)), and the augment / de-augment(reduction=None) groups built on them (inference/tta.py:257-524): outputs of the
    unmodified reference for integer, boolean, half, float64 and complex inputs, 4-D / 5-D, square and non-square planes.  Stored as raw
    bytes (uint8 views) so that bfloat16 / complex survive the npz; a case names its dtype and shape."""
    A, cases = {}, []
    ops = ["torch_fliplr", "torch_flipud", "torch_rot90_ccw", "torch_rot90_cw", "torch_rot180", "torch_transpose", "torch_transpose2",
           "torch_rot90_ccw_transpose", "torch_rot90_cw_transpose", "torch_rot180_transpose", "torch_transpose_rot90_ccw",
           "torch_transpose_rot90_cw", "torch_transpose_rot180"]
    dtypes = ["uint8", "int8", "int16", "int32", "int64", "bool", "float16", "bfloat16", "float32", "float64", "complex64"]

    def raw(t):
        return t.detach().contiguous().cpu().view(torch.uint8).numpy() if t.dtype != torch.bool else t.detach().contiguous().cpu().numpy().view(np.uint8)

    def make(shape, name, seed):
        dt = getattr(torch, name)
        g = torch.Generator().manual_seed(seed)
        if dt == torch.bool:
            return torch.rand(shape, generator=g) < 0.5
        if dt.is_complex:
            return torch.complex(torch.randn(shape, generator=g), torch.randn(shape, generator=g)).to(dt)
        if dt.is_floating_point:
            return torch.randn(shape, generator=g, dtype=torch.float64).to(dt)
        info = torch.iinfo(dt)
        return torch.randint(max(info.min, -2**62), min(info.max, 2**62), shape, generator=g, dtype=torch.int64).to(dt)

    for di, name in enumerate(dtypes):
        for si, shape in enumerate(((2, 2, 12, 12), (1, 2, 9, 14), (1, 1, 6, 10, 3))):
            x = make(shape, name, 500 + 10 * di + si)
            xin = f"x_{name}_{si}"
            A[xin] = raw(x)
            for op in ops:
                y = getattr(rfn, op)(x)
                key = f"{op}_{name}_{si}"
                A[key] = raw(y)
                cases.append(dict(name=key, fn=op, kwargs=dict(dtype=name, shape=list(shape), out_shape=list(y.shape)), inputs=[xin], output=key))
            if shape[2] == shape[3]:
                for group in ("fliplr", "flipud", "flips", "d2", "d4"):
                    aug = getattr(rtta, f"{group}_image_augment")(x)
                    key = f"{group}_image_augment_{name}_{si}"
                    A[key] = raw(aug)
                    cases.append(dict(name=key, fn=f"{group}_image_augment", kwargs=dict(dtype=name, shape=list(shape), out_shape=list(aug.shape)),
                                      inputs=[xin], output=key))
                    back = getattr(rtta, f"{group}_image_deaugment")(aug, reduction=None)
                    key = f"{group}_image_deaugment_none_{name}_{si}"
                    A[key] = raw(back)
                    cases.append(dict(name=key, fn=f"{group}_image_deaugment_none", kwargs=dict(dtype=name, shape=list(aug.shape), out_shape=list(back.shape)),
                                      inputs=[f"{group}_image_augment_{name}_{si}"], output=key))
    save("tta5.npz", A, cases)


# ----------------------------------------------------------------------------------------------- TileMerger(dtype=...)
def gen_tiles2():
    """TileMerger with accumulators of the caller's dtype (inference/tiles.py:295-308: image / norm_mask / weight in `dtype`;
    :330-339: integrate_batch casts the batch with type_as and every `+=` rounds to that dtype; :310-319: accumulate_single moves the
    tile but does not cast it): image, norm_mask and merge() of the unmodified reference for float16 / bfloat16 / float64 accumulators fed
    float32 and same-dtype tile batches.  Stored as raw bytes (bfloat16 does not survive an npz otherwise)."""
    A, cases = {}, []

    def raw(t):
        return t.detach().contiguous().cpu().view(torch.uint8).numpy()

    g = torch.Generator().manual_seed(77)
    geoms = [dict(image_shape=[96, 80, 3], tile_size=32, tile_step=16), dict(image_shape=[70, 61, 3], tile_size=[24, 20], tile_step=[9, 11])]
    for gi, kw in enumerate(geoms):
        s = rt.ImageSlicer(kw["image_shape"], kw["tile_size"], kw["tile_step"], weight="pyramid")
        C, n = 2 + gi, len(s.crops)
        th, tw = s.tile_size
        pred = torch.randn((n, C, th, tw), generator=g)
        A[f"m{gi}_pred"] = t2n(pred)
        for name in ("float16", "bfloat16", "float64"):
            dt = getattr(torch, name)
            for feed in ("float32", "same"):
                m = rt.TileMerger(s.target_shape, C, s.weight, dtype=dt)
                x = pred if feed == "float32" else pred.to(dt)
                for b0 in range(0, n - 1, 5):
                    m.integrate_batch(x[b0:min(n - 1, b0 + 5)], s.crops[b0:min(n - 1, b0 + 5)])
                m.accumulate_single(pred[n - 1].to(dt), s.crops[n - 1])          # (no cast inside: the caller supplies the dtype)
                key = f"m{gi}_{name}_{feed}"
                A[key + "_image"], A[key + "_norm"], A[key + "_merged"] = raw(m.image), raw(m.norm_mask), raw(m.merge())
                cases.append(dict(name=key, fn="tile_merger_dtype", kwargs=dict(kw, channels=C, dtype=name, feed=feed, batch=5)))
    save("tiles2.npz", A, cases)


def gen_tiles3():
    """The literal loop on HALF-PRECISION model outputs (torch.autocast): `merger.integrate_batch(tta.<group>_image_deaugment(y), crops)` of
    the unmodified reference with float16 / bfloat16 `y` (inference/tta.py:287-316, 344-365, 442-467 feeding inference/tiles.py:321-339):
    the de-augmentation returns a HALF tensor (the reduced value rounded to the source dtype), integrate_batch widens it to the float32
    accumulator.  Stored: the model outputs and the de-augmented tiles as raw bytes, image / merge() of a float32 TileMerger."""
    A, cases = {}, []

    def raw(t):
        return t.detach().contiguous().cpu().view(torch.uint8).numpy()

    g = torch.Generator().manual_seed(2026)
    s = rt.ImageSlicer([64, 48, 3], 32, 16, weight="pyramid")
    C, n, B = 2, len(s.crops), 4
    th, tw = s.tile_size
    fns = {"d4": (rtta.d4_image_deaugment, 8), "d2": (rtta.d2_image_deaugment, 4), "flips": (rtta.flips_image_deaugment, 3),
           "fliplr": (rtta.fliplr_image_deaugment, 2), "flipud": (rtta.flipud_image_deaugment, 2)}
    for name in ("float16", "bfloat16"):
        dt = getattr(torch, name)
        for group, reduction in (("d4", "mean"), ("d4", "sum"), ("d4", "gmean"), ("d2", "mean"), ("flips", "mean"), ("fliplr", "hmean"), ("flipud", "mean")):
            fn, V = fns[group]
            positive = reduction in ("gmean", "hmean")
            m = rt.TileMerger(s.target_shape, C, s.weight)
            key = f"{name}_{group}_{reduction}"
            for bi, b0 in enumerate(range(0, n, B)):
                nb = min(B, n - b0)
                y = torch.rand((V * nb, C, th, tw), generator=g).clamp(1e-3, 1.0) if positive else torch.randn((V * nb, C, th, tw), generator=g)
                y = y.to(dt)
                tiles = fn(y, reduction=reduction)
                assert tiles.dtype == dt
                m.integrate_batch(tiles, s.crops[b0:b0 + nb])
                A[f"{key}_y{bi}"], A[f"{key}_t{bi}"] = raw(y), raw(tiles)
            A[key + "_image"], A[key + "_merged"] = t2n(m.image), t2n(m.merge())
            cases.append(dict(name=key, fn="literal_loop_half", kwargs=dict(image_shape=[64, 48, 3], tile_size=32, tile_step=16, channels=C, dtype=name,
                                                                            group=group, reduction=reduction, batch=B, views=V)))
    save("tiles3.npz", A, cases)


# ----------------------------------------------------------------------------------------------- focal, activation="softmax"
def gen_losses4():
    """focal_loss_with_logits / BinaryFocalLoss with activation="softmax" (functional.py:61-66): values and autograd gradients of
    the unmodified reference over the option matrix, several softmax dimensions included."""
    A, cases = {}, []
    g = torch.Generator().manual_seed(31)
    B, C, H, W = 3, 5, 12, 10
    x = torch.randn((B, C, H, W), generator=g) * 2.5
    x[1] += 1.0
    lab = torch.randint(0, C, (B, H, W), generator=g)
    lab_ign = lab.clone()
    lab_ign[torch.rand((B, H, W), generator=g) < 0.15] = 255
    onehot = torch.nn.functional.one_hot(lab, C).permute(0, 3, 1, 2).float()
    soft = (torch.rand((B, C, H, W), generator=g) * 0.8 + 0.1)
    onehot_ign = onehot.clone()
    onehot_ign[torch.rand((B, C, H, W), generator=g) < 0.1] = 255
    cw = torch.tensor([0.5, 1.0, 2.0, 1.5, 0.25])
    x3 = torch.randn((7, 9, 11), generator=g) * 2.0          # odd sizes: scalar kernels; 3-D input
    t3 = (torch.rand((7, 9, 11), generator=g) < 0.3).float()
    cw3 = torch.rand(9, generator=g) + 0.5
    for k, v in dict(x=x, lab=lab, lab_ign=lab_ign, onehot=onehot, soft=soft, onehot_ign=onehot_ign, cw=cw, x3=x3, t3=t3, cw3=cw3).items():
        A[k] = t2n(v)

    def add(name, fn, kwargs, inputs, make, wkey=None):
        xin = torch.from_numpy(A[inputs[0]]).clone().requires_grad_(True)
        val = make(xin, torch.from_numpy(A[inputs[1]]))
        A[name] = t2n(val)
        (val * (torch.arange(val.numel(), dtype=torch.float32).reshape(val.shape) % 7 + 1.0) if val.dim() else val).sum().backward()
        A[name + "_grad"] = t2n(xin.grad)
        cases.append(dict(name=name, fn=fn, kwargs=kwargs, inputs=inputs, output=name, weights=wkey))

    opts = [
        dict(softmax_dim=1), dict(softmax_dim=1, alpha=None), dict(softmax_dim=1, gamma=1.5, alpha=0.6), dict(softmax_dim=1, reduction="sum"),
        dict(softmax_dim=1, reduction="none"), dict(softmax_dim=1, reduction="batchwise_mean"), dict(softmax_dim=1, normalized=True),
        dict(softmax_dim=1, reduced_threshold=0.5), dict(softmax_dim=1, reduced_threshold=0.3, normalized=True, gamma=3.0),
        dict(softmax_dim=1, gamma=0.0, alpha=None), dict(softmax_dim=1, gamma=1.0),
        dict(softmax_dim=-1), dict(softmax_dim=0, alpha=None), dict(softmax_dim=2, reduction="none"), dict(softmax_dim=-3, gamma=2.5),
    ]
    for i, kw in enumerate(opts):
        add(f"fsm_{i}", "focal_softmax_fn", kw, ["x", "onehot"], lambda a, b, kw=kw: rlf.focal_loss_with_logits(a, b, activation="softmax", **kw))
    add("fsm_soft", "focal_softmax_fn", dict(softmax_dim=1, alpha=0.3), ["x", "soft"], lambda a, b: rlf.focal_loss_with_logits(a, b, activation="softmax", softmax_dim=1, alpha=0.3))
    add("fsm_ign", "focal_softmax_fn", dict(softmax_dim=1, ignore_index=255, normalized=True), ["x", "onehot_ign"],
        lambda a, b: rlf.focal_loss_with_logits(a, b, activation="softmax", softmax_dim=1, ignore_index=255, normalized=True))
    add("fsm_ign_none", "focal_softmax_fn", dict(softmax_dim=1, ignore_index=255, reduction="none"), ["x", "onehot_ign"],
        lambda a, b: rlf.focal_loss_with_logits(a, b, activation="softmax", softmax_dim=1, ignore_index=255, reduction="none"))
    for i, d in enumerate([1, 0, 2, 3]):   # class weights always follow dim 1 of the tensor, whatever the softmax dimension
        add(f"fsm_cw_{i}", "focal_softmax_fn", dict(softmax_dim=d, class_weights=True), ["x", "onehot"],
            lambda a, b, d=d: rlf.focal_loss_with_logits(a, b, activation="softmax", softmax_dim=d, class_weights=cw), wkey="cw")
    for i, d in enumerate([0, 1, 2]):
        add(f"fsm_3d_{i}", "focal_softmax_fn", dict(softmax_dim=d, class_weights=True, reduction="sum"), ["x3", "t3"],
            lambda a, b, d=d: rlf.focal_loss_with_logits(a, b, activation="softmax", softmax_dim=d, class_weights=cw3, reduction="sum"), wkey="cw3")
    # the module with a label map (one-hot along dim 1, ignore handling) and a dense map
    for i, (kw, t) in enumerate([(dict(), "lab"), (dict(alpha=0.25), "lab"), (dict(ignore_index=255), "lab_ign"),
                                 (dict(ignore_index=255, normalized=True, alpha=0.4), "lab_ign"), (dict(reduction="sum", gamma=1.0), "lab"),
                                 (dict(), "onehot")]):
        add(f"fsm_mod_{i}", "focal_softmax_module", dict(kw, softmax_dim=1), ["x", t],
            lambda a, b, kw=kw: rl.BinaryFocalLoss(activation="softmax", softmax_dim=1, **kw)(a, b))
    add("fsm_mod_cw", "focal_softmax_module", dict(softmax_dim=1, class_weights=True), ["x", "lab"],
        lambda a, b: rl.BinaryFocalLoss(activation="softmax", softmax_dim=1, class_weights=cw)(a, b), wkey="cw")
    save("losses4.npz", A, cases)


# ----------------------------------------------------------------------------------------------- 3-D tiles (8f-4)
def gen_losses5():
    """Lovasz at sizes that span several 4096-element tiles and pixel blocks of the HIP sort (70 x 61 pixels: 8 540 per batch segment),
    every `classes` form of _lovasz_softmax (losses/lovasz.py:92-140) and the hinge form, values AND the reference's autograd gradients."""
    from pytorch_toolbelt.losses import lovasz as rlv

    g = torch.Generator().manual_seed(20260924)
    A, cases = {}, []
    B, C, H, W = 2, 4, 70, 61
    probs = torch.softmax(torch.randn((B, C, H, W), generator=g) * 2, 1)
    labels = torch.randint(0, C, (B, H, W), generator=g)
    labels[labels == C - 1] = 0                       # class 3 absent everywhere: "present" and "all" differ
    labels[1][labels[1] == 0] = 1                     # class 0 absent in image 1
    labels_ign = labels.clone()
    labels_ign[torch.rand((B, H, W), generator=g) < 0.1] = 255
    bl = torch.randn((B, H, W), generator=g) * 2
    bt = (torch.rand((B, H, W), generator=g) < 0.4).float()
    bti = bt.clone()
    bti[torch.rand((B, H, W), generator=g) < 0.1] = 255.0
    A.update(probs=t2n(probs), labels=t2n(labels), labels_ign=t2n(labels_ign), bl=t2n(bl), bt=t2n(bt), bti=t2n(bti))

    def add(name, fn, kw, inputs, value, grad):
        A[name], A[name + "_grad"] = t2n(value), t2n(grad)
        cases.append(dict(name=name, fn=fn, kwargs=kw, inputs=inputs, output=name, grad=name + "_grad"))

    i = 0
    for classes in ("present", "all", [0, 2], [1]):
        for per_image in (False, True):
            for ign in (None, 255):
                x = probs.clone().requires_grad_(True)
                val = rlv._lovasz_softmax(x, labels_ign if ign is not None else labels, classes=classes, per_image=per_image, ignore_index=ign)
                val.backward()
                add(f"lovasz_softmax_{i}", "lovasz_softmax", dict(classes=classes, per_image=per_image, ignore_index=ign),
                    ["probs", "labels_ign" if ign is not None else "labels"], val, x.grad)
                i += 1
    i = 0
    for per_image in (False, True):
        for ign in (None, 255):
            x = bl.clone().requires_grad_(True)
            val = rlv._lovasz_hinge(x, bti if ign is not None else bt, per_image=per_image, ignore_index=ign)
            val.backward()
            add(f"lovasz_hinge_{i}", "lovasz_hinge", dict(per_image=per_image, ignore_index=ign), ["bl", "bti" if ign is not None else "bt"], val, x.grad)
            i += 1
    save("losses5.npz", A, cases)


def gen_losses6():
    """The fractional-gamma x ignore_index corner of the sigmoid focal loss on LABEL targets: BinaryFocalLoss
    writes `ignore_index` into every channel of an ignored pixel (losses/focal.py:99-105), `(1 - pt).pow(gamma)` of the then negative base
    is NaN for a non-integer gamma (losses/functional.py:70), `masked_fill` hides it in the VALUE (:90-94) and `0 * NaN` brings it back
    in the GRADIENT of exactly the ignored pixels.  Values and autograd gradients of the unmodified reference, NaNs included."""
    A, cases = {}, []
    g = torch.Generator().manual_seed(606)
    B, C, H, W = 2, 4, 12, 16
    logits = torch.randn((B, C, H, W), generator=g) * 2
    labels = torch.randint(0, C, (B, H, W), generator=g)
    labels[torch.rand((B, H, W), generator=g) < 0.2] = 255
    A["logits"], A["labels_ign"] = t2n(logits), t2n(labels)
    opts = [dict(gamma=1.5, ignore_index=255), dict(gamma=0.5, ignore_index=255, alpha=0.3), dict(gamma=1.5, ignore_index=255, normalized=True),
            dict(gamma=2.5, ignore_index=255, reduced_threshold=0.5), dict(gamma=1.5, ignore_index=255, reduction="sum"),
            dict(gamma=2.0, ignore_index=255)]
    for i, kw in enumerate(opts):
        x = logits.clone().requires_grad_(True)
        value = rl.BinaryFocalLoss(**kw)(x, labels)
        value.backward()
        name = f"binary_focal_frac_ign_{i}"
        A[name], A[name + "_grad"] = t2n(value), t2n(x.grad)
        cases.append(dict(name=name, fn="binary_focal_loss_grad", kwargs=kw, inputs=["logits", "labels_ign"], output=name,
                          nan_grads=int(torch.isnan(x.grad).sum())))
    save("losses6.npz", A, cases)


def gen_volumes():
    from pytorch_toolbelt.inference import tiles_3d as rt3

    A, cases = {}, []
    rng = np.random.default_rng(13)
    g = torch.Generator().manual_seed(13)
    geoms = [dict(volume_shape=[20, 33, 17], voxel_size=[8, 16, 8], voxel_step=[4, 8, 8]),
             dict(volume_shape=[16, 16, 16], voxel_size=8, voxel_step=8),
             dict(volume_shape=[9, 21, 30], voxel_size=[4, 8, 12], voxel_step=[3, 5, 12]),
             dict(volume_shape=[5, 6, 7], voxel_size=[8, 8, 8], voxel_step=[4, 4, 4]),      # volume smaller than a tile
             dict(volume_shape=[64, 96, 80, 2], voxel_size=[32, 32, 32], voxel_step=[16, 16, 16])]
    for k, kw in enumerate(geoms):
        s = rt3.VolumeSlicer(kw["volume_shape"], kw["voxel_size"], kw["voxel_step"])
        A[f"vgeom{k}_starts"] = np.array([[r.start for r in roi] for roi in s.crops], dtype=np.int64)
        A[f"vgeom{k}_stops"] = np.array([[r.stop for r in roi] for roi in s.crops], dtype=np.int64)
        A[f"vgeom{k}_bbox_starts"] = np.array([[r.start for r in roi] for roi in s.bbox_crops], dtype=np.int64)
        A[f"vgeom{k}_meta"] = np.array([*s.pad_before, *s.pad_after, *s.target_shape, *s.num_tiles], dtype=np.int64)
        cases.append(dict(name=f"vgeom{k}", fn="vgeometry", kwargs=kw))
    for k, kw in enumerate(geoms[:4]):
        vol = rng.integers(0, 256, kw["volume_shape"], dtype=np.uint8)
        s = rt3.VolumeSlicer(vol.shape, kw["voxel_size"], kw["voxel_step"])
        tiles = s.split(vol, value=5)
        A[f"vsplit{k}_volume"] = vol
        A[f"vsplit{k}_tiles"] = np.stack(tiles)
        it = list(s.iter_split(vol, value=5))
        assert all(np.array_equal(a, b) for (a, _), b in zip(it, tiles))
        A[f"vsplit{k}_crop"] = s.crop_to_orignal_size(np.pad(vol, np.stack([s.pad_before, s.pad_after], -1)))
        cases.append(dict(name=f"vsplit{k}", fn="vsplit", kwargs=kw))
    for k, (kw, C, bs) in enumerate([(geoms[0], 2, 5), (geoms[2], 1, 4), (geoms[1], 3, 8), (geoms[3], 2, 1)]):
        s = rt3.VolumeSlicer(kw["volume_shape"], kw["voxel_size"], kw["voxel_step"])
        ts = tuple(int(v) for v in s.tile_size)
        weight = (rng.random(ts) + 0.25).astype(np.float32)
        pred = torch.randn((len(s.crops), C, *ts), generator=g)
        m = rt3.VolumeMerger(s.target_shape, C, weight)
        for b0 in range(0, len(s.crops), bs):
            m.integrate_batch(pred[b0:b0 + bs], s.crops[b0:b0 + bs])
        A[f"vmerger{k}_weight"], A[f"vmerger{k}_pred"] = weight, t2n(pred)
        A[f"vmerger{k}_volume"], A[f"vmerger{k}_norm"], A[f"vmerger{k}_merged"] = t2n(m.volume), t2n(m.norm_mask), t2n(m.merge())
        cases.append(dict(name=f"vmerger{k}", fn="vmerger", kwargs=dict(kw, channels=C, batch=bs)))
    save("volumes.npz", A, cases)

# ----------------------------------------------------------------------------------------------- BASELINE configs at full size
def gen_fullsize():
    """Digests of the UNMODIFIED reference's outputs on every BASELINE.json config at its stated size.

    The inputs are too big to store (cfg2: 12.1 GB), so they come from oracle/synth.py -- an integer hash that numpy (here)
    and torch on the GPU (the `-m gpu` tests) evaluate bit-identically -- and only a strided subsample plus float64 sums
    of every output are kept.  Each entry: `<cfg>_sub` (subsample, steps in `<cfg>_meta`), `<cfg>_sums` (sum, abs-sum)."""
    sys.path.insert(0, ROOT)
    from oracle import synth as SY

    A, cases = {}, []
    torch.set_num_threads(os.cpu_count() or 8)

    # cfg1: 1024x1024x3, ImageSlicer 256/128, pyramid window, C = 3 (reference tests/test_tiles.py path, CPU TileMerger)
    s = rt.ImageSlicer((1024, 1024, 3), tile_size=256, tile_step=128, weight="pyramid")
    assert len(s.crops) == 49
    pred = torch.from_numpy(SY.synth_np((49, 3, 256, 256), 101))
    m = rt.TileMerger(s.target_shape, 3, s.weight)
    for b0 in range(0, 49, 8):
        m.integrate_batch(pred[b0:b0 + 8], s.crops[b0:b0 + 8])
    merged = t2n(m.merge())
    A["cfg1_sub"], A["cfg1_sums"] = SY.digest(merged, 13, 17)
    A["cfg1_meta"] = np.array([13, 17, 101])
    # the host fp64 path of the same tiles (ImageSlicer.merge, HWC tiles)
    tiles_hwc = [np.moveaxis(t2n(p), 0, -1) for p in pred]
    host = s.merge(tiles_hwc, dtype=np.float32)
    A["cfg1_host_sub"], A["cfg1_host_sums"] = SY.digest(np.moveaxis(host, -1, 0), 13, 17)
    cases.append(dict(name="cfg1", fn="fullsize", kwargs=dict(image=[1024, 1024, 3], tile=256, step=128, channels=3, seed=101, batch=8)))

    # cfg2: 5000x5000x3, 512/256 pyramid, d4 TTA model outputs C = 4, batches of 8 tiles, seeds 2000 + batch index
    s = rt.ImageSlicer((5000, 5000, 3), tile_size=(512, 512), tile_step=(256, 256), weight="pyramid")
    assert len(s.crops) == 361
    m = rt.TileMerger(s.target_shape, 4, s.weight)
    for k, b0 in enumerate(range(0, 361, 8)):
        nb = min(8, 361 - b0)
        y = torch.from_numpy(SY.synth_np((8 * nb, 4, 512, 512), 2000 + k))
        m.integrate_batch(rtta.d4_image_deaugment(y, reduction="mean"), s.crops[b0:b0 + nb])
    merged = t2n(m.merge())
    A["cfg2_sub"], A["cfg2_sums"] = SY.digest(merged, 97, 101)
    A["cfg2_meta"] = np.array([97, 101, 2000])
    A["cfg2_rows"] = merged[:, [0, 255, 256, 2559, 2560, 5119], :].copy()     # whole rows across band / tile edges
    cases.append(dict(name="cfg2", fn="fullsize", kwargs=dict(image=[5000, 5000, 3], tile=512, step=256, channels=4, seed=2000, batch=8, group="d4")))
    del merged, m

    # cfg4: [32, 16, 512, 512] logits, int64 labels: BinaryFocalLoss (one-hot target), DiceLoss / JaccardLoss multiclass
    B, C, H, W = 32, 16, 512, 512
    x = torch.from_numpy(SY.synth_np((B, C, H, W), 4001)) * 2.0          # uniform on [-4, 4), exact scaling
    lab = torch.from_numpy(SY.labels_np((B, H, W), 4002, C))
    onehot = torch.nn.functional.one_hot(lab, C).permute(0, 3, 1, 2).float()
    vals = dict(
        focal=float(rl.BinaryFocalLoss()(x, onehot)),
        focal_alpha=float(rl.BinaryFocalLoss(alpha=0.25, gamma=2.0)(x, onehot)),
        dice=float(rl.DiceLoss("multiclass")(x, lab)),
        jaccard=float(rl.JaccardLoss("multiclass")(x, lab)),
        ce_focal=float(rl.CrossEntropyFocalLoss()(x, lab)),
    )
    del onehot
    for k_, v in vals.items():
        A[f"cfg4_{k_}"] = np.array(v, dtype=np.float64)
    # gradient digest of focal + dice + jaccard (the fused loss of BASELINE configs[3])
    xg = x.clone().requires_grad_(True)
    onehot = torch.nn.functional.one_hot(lab, C).permute(0, 3, 1, 2).float()
    total = rl.BinaryFocalLoss()(xg, onehot) + rl.DiceLoss("multiclass")(xg, lab) + rl.JaccardLoss("multiclass")(xg, lab)
    total.backward()
    A["cfg4_fused"] = np.array(float(total), dtype=np.float64)
    A["cfg4_grad_sub"], A["cfg4_grad_sums"] = SY.digest(t2n(xg.grad), 37, 41)
    A["cfg4_meta"] = np.array([37, 41, 4001, 4002])
    cases.append(dict(name="cfg4", fn="fullsize", kwargs=dict(shape=[B, C, H, W], seeds=[4001, 4002], scale=2.0)))
    del x, xg, onehot, lab

    # cfg5: multiscale 0.75 / 1.0 / 1.25 of 4096x4096 (pixel offsets -1024, 0, +1024), each scale wrapped in fliplr TTA,
    # gmean merge; C = 4, probabilities in (0, 1]
    offs = [-1024, 0, 1024]
    per_scale = []
    for i, o in enumerate(offs):
        y = torch.from_numpy(SY.synth_np((2, 4, 4096 + o, 4096 + o), 5000 + i, "unit"))
        per_scale.append(rtta.fliplr_image_deaugment(y, reduction="gmean"))
        del y
    for ac in (False, True):
        out = t2n(rtta.ms_image_deaugment(per_scale, offs, reduction="gmean", mode="bilinear", align_corners=ac))
        A[f"cfg5_ac{int(ac)}_sub"], A[f"cfg5_ac{int(ac)}_sums"] = SY.digest(out, 89, 83)
    A["cfg5_meta"] = np.array([89, 83, 5000])
    cases.append(dict(name="cfg5", fn="fullsize", kwargs=dict(size=4096, offsets=offs, channels=4, seed=5000, inner="fliplr:gmean", reduction="gmean")))
    save("fullsize.npz", A, cases)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "fullsize":
        gen_fullsize()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] in globals():    # one fixture: python oracle/make_golden.py gen_losses4
        globals()[sys.argv[1]]()
        sys.exit(0)
    gen_tiles()
    gen_tiles2()
    gen_tiles3()
    gen_tta()
    gen_losses()
    gen_edges()
    gen_ensembling()
    gen_losses2()
    gen_losses3()
    gen_losses4()
    gen_losses5()
    gen_losses6()
    gen_tta2()
    gen_tta3()
    gen_tta4()
    gen_tta5()
    gen_tta6()
    gen_volumes()
    gen_fullsize()
