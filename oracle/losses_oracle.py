"""Oracle (test infrastructure only) for the focal / Dice / Jaccard / Lovasz losses of the reference.

numpy restatement evaluated in float64 (inputs are promoted), i.e. the mathematically exact value of the
formula the reference evaluates in float32; the parity tolerance for loss scalars is 1e-5 absolute.
Citations: ``lf.py`` = pytorch_toolbelt/losses/functional.py, ``focal.py``/``dice.py``/``jaccard.py``/``lovasz.py``
= pytorch_toolbelt/losses/<name>.
"""
import numpy as np

F64 = np.float64


# --------------------------------------------------------------------------- elementwise helpers
def _sigmoid(x):
    return 0.5 * (1.0 + np.tanh(0.5 * x))


def _log_sigmoid(x):
    return np.minimum(x, 0.0) - np.log1p(np.exp(-np.abs(x)))


def _softmax(x, axis):
    z = x - np.max(x, axis=axis, keepdims=True)
    e = np.exp(z)
    return e / np.sum(e, axis=axis, keepdims=True)


def _log_softmax(x, axis):
    z = x - np.max(x, axis=axis, keepdims=True)
    return z - np.log(np.sum(np.exp(z), axis=axis, keepdims=True))


def _bce_with_logits(x, t):
    """torch BCEWithLogits(reduction='none'): max(x,0) - x*t + log(1+exp(-|x|))."""
    return np.maximum(x, 0.0) - x * t + np.log1p(np.exp(-np.abs(x)))


def one_hot(labels, num_classes, axis=1):
    """F.one_hot + moveaxis(-1, axis); out-of-range labels raise like torch (quirk Q12)."""
    labels = np.asarray(labels)
    if labels.size and (labels.min() < 0 or labels.max() >= num_classes):
        raise RuntimeError("Class values must be smaller than num_classes.")
    oh = (labels[..., None] == np.arange(num_classes)).astype(np.int64)
    return np.moveaxis(oh, -1, axis)


def _reduce(loss, reduction):
    """lf.py:100-105 / :166-171 -- 'batchwise_mean' is a SUM over dim 0 (quirk Q7); unknown -> unreduced."""
    if reduction == "mean":
        return loss.mean()
    if reduction == "sum":
        return loss.sum()
    if reduction == "batchwise_mean":
        return loss.sum(axis=0)
    return loss


# --------------------------------------------------------------------------- focal
def focal_loss_with_logits(output, target, gamma=2.0, alpha=0.25, reduction="mean", normalized=False,
                           reduced_threshold=None, eps=1e-6, ignore_index=None, activation="sigmoid",
                           softmax_dim=None, class_weights=None):
    """lf.py:19-107."""
    x = np.asarray(output, dtype=F64)
    t = np.asarray(target, dtype=F64)
    p = _sigmoid(x) if activation == "sigmoid" else _softmax(x, softmax_dim)       # :61-64
    ce = _bce_with_logits(x, t)                                                     # :66
    pt = p * t + (1.0 - p) * (1.0 - t)                                              # :67
    if reduced_threshold is None:
        focal = np.power(1.0 - pt, gamma)                                           # :71
    else:
        focal = np.power((1.0 - pt) / (1.0 - reduced_threshold), gamma)             # :73-75
        focal = np.where(pt < reduced_threshold, 1.0, focal)                        # :76
    loss = focal * ce                                                               # :78
    if alpha is not None:
        loss = loss * (alpha * t + (1.0 - alpha) * (1.0 - t))                       # :80-81
    if class_weights is not None:
        cw = np.asarray(class_weights, dtype=F64).reshape((1, -1) + (1,) * (loss.ndim - 2))
        loss = loss * cw                                                            # :83-88
    if ignore_index is not None:
        ign = t == ignore_index                                                     # :90-94
        loss = np.where(ign, 0.0, loss)
        if normalized:
            focal = np.where(ign, 0.0, focal)
    if normalized:
        loss = loss / max(focal.sum(), eps)                                         # :96-98
    return _reduce(loss, reduction)


def binary_focal_targets(targets, num_classes, ignore_index=None):
    """BinaryFocalLoss one-hot expansion, focal.py:88-105: ignored pixels carry ignore_index in EVERY channel."""
    targets = np.asarray(targets)
    if ignore_index is None:
        return one_hot(targets, num_classes)
    ign = targets == ignore_index
    oh = one_hot(np.where(ign, 0, targets), num_classes)
    return np.where(ign[:, None], ignore_index, oh)


def binary_focal_loss(inputs, targets, alpha=None, gamma=2.0, ignore_index=None, reduction="mean",
                      normalized=False, reduced_threshold=None, activation="sigmoid", softmax_dim=None,
                      class_weights=None):
    """BinaryFocalLoss.forward, focal.py:77-92 (module default alpha=None, quirk Q9)."""
    inputs = np.asarray(inputs)
    targets = np.asarray(targets)
    if targets.ndim + 1 == inputs.ndim:
        targets = binary_focal_targets(targets, inputs.shape[1], ignore_index)
    return focal_loss_with_logits(inputs, targets, gamma=gamma, alpha=alpha, reduction=reduction,
                                  normalized=normalized, reduced_threshold=reduced_threshold,
                                  ignore_index=ignore_index, activation=activation, softmax_dim=softmax_dim,
                                  class_weights=class_weights)


def softmax_focal_loss_with_logits(output, target, class_weights=None, gamma=2.0, reduction="mean",
                                   normalized=False, reduced_threshold=None, eps=1e-6, ignore_index=-100):
    """lf.py:110-173 (CrossEntropyFocalLoss)."""
    x = np.asarray(output, dtype=F64)
    target = np.asarray(target)
    ign = target == ignore_index                                                    # :137
    oh = one_hot(np.where(ign, 0, target), x.shape[1]).astype(F64)                  # :139-140
    probs = _softmax(x, 1)                                                          # :141
    pt = (1.0 - oh) * probs + oh * (1.0 - probs)                                    # :142
    loss = _bce_with_logits(x, oh)                                                  # :144
    if reduced_threshold is None:
        focal = np.power(pt, gamma)                                                 # :148
    else:
        focal = np.power(pt / reduced_threshold, gamma)                             # :150
        focal = np.where(pt < reduced_threshold, 1.0, focal)                        # :151
    loss = focal * loss
    if class_weights is not None:
        loss = loss * np.asarray(class_weights, dtype=F64).reshape((1, -1) + (1,) * (loss.ndim - 2))
    loss = loss.sum(axis=1) * (~ign)                                                # :159
    if normalized:
        loss = loss / max(focal.sum(), eps)                                         # :161-164 (focal NOT masked)
    return _reduce(loss, reduction)


# --------------------------------------------------------------------------- dice / jaccard
def soft_jaccard_score(output, target, smooth=0.0, eps=1e-7, dims=None):
    """lf.py:188-218."""
    o = np.asarray(output, dtype=F64)
    t = np.asarray(target, dtype=F64)
    assert o.shape == t.shape
    ax = None if dims is None else tuple(dims)
    inter = np.sum(o * t, axis=ax)
    card = np.sum(o + t, axis=ax)
    union = card - inter
    return (inter + smooth) / np.maximum(union + smooth, eps)


def soft_dice_score(output, target, smooth=0.0, eps=1e-7, dims=None):
    """lf.py:221-247."""
    o = np.asarray(output, dtype=F64)
    t = np.asarray(target, dtype=F64)
    assert o.shape == t.shape
    ax = None if dims is None else tuple(dims)
    inter = np.sum(o * t, axis=ax)
    card = np.sum(o + t, axis=ax)
    return (2.0 * inter + smooth) / np.maximum(card + smooth, eps)


def _region_loss(score_fn, y_pred, y_true, mode, classes, log_loss, from_logits, smooth, ignore_index, eps):
    """Shared body of DiceLoss.forward (dice.py:59-131) and JaccardLoss.forward (jaccard.py:48-103)."""
    y_pred = np.asarray(y_pred, dtype=F64)
    y_true = np.asarray(y_true)
    assert y_true.shape[0] == y_pred.shape[0]
    if from_logits:                                                      # dice.py:68-75
        y_pred = np.exp(_log_softmax(y_pred, 1)) if mode == "multiclass" else np.exp(_log_sigmoid(y_pred))
    bs, C = y_true.shape[0], y_pred.shape[1]
    if mode == "binary":                                                 # dice.py:81-88
        y_true = y_true.reshape(bs, 1, -1)
        y_pred = y_pred.reshape(bs, 1, -1)
        if ignore_index is not None:
            m = y_true != ignore_index
            y_pred = y_pred * m
            y_true = y_true * m
    elif mode == "multiclass":                                           # dice.py:90-102
        y_true = y_true.reshape(bs, -1)
        y_pred = y_pred.reshape(bs, C, -1)
        if ignore_index is not None:
            m = y_true != ignore_index
            y_pred = y_pred * m[:, None]
            y_true = one_hot((y_true * m).astype(np.int64), C) * m[:, None]
        else:
            y_true = one_hot(y_true, C)
    else:                                                                # multilabel, dice.py:104-111
        y_true = y_true.reshape(bs, C, -1)
        y_pred = y_pred.reshape(bs, C, -1)
        if ignore_index is not None:
            m = y_true != ignore_index
            y_pred = y_pred * m
            y_true = y_true * m
    y_true = y_true.astype(F64)
    scores = score_fn(y_pred, y_true, smooth=smooth, eps=eps, dims=(0, 2))
    loss = -np.log(np.maximum(scores, eps)) if log_loss else 1.0 - scores   # dice.py:115-118
    loss = loss * (y_true.sum(axis=(0, 2)) > 0)                             # dice.py:125-126
    if classes is not None:
        loss = loss[np.asarray(classes, dtype=np.int64)]                    # dice.py:128-129 (evident intent, quirk Q17)
    return loss.mean()


def dice_loss(y_pred, y_true, mode, classes=None, log_loss=False, from_logits=True, smooth=0.0,
              ignore_index=None, eps=1e-7):
    return _region_loss(soft_dice_score, y_pred, y_true, mode, classes, log_loss, from_logits, smooth, ignore_index, eps)


def jaccard_loss(y_pred, y_true, mode, classes=None, log_loss=False, from_logits=True, smooth=0.0, eps=1e-7):
    """JaccardLoss has no ignore_index (quirk Q11)."""
    return _region_loss(soft_jaccard_score, y_pred, y_true, mode, classes, log_loss, from_logits, smooth, None, eps)


# --------------------------------------------------------------------------- lovasz
def lovasz_grad(gt_sorted):
    """lovasz.py:23-34: gradient of the Lovasz extension w.r.t. sorted errors."""
    g = np.asarray(gt_sorted, dtype=F64)
    total = g.sum()
    inter = total - np.cumsum(g)
    union = total + np.cumsum(1.0 - g)
    jac = 1.0 - inter / union
    if len(g) > 1:
        jac[1:] = jac[1:] - jac[:-1]
    return jac


def _hinge_flat(logits, labels):
    """lovasz.py:52-72."""
    if len(labels) == 0:
        return F64(0.0)
    lab = np.asarray(labels, dtype=F64)
    err = 1.0 - np.asarray(logits, dtype=F64) * (2.0 * lab - 1.0)
    # (equal errors: the reference's torch.sort leaves their order open and the loss does not depend on it; stable here, by the
    # foreground flag and then by index in the HIP sort -- tests compare gradients of tied elements with a tolerance, not bits)
    order = np.argsort(-err, kind="stable")
    return float(np.dot(np.maximum(err[order], 0.0), lovasz_grad(lab[order])))


def _mean_py(values):
    """lovasz.py:168-184 `mean`: acc/n, 0 for an empty sequence, the single value for n == 1."""
    values = list(values)
    if not values:
        return 0
    if len(values) == 1:
        return values[0]
    return sum(values[1:], values[0]) / len(values)


def lovasz_hinge(logits, labels, per_image=False, ignore_index=None):
    """BinaryLovaszLoss.forward -> _lovasz_hinge, lovasz.py:37-49 (+ _flatten_binary_scores :75-86)."""
    logits = np.asarray(logits)
    labels = np.asarray(labels)

    def flat(lg, lb):
        lg, lb = lg.reshape(-1), lb.reshape(-1)
        if ignore_index is not None:
            keep = lb != ignore_index
            lg, lb = lg[keep], lb[keep]
        return _hinge_flat(lg, lb)

    if per_image:
        return _mean_py(flat(lg, lb) for lg, lb in zip(logits, labels))
    return flat(logits, labels)


def _softmax_flat(probas, labels, classes="present"):
    """lovasz.py:110-140.  probas [P, C] are PROBABILITIES (no softmax applied, quirk Q13)."""
    if probas.size == 0:
        return F64(0.0)
    C = probas.shape[1]
    todo = list(range(C)) if classes in ("all", "present") else classes
    per_class = []
    for c in todo:
        fg = (labels == c).astype(F64)
        if classes == "present" and fg.sum() == 0:
            continue
        pred = probas[:, 0] if C == 1 else probas[:, c]
        err = np.abs(fg - pred)
        order = np.argsort(-err, kind="stable")
        per_class.append(float(np.dot(err[order], lovasz_grad(fg[order]))))
    return _mean_py(per_class)


def lovasz_softmax(probas, labels, classes="present", per_image=False, ignore_index=None):
    """LovaszLoss.forward -> _lovasz_softmax, lovasz.py:92-107 (+ _flatten_probas :143-160)."""
    probas = np.asarray(probas, dtype=F64)
    labels = np.asarray(labels)
    if probas.ndim == 3:
        probas = probas[:, None]

    def flat(pr, lb):
        C = pr.shape[1]
        pr = np.moveaxis(pr, 1, -1).reshape(-1, C)
        lb = lb.reshape(-1)
        if ignore_index is not None:
            keep = lb != ignore_index
            pr, lb = pr[keep], lb[keep]
        return _softmax_flat(pr, lb, classes)

    if per_image:
        return _mean_py(flat(p[None], l[None]) for p, l in zip(probas, labels))
    return flat(probas, labels)
