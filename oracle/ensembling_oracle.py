"""Oracle (test infrastructure only) for inference/ensembling.py of the reference: numpy restatement.

Citations are ``pytorch_toolbelt/inference/ensembling.py:LINE``."""
import numpy as np

from . import tta_oracle as AO


def sigmoid_to(x, temperature=1.0):
    """ApplySigmoidTo.forward, :62-66: ``output.mul(temperature).sigmoid_()`` in float32."""
    z = (np.asarray(x, dtype=np.float32) * np.float32(temperature)).astype(np.float64)
    return (1.0 / (1.0 + np.exp(-z))).astype(np.float32)


def softmax_to(x, temperature=1.0, dim=1):
    """ApplySoftmaxTo.forward, :38-42: ``output.mul(temperature).softmax(dim)``; float64 internally, float32 out."""
    z = (np.asarray(x, dtype=np.float32) * np.float32(temperature)).astype(np.float64)
    z = z - z.max(axis=dim, keepdims=True)
    e = np.exp(z)
    return (e / e.sum(axis=dim, keepdims=True)).astype(np.float32)


def ensemble(outputs, reduction="mean"):
    """Ensembler.forward for one key, :108-117: ``torch.stack(outputs)`` then ``_deaugment_averaging`` over dim 0."""
    return AO.deaugment_averaging(np.stack([np.asarray(o, dtype=np.float32) for o in outputs]), reduction)
