"""CPU: the ``compat/pytorch_toolbelt`` alias exposes the reference's hot-path names."""
import os
import sys

from conftest import ROOT


def test_alias_surface():
    sys.path.insert(0, os.path.join(ROOT, "compat"))
    try:
        from pytorch_toolbelt import losses as L
        from pytorch_toolbelt.inference import functional as F
        from pytorch_toolbelt.inference import tiles, tta
        from pytorch_toolbelt.utils.torch_utils import image_to_tensor, rgb_image_from_tensor, tensor_from_rgb_image, to_numpy  # noqa: F401
    finally:
        sys.path.pop(0)
    from pytorch_toolbelt.inference import ensembling
    import pytorch_toolbelt.inference as inf

    for n in ("ApplySoftmaxTo", "ApplySigmoidTo", "Ensembler", "PickModelOutput", "SelectByIndex", "average_checkpoints"):
        assert hasattr(ensembling, n) and hasattr(inf, n)
    for n in ("ImageSlicer", "TileMerger", "compute_pyramid_patch_weight_loss"):
        assert hasattr(tiles, n)
    ref_tta = ["GeneralizedTTA", "MultiscaleTTA", "d2_image_augment", "d2_labels_augment", "d2_image_deaugment", "d2_labels_deaugment",
               "d4_image2label", "d4_image2mask", "d4_image_augment", "d4_labels_augment", "d4_image_deaugment", "d4_labels_deaugment",
               "fivecrop_image2label", "fivecrop_image_augment", "fivecrop_label_deaugment", "fliplr_image2label", "fliplr_image2mask",
               "fliplr_image_augment", "fliplr_labels_augment", "fliplr_image_deaugment", "fliplr_labels_deaugment", "flips_image_augment",
               "flips_labels_augment", "flips_image_deaugment", "flips_labels_deaugment", "flipud_image_augment", "flipud_image_deaugment",
               "flipud_labels_deaugment", "ms_image_augment", "ms_labels_augment", "ms_image_deaugment", "tencrop_image2label",
               "ms_labels_deaugment", "TTAWrapper", "split_into_chunks", "_deaugment_averaging"]
    assert all(hasattr(tta, n) for n in ref_tta)
    ref_fn = ["geometric_mean", "harmonic_mean", "harmonic1p_mean", "logodd_mean", "log1p_mean", "pad_image_tensor", "pad_tensor_to_size",
              "torch_fliplr", "torch_flipud", "torch_none", "torch_rot180", "torch_rot270", "torch_rot90", "torch_rot90_ccw",
              "torch_rot90_ccw_transpose", "torch_rot90_cw", "torch_rot90_cw_transpose", "torch_transpose", "torch_transpose2",
              "torch_transpose_", "torch_transpose_rot90_ccw", "torch_transpose_rot90_cw", "unpad_image_tensor", "unpad_xyxy_bboxes",
              "torch_rot180_transpose", "torch_transpose_rot180"]
    assert all(hasattr(F, n) for n in ref_fn)
    ref_losses = ["BinaryFocalLoss", "CrossEntropyFocalLoss", "FocalLoss", "DiceLoss", "JaccardLoss", "BinaryLovaszLoss", "LovaszLoss",
                  "BINARY_MODE", "MULTICLASS_MODE", "MULTILABEL_MODE", "focal_loss_with_logits", "softmax_focal_loss_with_logits",
                  "sigmoid_focal_loss", "soft_dice_score", "soft_jaccard_score", "wing_loss", "log_cosh_loss",
                  "BalancedBCEWithLogitsLoss", "balanced_binary_cross_entropy_with_logits", "BiTemperedLogisticLoss",
                  "BinaryBiTemperedLogisticLoss", "FocalCosineLoss", "SoftBCEWithLogitsLoss", "SoftCrossEntropyLoss", "soft_micro_f1",
                  "BinarySoftF1Loss", "SoftF1Loss", "WingLoss", "LogCoshLoss", "QualityFocalLoss", "label_smoothed_nll_loss"]
    assert all(hasattr(L, n) for n in ref_losses)
