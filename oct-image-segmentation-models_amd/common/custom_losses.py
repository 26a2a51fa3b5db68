"""Loss registry with the reference's names and factory signatures
(oct_image_segmentation_models/common/custom_losses.py:47-81,230-255).

On the accelerated path the Dice sums are fused into the HIP head kernel, so ``Model.compile`` selects the
loss by the ``oct_loss`` tag of the callable; the callables themselves are numpy restatements usable on host
arrays (e.g. for evaluation code).  ``focal_dice_loss`` (reference :98-178, built on the third-party ``focal-loss``
package: published formula restated, parity unpinned) is implemented the same way; the BCE mixes and the plain
focal loss are out of scope and raise."""
from __future__ import annotations

import numpy as np


def _one_hot(y_true, num_classes):
    lab = np.asarray(y_true)
    if lab.ndim == 4 and lab.shape[-1] == 1:
        lab = lab[..., 0]
    return np.eye(num_classes, dtype=np.float64)[lab.astype(np.int64)]


def dice_loss_micro(*, is_y_true_sparse: bool, num_classes: int, **kwargs):
    def _dice_loss_micro(y_true, y_pred, smooth=1e-05):
        if is_y_true_sparse:
            y_true = _one_hot(y_true, num_classes)
        y_true_f = np.asarray(y_true, np.float64).ravel()
        y_pred_f = np.asarray(y_pred, np.float64).ravel()
        score = (2.0 * np.sum(y_true_f * y_pred_f) + smooth) / (np.sum(y_true_f) + np.sum(y_pred_f) + smooth)
        return 1.0 - score

    _dice_loss_micro.oct_loss = "dice_loss_micro"
    return _dice_loss_micro


def dice_loss_macro(*, is_y_true_sparse: bool, num_classes: int, **kwargs):
    def _dice_loss_macro(y_true, y_pred, smooth=1e-05):
        if is_y_true_sparse:
            y_true = _one_hot(y_true, num_classes)
        y_true = np.asarray(y_true, np.float64)
        y_pred = np.asarray(y_pred, np.float64)
        reduce_axis = tuple(range(1, y_pred.ndim - 1))
        intersection = np.sum(y_true * y_pred, axis=reduce_axis)
        denominator = np.sum(y_true, axis=reduce_axis) + np.sum(y_pred, axis=reduce_axis)
        score = (2.0 * intersection + smooth) / (denominator + smooth)
        return 1.0 - np.mean(score)

    _dice_loss_macro.oct_loss = "dice_loss_macro"
    return _dice_loss_macro


FOCAL_EPS = 1e-7   # keras backend epsilon: probabilities are clipped to [eps, 1 - eps] before the logarithm


def focal_dice_loss(*, num_classes: int, gamma: float = 2, class_weight=None, focal_loss_weight: float = 0.5,
                    dice_macro: bool = True, **kwargs):
    """``focal_dice_loss`` / ``SparseCategoricalFocalDiceLoss`` (reference custom_losses.py:98-178), sparse labels:
    ``w * sum_px[cw[y] (1-p_y)^gamma (-log p_y)] / size(y_true) + (1 - w) * dice_loss_{macro|micro}``."""
    cw = None if class_weight is None else np.asarray(class_weight, np.float64)
    if cw is not None and cw.shape != (num_classes,):
        raise ValueError(f"class_weight must have {num_classes} entries")
    dice_fn = (dice_loss_macro if dice_macro else dice_loss_micro)(is_y_true_sparse=True, num_classes=num_classes)

    def _focal_dice_loss(y_true, y_pred):
        lab = np.asarray(y_true)
        lab = (lab[..., 0] if (lab.ndim == 4 and lab.shape[-1] == 1) else lab).astype(np.int64)
        p = np.asarray(y_pred, np.float64)
        py = np.clip(np.take_along_axis(p, lab[..., None], axis=-1)[..., 0], FOCAL_EPS, 1.0 - FOCAL_EPS)
        w = 1.0 if cw is None else cw[lab]
        focal = np.sum(w * (1.0 - py) ** gamma * -np.log(py)) / lab.size
        return focal_loss_weight * focal + (1.0 - focal_loss_weight) * dice_fn(y_true, y_pred)

    _focal_dice_loss.oct_loss = "focal_dice_loss"
    _focal_dice_loss.oct_focal = {"gamma": float(gamma), "class_weight": None if cw is None else cw.tolist(),
                                  "focal_loss_weight": float(focal_loss_weight), "dice_macro": bool(dice_macro)}
    return _focal_dice_loss


def _out_of_scope(name):
    def factory(**kwargs):
        raise NotImplementedError(f"loss '{name}' is outside the accelerated path (only the Dice losses and "
                                  "focal_dice_loss are implemented; see DESIGN.md section 7)")
    return factory


custom_loss_objects = {
    "bce_dice_loss": {"function": _out_of_scope("bce_dice_loss"), "takes_sparse": False},
    "dice_loss_micro": {"function": dice_loss_micro, "takes_sparse": False},
    "dice_loss_macro": {"function": dice_loss_macro, "takes_sparse": False},
    "focal_loss": {"function": _out_of_scope("focal_loss"), "takes_sparse": True},
    "bce_focal_loss": {"function": _out_of_scope("bce_focal_loss"), "takes_sparse": False},
    "focal_dice_loss": {"function": focal_dice_loss, "takes_sparse": True},
}
