"""Loss registry with the reference's names and factory signatures
(oct_image_segmentation_models/common/custom_losses.py:47-81,230-255).

On the accelerated path the Dice sums are fused into the HIP head kernel, so ``Model.compile`` selects the
loss by the ``oct_loss`` tag of the callable; the callables themselves are numpy restatements usable on host
arrays (e.g. for evaluation code).  Non-Dice entries of the reference registry (third-party ``focal-loss``
package, BCE mixes) are out of scope and raise."""
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


def _out_of_scope(name):
    def factory(**kwargs):
        raise NotImplementedError(f"loss '{name}' is outside the accelerated path (only the Dice losses are "
                                  "implemented; see DESIGN.md section 7)")
    return factory


custom_loss_objects = {
    "bce_dice_loss": {"function": _out_of_scope("bce_dice_loss"), "takes_sparse": False},
    "dice_loss_micro": {"function": dice_loss_micro, "takes_sparse": False},
    "dice_loss_macro": {"function": dice_loss_macro, "takes_sparse": False},
    "focal_loss": {"function": _out_of_scope("focal_loss"), "takes_sparse": True},
    "bce_focal_loss": {"function": _out_of_scope("bce_focal_loss"), "takes_sparse": False},
    "focal_dice_loss": {"function": _out_of_scope("focal_dice_loss"), "takes_sparse": True},
}
