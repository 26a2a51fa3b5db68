"""Metric registry with the reference's names
(oct_image_segmentation_models/common/custom_metrics.py:18-100).  The training-monitor Dice coefficients are
computed on the device by the head kernel (selected through the ``oct_metric`` tag); the numpy bodies here
serve the evaluation code.  Surface-distance metrics (un-vendored google-deepmind/surface-distance) are out of
scope."""
from __future__ import annotations

import numpy as np

from . import TRAINING_MONITOR_METRIC_DICE_MACRO, TRAINING_MONITOR_METRIC_DICE_MICRO
from .custom_losses import _one_hot


def dice_coef_micro(is_y_true_sparse: bool, num_classes: int):
    def _dice_coef_micro(y_true, y_pred):
        if is_y_true_sparse:
            y_true = _one_hot(y_true, num_classes)
        y_true_f = np.asarray(y_true, np.float32).ravel()
        y_pred_f = (np.asarray(y_pred, np.float32).ravel() > 0.5).astype(np.float32)
        with np.errstate(invalid="ignore", divide="ignore"):   # no epsilon in the reference: 0/0 -> nan
            return np.float32(2.0) * np.sum(y_true_f * y_pred_f) / (np.sum(y_true_f) + np.sum(y_pred_f))

    _dice_coef_micro.__name__ = "dice_coef_micro"
    _dice_coef_micro.oct_metric = "dice_coef_micro"
    return _dice_coef_micro


def dice_coef_macro(is_y_true_sparse: bool, num_classes: int):
    def _dice_coef_macro(y_true, y_pred, eps=1e-05):
        if is_y_true_sparse:
            y_true = _one_hot(y_true, num_classes)
        y_true = np.asarray(y_true, np.float32)
        y_pred = (np.asarray(y_pred) > 0.5).astype(np.float32)
        reduce_axis = tuple(range(1, y_pred.ndim - 1))
        intersection = np.sum(y_true * y_pred, axis=reduce_axis)
        denominator = np.sum(y_true, axis=reduce_axis) + np.sum(y_pred, axis=reduce_axis)
        return np.mean((2.0 * intersection + eps) / (denominator + eps))

    _dice_coef_macro.__name__ = "dice_coef_macro"
    _dice_coef_macro.oct_metric = "dice_coef_macro"
    return _dice_coef_macro


training_monitor_metric_objects = {
    TRAINING_MONITOR_METRIC_DICE_MACRO: dice_coef_macro,
    TRAINING_MONITOR_METRIC_DICE_MICRO: dice_coef_micro,
}


def soft_dice_class(y_true, y_pred, eps=1e-5):
    """(b, c, X, Y...) inputs -> per-class Dice (b, c)."""
    axes = tuple(range(2, len(y_pred.shape)))
    intersect = np.sum(y_pred * y_true, axis=axes)
    denom = np.sum(y_pred + y_true, axis=axes)
    return ((2.0 * intersect) + eps) / (denom + eps)


def average_surface_distance(*args, **kwargs):
    raise NotImplementedError("surface-distance metrics are outside the accelerated path (DESIGN.md section 7)")


def hausdorff_distance(*args, **kwargs):
    raise NotImplementedError("surface-distance metrics are outside the accelerated path (DESIGN.md section 7)")
