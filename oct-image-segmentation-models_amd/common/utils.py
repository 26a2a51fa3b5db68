"""Post-step and housekeeping helpers of the reference's ``common/utils.py`` (:19-176), numpy only."""
from __future__ import annotations

import datetime
import hashlib
import json
import logging as log
from pathlib import Path
from typing import Tuple

import numpy as np


def get_timestamp():
    return datetime.datetime.now().strftime("%Y-%m-%d_%H_%M_%S")


def md5(file_path: Path) -> str:
    log.info(f"Calculating md5 of file: {file_path}")
    h = hashlib.md5()
    with open(file_path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 24), b""):
            h.update(chunk)
    return h.hexdigest()


def to_categorical(y, num_classes: int) -> np.ndarray:
    """keras.utils.to_categorical: a trailing axis of size 1 is dropped, result float32 (Appendix B.8)."""
    y = np.asarray(y, dtype="int64")
    shape = y.shape
    if shape and shape[-1] == 1 and len(shape) > 1:
        shape = shape[:-1]
    out = np.zeros((y.size, num_classes), dtype=np.float32)
    out[np.arange(y.size), y.ravel()] = 1.0
    return out.reshape(shape + (num_classes,))


def load_model_and_config(model_path: Path, **kwargs):
    """Load a model saved by ``Model.save`` plus its sibling ``model_config.json``
    (reference: utils.py:27-70; the MLflow branch is out of scope)."""
    from ..models.engine_model import load_model
    if kwargs.get("mlflow_tracking_uri"):
        raise NotImplementedError("MLflow model loading is outside the accelerated path")
    model_path = Path(model_path)
    loaded_model = load_model(model_path)
    with open(model_path.parent / Path("model_config.json"), "r") as config_file:
        model_config = json.load(config_file)
    return loaded_model, model_config


def convert_maps_uint8(prob_maps):
    prob_maps *= 255
    return prob_maps.astype("uint8")


def perform_argmax(predictions, bin=True):
    """(n,H,W,C) probabilities -> [argmax (n,H,W), categorical (n,C,H,W)] (utils.py:80-112, channels_last)."""
    num_maps = predictions.shape[3]
    argmax_pred = np.argmax(predictions, axis=3)
    if bin:
        categorical_pred = np.transpose(to_categorical(argmax_pred, num_maps), axes=(0, 3, 1, 2))
    else:
        categorical_pred = np.transpose(predictions, axes=(0, 3, 1, 2))
    return [argmax_pred, categorical_pred]


def labels_to_categorical(label_maps: np.ndarray, num_classes: int) -> np.ndarray:
    """(n,H,W) class maps (e.g. the device arg-max) -> categorical (n,C,H,W) float32, as perform_argmax(bin=True)."""
    return np.transpose(to_categorical(label_maps, num_classes), axes=(0, 3, 1, 2))


def convert_predictions_to_maps_semantic(categorical_pred, bg_ilm=True, bg_csi=False):
    """Vertical-gradient boundary maps, uint8 (n, C-1, H, W) (utils.py:115-168)."""
    num_samples, num_maps, img_height, img_width = categorical_pred.shape
    boundary_maps = np.zeros((num_samples, num_maps - 1, img_height, img_width), dtype="uint8")
    for sample_ind in range(num_samples):
        for map_ind in range(1, num_maps):
            flip = (map_ind == 1 and bg_ilm is True) or (map_ind == num_maps - 1 and bg_csi is True)
            cur_map = categorical_pred[sample_ind, map_ind - 1 if flip else map_ind, :, :]
            grad_map = np.gradient(cur_map, axis=0)
            if flip:
                grad_map = -grad_map
            grad_map[grad_map < 0] = 0
            grad_map *= 2
            grad_map -= np.roll(grad_map, -1, axis=0)
            grad_map[grad_map < 0] = 0
            boundary_maps[sample_ind, map_ind - 1, :, :] = convert_maps_uint8(grad_map)
    return boundary_maps


def create_area_mask(image_shape: tuple, segs) -> np.ndarray:
    """Boundaries -> stacked-region mask (dataset_construction.py:654-708, channels_last).  ``image_shape`` is
    (width, height[, channels]) of the TRANSPOSED image the graph search works on; regions do not include the
    boundary pixel that ends them."""
    mask_shape = image_shape[:-1] if len(image_shape) == 3 else image_shape
    mask = np.zeros(mask_shape, dtype="uint8")
    image_width, image_height = mask_shape[0], mask_shape[1]
    if len(image_shape) == 3:
        mask = np.expand_dims(mask, axis=-1)
    segs = np.array(segs)
    for col in range(image_width):
        for seg_ind in range(len(segs)):
            seg = segs[seg_ind, col]
            if np.isnan(seg) or seg == 0:
                found_rep = False
                for rep_ind in range(seg_ind + 1, len(segs)):
                    rep_seg = segs[rep_ind, col]
                    if not np.isnan(rep_seg) and not rep_seg == 0:
                        found_rep = True
                        segs[seg_ind, col] = rep_seg
                        break
                if found_rep is False:
                    segs[seg_ind, col] = image_height
        for seg_ind in range(len(segs)):
            cur_seg = segs[seg_ind, col]
            if seg_ind == 0:
                mask[col, 0:cur_seg] = seg_ind
            else:
                mask[col, segs[seg_ind - 1, col]:cur_seg] = seg_ind
        mask[col, segs[len(segs) - 1, col]:] = len(segs)
    return mask
