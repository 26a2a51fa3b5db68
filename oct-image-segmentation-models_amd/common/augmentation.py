"""Augmentation functions with the reference's names, signatures and registry
(oct_image_segmentation_models/common/augmentation.py:43-103).

``flip`` is exact.  ``add_noise`` follows the documented semantics of ``skimage.util.random_noise`` (third-party,
not installed here: parity with skimage's RNG stream is unpinned) for the modes gaussian / speckle / s&p / salt /
pepper on images in [0, 1]; the result is clipped to [0, 1] as skimage does for unsigned input."""
from __future__ import annotations

import numpy as np

_rng = np.random.default_rng()


def seed(value) -> None:
    """Seed the module RNG used by ``add_noise`` (the reference is unseeded)."""
    global _rng
    _rng = np.random.default_rng(value)


def no_aug(image, mask, _aug_args, desc_only=False):
    if desc_only is False:
        return image, mask
    return "no aug"


def flip_aug(image, mask, aug_args, desc_only=False):
    flip_type = aug_args["flip_type"]
    if flip_type == "up-down":
        axis = 0
    elif flip_type == "left-right":
        axis = 1
    else:
        raise ValueError(f"flip_type must be 'up-down' or 'left-right', got {flip_type!r}")
    if desc_only is False:
        aug_image = np.flip(image, axis=axis)
        aug_mask = np.flip(mask, axis=axis) if mask is not None else None
        return aug_image, aug_mask
    return "flip aug: " + flip_type


def add_noise_aug(image, mask, aug_args, desc_only=False):
    if desc_only is not False:
        return "add noise: " + str(aug_args)
    mode = aug_args["mode"]
    mean = aug_args.get("mean", 0.0)
    var = aug_args.get("variance", 0.01)
    img = np.asarray(image, dtype=np.float64)
    if mode == "gaussian":
        out = img + _rng.normal(mean, var ** 0.5, img.shape)
    elif mode == "speckle":
        out = img + img * _rng.normal(mean, var ** 0.5, img.shape)
    elif mode in ("s&p", "salt", "pepper"):
        amount = aug_args.get("amount", 0.05)
        svp = {"s&p": aug_args.get("salt_vs_pepper", 0.5), "salt": 1.0, "pepper": 0.0}[mode]
        out = img.copy()
        flipped = _rng.random(img.shape) <= amount
        salted = _rng.random(img.shape) <= svp
        out[flipped & salted] = 1.0
        out[flipped & ~salted] = 0.0
    else:
        raise ValueError(f"add_noise mode {mode!r} is not supported")
    return np.clip(out, 0.0, 1.0), mask


augmentation_map = {
    "add_noise": add_noise_aug,
    "flip": flip_aug,
    "no_augmentation": no_aug,
}


def normalize(x):
    x = np.asarray(x)
    return (x - x.min()) / (np.ptp(x))
