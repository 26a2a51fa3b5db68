"""Batch generator with the reference's input contract
(oct_image_segmentation_models/common/data_generator.py:10-416), re-stated with all three augmentation modes:

* ``X = float32(images / 255)`` of shape (B,H,W,C) -- the reference computes ``u8/255*255/255``
  (data_generator.py:76,239 + models/unet.py:89), within one ulp of this;
* labels are passed through unchanged (one-hot for dense losses, (B,H,W,1) for sparse ones);
* ``len() = floor(total_samples / batch_size)`` (tail dropped); the generator is stateful, ignores the
  ``index`` argument and must be consumed sequentially; ``on_epoch_end`` composes a fresh permutation onto
  the previous one (``sample_shuffle = sample_shuffle[s]``), seeded from the OS unless a seed is given.

Augmentation (reference :140-283): ``aug_fn_args`` is a list of ``(function, arguments)`` pairs
from ``common.augmentation``; mode "all" repeats every image once per augmentation (``total_samples = N * n_augs``),
mode "one" draws one augmentation per sample with ``aug_probs``; ``aug_fly=False`` pre-computes the augmented set.
The pre-computed set is kept in float32 (the reference stores the [0,1] floats into a uint8 array, which zeroes
every image -- a defect not reproduced here).  Augmented batches take the float32 input path of the engine.

Without augmentation, batches are assembled with one fancy-indexing gather instead of the reference's per-sample Python loop
(SURVEY 3.1 hot loop ii).  ``next_batch_u8`` is the engine's fast path: uint8 images + uint8 sparse labels,
so the /255 happens on the GPU while the first conv loads the image."""
from __future__ import annotations

import logging as log
from math import floor
from typing import Callable, List, Optional, Tuple

import numpy as np


class BatchGenerator:
    def __init__(self, images: np.ndarray, labels: np.ndarray, batch_size: int, aug_fn_args: List[Tuple],
                 aug_mode: str, aug_probs: Tuple, aug_fly: bool, preprocess_input_fn: Callable,
                 seed: Optional[int] = None):
        if aug_mode not in ("none", "one", "all"):
            log.error(f"Unrecognized augmentation mode: {aug_mode}. Allowed values: 'none', 'one', 'all'. Exiting...")
            exit(1)
        self.images_u8 = np.ascontiguousarray(images)
        self.labels = labels
        self.batch_size = int(batch_size)
        self.aug_fn_args, self.aug_mode, self.aug_probs, self.aug_fly = aug_fn_args, aug_mode, aug_probs, aug_fly
        self.preprocess_input_fn = preprocess_input_fn
        self.total_full_images = self.images_u8.shape[0]
        self.total_raw_samples = self.total_full_images
        self.total_augs = 0 if aug_mode == "none" else len(aug_fn_args)
        self.total_samples = self.total_raw_samples * self.total_augs if aug_mode == "all" else self.total_raw_samples
        if aug_mode != "none" and self.total_augs == 0:
            raise ValueError("aug_mode '%s' needs at least one augmentation function" % aug_mode)
        if aug_mode == "one" and (len(aug_probs) != self.total_augs or abs(sum(aug_probs) - 1.0) > 1e-6):
            raise ValueError("aug_probs must hold one probability per augmentation and sum to 1")
        self.image_height, self.image_width, self.num_channels = self.images_u8.shape[1:4]
        self.labels_shape = self.labels.shape
        self.batch_labels_shape = (self.batch_size,) + tuple(self.labels_shape[1:])
        self.sample_shuffle = np.arange(self.total_full_images)
        self.num_batches = int(floor(1.0 * self.total_samples / self.batch_size))
        self._rng = np.random.default_rng(seed)  # seed=None: OS entropy, as the reference's np.random.seed()
        self._sparse_cache = None
        self.batch_counter = self.full_counter = self.aug_counter = 0
        self.images = None if aug_mode == "none" else self.images_u8.astype(np.float32) / np.float32(255.0)
        if self.aug_fly is False and self.aug_mode != "none":
            self.aug_images, self.aug_labels = self.setup_augnofly_data()
        self.handle_epoch_end()

    def setup_augnofly_data(self):
        """Pre-computed augmentations: (N, n_augs, H, W, C) float32 images and (N, n_augs, ...) labels."""
        aug_images = np.zeros((self.total_full_images, self.total_augs) + self.images.shape[1:], dtype=np.float32)
        aug_labels = np.zeros((self.total_full_images, self.total_augs) + tuple(self.labels_shape[1:]), dtype=self.labels.dtype)
        for i in range(self.total_full_images):
            for j, (aug_fn, aug_arg) in enumerate(self.aug_fn_args):
                aug_images[i, j], aug_labels[i, j] = aug_fn(self.images[i], self.labels[i], aug_arg)
        return aug_images, aug_labels

    def _next_augmented(self):
        """One (image in [0,1], label) sample with the reference's counter semantics (get_aug_fly / get_aug_nofly)."""
        ind = self.sample_shuffle[self.full_counter]
        if self.aug_mode == "all":
            j = self.aug_counter
            self.aug_counter += 1
            if self.aug_counter == self.total_augs:
                self.aug_counter = 0
                self.full_counter += 1
        else:  # "one"
            j = int(self._rng.choice(np.arange(self.total_augs), p=self.aug_probs))
            self.full_counter += 1
        if self.aug_fly:
            aug_fn, aug_arg = self.aug_fn_args[j]
            img, lab = aug_fn(self.images[ind], self.labels[ind], aug_arg)
        else:
            img, lab = self.aug_images[ind, j], self.aug_labels[ind, j]
        if self.full_counter == self.total_full_images:
            self.full_counter = 0
        return img, lab

    def _next_indices(self) -> np.ndarray:
        idx = np.empty(self.batch_size, dtype=np.int64)
        for k in range(self.batch_size):
            idx[k] = self.sample_shuffle[self.full_counter]
            self.full_counter += 1
            if self.full_counter == self.total_full_images:
                self.full_counter = 0
        self.batch_counter += 1
        if self.batch_counter == self.num_batches:
            self.batch_counter = 0
        return idx

    def get_batch_list(self):
        if self.aug_mode != "none":
            batch_images = np.zeros((self.batch_size,) + self.images.shape[1:], dtype="float32")
            batch_labels = np.zeros(self.batch_labels_shape)
            for k in range(self.batch_size):
                batch_images[k], batch_labels[k] = self._next_augmented()
            self.batch_counter += 1
            if self.batch_counter == self.num_batches:
                self.batch_counter = 0
            return [batch_images, batch_labels]
        idx = self._next_indices()
        batch_images = (self.images_u8[idx].astype(np.float32)) / np.float32(255.0)
        batch_labels = np.asarray(self.labels[idx], dtype=np.float64)  # the reference's label buffer is float64
        return [batch_images, batch_labels]

    def sparse_labels(self) -> np.ndarray:
        """uint8 (N,H,W) class map of the label array (argmax of a one-hot array, or the squeezed sparse map)."""
        if self._sparse_cache is None:
            lab = self.labels
            if lab.ndim == 4 and lab.shape[-1] > 1:
                lab = np.argmax(lab, axis=-1)
            elif lab.ndim == 4:
                lab = lab[..., 0]
            # (no copy for the usual (N,H,W,1) uint8 array: a view -- the copy was 1 GB per 8192 scans, inside the first epoch)
            self._sparse_cache = np.ascontiguousarray(lab, dtype=np.uint8)
        return self._sparse_cache

    def next_batch_u8(self, shard: Optional[Tuple[int, int]] = None):
        """The next GLOBAL batch; with ``shard = (lo, hi)`` only samples lo..hi-1 of it are gathered (a DP rank's slice:
        every rank advances the same shuffled order, none of them assembles the other ranks' scans)."""
        if self.images_u8.dtype != np.uint8:
            raise TypeError(f"next_batch_u8 needs a uint8 image array, the dataset holds {self.images_u8.dtype}; "
                            "use get_batch_list() (float32(images) / 255)")
        idx = self._next_indices()
        if shard is not None:
            idx = idx[shard[0]:shard[1]]
        return self.images_u8[idx], self.sparse_labels()[idx]

    def handle_epoch_end(self):
        self.batch_counter = 0
        self.full_counter = 0
        self.aug_counter = 0
        s = self._rng.permutation(self.total_raw_samples)
        self.sample_shuffle = self.sample_shuffle[s]


class DataGenerator:
    """``keras.utils.Sequence`` duck type: ``__len__``, ``__getitem__``, ``on_epoch_end``."""

    def __init__(self, images: np.ndarray, labels: np.ndarray, batch_size: int, aug_fn_args: List[Tuple],
                 aug_mode: str, aug_probs: Tuple, aug_fly: bool, preprocess_input_fn: Callable,
                 seed: Optional[int] = None):
        # uint8 fast path (the /255 happens on the GPU) only without augmentation AND only for uint8 storage: the
        # reference divides by 255 whatever the dataset's dtype (data_generator.py:76), so any other dtype goes through
        # get_batch_list(), which computes float32(images) / 255 on the host
        self.oct_fast_path = aug_mode == "none" and np.asarray(images).dtype == np.uint8
        self.batch_gen = BatchGenerator(images=images, labels=labels, batch_size=batch_size, aug_fn_args=aug_fn_args,
                                        aug_mode=aug_mode, aug_probs=aug_probs, aug_fly=aug_fly,
                                        preprocess_input_fn=preprocess_input_fn, seed=seed)

    def __len__(self):
        return self.batch_gen.num_batches

    def __getitem__(self, index):
        X, y = self.batch_gen.get_batch_list()
        return X, y

    def next_batch_u8(self, shard: Optional[Tuple[int, int]] = None):
        return self.batch_gen.next_batch_u8(shard)

    @property
    def batch_size(self) -> int:
        return self.batch_gen.batch_size

    def on_epoch_end(self):
        self.batch_gen.handle_epoch_end()

    def get_total_samples(self) -> int:
        return self.batch_gen.total_samples
