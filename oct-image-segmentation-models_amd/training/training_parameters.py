"""``TrainingParams`` with the reference's constructor contract
(oct_image_segmentation_models/training/training_parameters.py:11-135)."""
from __future__ import annotations

import logging as log
from pathlib import Path
from typing import List, Tuple, Union

from ..common import AUG_MODES
from ..common import augmentation as aug


class TrainingParams:
    def __init__(self, model_architecture: Union[str, None], training_dataset_path: Path,
                 initial_model: Union[Path, None], results_location: Path, opt_con, loss: str, metric: str,
                 epochs: int, batch_size: int, model_hyperparameters: dict = {}, opt_params: dict = {},
                 loss_fn_kwargs: dict = {}, augmentations: List[dict] = [], aug_mode: str = "none",
                 aug_probs: Tuple = (), aug_fly: bool = False, aug_val: bool = True, shuffle: bool = True,
                 model_save_best: bool = True, model_save_monitor=("val_acc", "max"),
                 class_weight: Union[list, str, None] = None, channels_last: bool = True,
                 early_stopping: bool = True, restore_best_weights: bool = True, patience: int = 50,
                 seed: Union[int, None] = None):
        if (model_architecture is None and initial_model is None) or (
                model_architecture is not None and initial_model is not None):
            log.error("Either 'model_architecture' or 'initial_model' need to be provided in the `config.json`.")
            exit(1)
        self.model_architecture = model_architecture
        self.model_hyperparameters = model_hyperparameters
        self.training_dataset_path = Path(training_dataset_path)
        self.initial_model = initial_model
        self.results_location = Path(results_location)
        self.opt_con = opt_con
        self.opt_params = opt_params
        self.loss = loss
        self.loss_fn_kwargs = loss_fn_kwargs
        self.metric = metric
        self.epochs = epochs
        self.batch_size = batch_size
        if aug_mode not in AUG_MODES:
            log.error(f"Augmentation mode: '{aug_mode}' is not supported.")
            exit(1)
        self.aug_mode = aug_mode
        self.aug_fn_args = []
        for augmentation in augmentations:
            aug_fn = aug.augmentation_map.get(augmentation["name"])
            if aug_fn is None:
                log.error(f"Augmentation: '{augmentation['name']}' is not supported.")
                exit(1)
            self.aug_fn_args.append((aug_fn, augmentation.get("arguments", {})))
        self.augmentations = augmentations
        self.aug_probs = aug_probs
        self.aug_fly = aug_fly
        self.aug_val = aug_val
        self.shuffle = shuffle
        self.model_save_best = model_save_best
        self.model_save_monitor = model_save_monitor
        self.class_weight = class_weight
        self.channels_last = channels_last
        self.early_stopping = early_stopping
        self.restore_best_weights = restore_best_weights
        self.patience = patience
        self.seed = seed  # extension: reproducible shuffling / init (the reference is unseeded)
        if self.model_save_monitor[0] == "val_acc":
            self.model_save_monitor = ["val_" + self.metric, model_save_monitor[1]]
