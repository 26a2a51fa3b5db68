"""``SaveEpochInfo``: rolling per-epoch statistics file, same datasets as the reference
(oct_image_segmentation_models/training/training_callbacks.py:11-75)."""
from __future__ import annotations

import time
from pathlib import Path

import numpy as np

from ..common import h5io
from ..models.engine_model import Callback


class SaveEpochInfo(Callback):
    def __init__(self, save_folder: Path, train_params):
        self.train_losses, self.train_accs, self.val_losses, self.val_accs, self.epoch_times = [], [], [], [], []
        self.start_epoch_time = self.start_time = self.train_time = -1
        self.acc_name = train_params.metric
        self.loss_name = train_params.loss
        self.save_folder = Path(save_folder)
        self.num_epochs = train_params.epochs

    def on_train_begin(self, logs=None):
        self.train_losses, self.train_accs, self.val_losses, self.val_accs, self.epoch_times = [], [], [], [], []
        self.start_time = time.time()

    def on_train_end(self, logs=None):
        self.train_time = time.time() - self.start_time

    def on_epoch_begin(self, epoch, logs=None):
        self.start_epoch_time = time.time()

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        nan = float("nan")
        self.train_losses.append(logs.get("loss", nan))
        self.train_accs.append(logs.get(self.acc_name, nan))
        self.val_losses.append(logs.get("val_loss", nan))
        self.val_accs.append(logs.get("val_" + self.acc_name, nan))
        self.epoch_times.append(time.time() - self.start_epoch_time)
        h5io.save(self.save_folder / Path("stats_epoch{:02d}.hdf5".format(epoch + 1)), {
            "train_acc": np.array(self.train_accs), "val_acc": np.array(self.val_accs),
            "train_loss": np.array(self.train_losses), "val_loss": np.array(self.val_losses),
            "epoch_time": np.array(self.epoch_times)})
        h5io.remove(self.save_folder / Path("stats_epoch{:02d}.hdf5".format(epoch)))
