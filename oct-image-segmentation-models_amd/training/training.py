"""``train_model``: the reference's training workflow
(oct_image_segmentation_models/training/training.py:135-408) on the HIP engine.

Same control flow and on-disk contract: read ``train_/val_`` images+labels, ``num_classes =
len(np.unique(train_labels))``, look the loss and metric up by name, build the model through the plugin
registry, checkpoint on the monitored validation metric, rolling ``stats_epochNN`` file, early stopping,
``model_config.json`` + ``training_params`` file.  MLflow calls are dropped (out of scope); ``initial_model``
works here (the reference's branch calls a non-existent ``utils.load_model``).
Under ``torchrun`` every rank runs this function; rank 0 writes the files."""
from __future__ import annotations

import json
import logging as log
import os
from pathlib import Path

import numpy as np

from .. import parallel
from ..common import custom_losses, custom_metrics, data_generator as data_gen, dataset_loader, h5io, utils
from ..models import get_model_class
from ..models.engine_model import EarlyStopping, ModelCheckpoint
from . import training_callbacks
from .training_parameters import TrainingParams


def save_training_params_file(save_foldername: Path, model_summary: str, model_config: dict,
                              training_dataset_md5: str, class_weight, timestamp, train_params: TrainingParams, opt):
    """``model_config.json`` + ``training_params.hdf5`` attributes (training.py:39-132)."""
    with open(save_foldername / Path("model_config.json"), "w") as config_file:
        json.dump(model_config, config_file)
    attrs = {
        "timestamp": np.array(timestamp, dtype="S100"),
        "model_summary": np.array(model_summary.encode("ascii", "replace")[:4000]),
        "train_dataset_md5": np.array(training_dataset_md5, dtype="S1000"),
        "epochs": train_params.epochs,
        "loss_name": np.array(train_params.loss, dtype="S1000"),
        "metric_name": np.array(train_params.metric, dtype="S1000"),
        "class_weight": np.array("None" if class_weight is None else "array", dtype="S1000"),
        "metric": np.array(train_params.metric, dtype="S100"),
        "loss": np.array(train_params.loss, dtype="S100"),
        "batch_size": train_params.batch_size,
        "shuffle": train_params.shuffle,
        "aug_mode": np.array(train_params.aug_mode, dtype="S100"),
        "optimizer": np.array(train_params.opt_con.__name__, dtype="S100"),
    }
    for key, val in opt.get_config().items():
        attrs["opt_param: " + key] = np.bytes_(str(val)) if isinstance(val, (dict, str)) else val
    datasets = {} if class_weight is None else {"class_weight": np.asarray(class_weight)}
    h5io.save(save_foldername / Path("training_params.hdf5"), datasets, attrs)


def _balanced_class_weight(labels: np.ndarray) -> np.ndarray:
    """sklearn ``compute_class_weight("balanced")``: n_samples / (n_classes * bincount)."""
    classes, counts = np.unique(labels, return_counts=True)
    return labels.size / (len(classes) * counts.astype(np.float64))


def train_model(training_params: TrainingParams, mlflow_params=None):
    if mlflow_params is not None:
        log.warning("MLflow tracking is outside the accelerated path; mlflow_params is ignored")
    rank, _, _ = parallel.init()

    training_dataset_path = training_params.training_dataset_path
    data = dataset_loader.open_dataset(training_dataset_path)
    train_images, train_labels = dataset_loader.load_training_data(data)
    val_images, val_labels = dataset_loader.load_validation_data(data)

    num_classes = len(np.unique(train_labels))
    log.info(f"Detected {num_classes} classes")
    _, image_height, image_width, input_channels = train_images.shape
    log.info(f"Detected input image dimensions (h x w): {image_height} x {image_width}.")
    log.info(f"Detected {input_channels} input channels.")
    log.info(f"Number of devices: {parallel.world_size()}")

    optimizer = training_params.opt_con(**training_params.opt_params)

    loss = custom_losses.custom_loss_objects.get(training_params.loss)
    if loss is None:
        log.error(f"Loss '{training_params.loss}' not found. Exiting...")
        exit(1)
    if training_params.class_weight == "balanced":
        c_weight = _balanced_class_weight(np.concatenate((train_labels, val_labels)))
    elif type(training_params.class_weight) == list:
        c_weight = np.array(training_params.class_weight)
    else:
        c_weight = None  # as in the reference, the weights are recorded but never reach the loss (Appendix D.2)
    sparse_labels = loss["takes_sparse"]
    try:
        loss_fn = loss["function"](num_classes=num_classes, is_y_true_sparse=sparse_labels,
                                   **training_params.loss_fn_kwargs)
    except NotImplementedError as e:
        log.error(e)
        exit(1)

    metric = custom_metrics.training_monitor_metric_objects.get(training_params.metric)
    if metric is None:
        log.error(f"Metric '{training_params.metric}' not found. Exiting...")
        exit(1)
    metric_fn = metric(sparse_labels, num_classes)

    # The reference one-hot encodes the label arrays here for dense losses (training.py:225-227); the engine
    # consumes sparse uint8 labels and one-hot encodes on the fly, so the arrays stay sparse (32x less memory).
    training_dataset_md5 = utils.md5(training_dataset_path) if Path(training_dataset_path).exists() else \
        utils.md5(Path(str(training_dataset_path) + ".npz"))

    model_architecture = training_params.model_architecture
    if training_params.initial_model:
        log.info(f"Starting training from model: {training_params.initial_model}")
        model, model_config = utils.load_model_and_config(Path(training_params.initial_model))
        model_architecture = model.name
        try:
            model_container = get_model_class(model_architecture)(**model_config)
        except ValueError as e:
            log.error(e)
            exit(1)
    else:
        log.info(f"Starting training from scratch {model_architecture} model")
        try:
            model_class = get_model_class(model_architecture)
        except ValueError as e:
            log.error(e)
            exit(1)
        model_container = model_class(input_channels=input_channels, num_classes=num_classes,
                                      image_height=image_height, image_width=image_width,
                                      **training_params.model_hyperparameters)
        model = model_container.build_model()
        if training_params.seed is not None:
            model.config["seed"] = int(training_params.seed)
    model.compile(optimizer=optimizer, loss=loss_fn, metrics=[metric_fn])

    batch_size = training_params.batch_size
    aug_val_mode = training_params.aug_mode if training_params.aug_val else "none"

    monitor = training_params.model_save_monitor
    timestamp = parallel.broadcast_object(utils.get_timestamp())   # one results folder for all ranks
    save_foldername = training_params.results_location / Path(timestamp + "_" + model_architecture)
    if rank == 0:
        os.makedirs(save_foldername, exist_ok=True)
    parallel.barrier()

    savemodel = ModelCheckpoint(filepath=save_foldername / Path("model_epoch{epoch:02d}.hdf5"),
                                save_best_only=training_params.model_save_best, monitor=monitor[0], mode=monitor[1])
    callbacks_list = [savemodel]
    if rank == 0:
        callbacks_list.append(training_callbacks.SaveEpochInfo(save_folder=save_foldername, train_params=training_params))
    if training_params.early_stopping:
        callbacks_list.append(EarlyStopping(monitor=f"val_{training_params.metric}", mode="max",
                                            patience=training_params.patience,
                                            restore_best_weights=training_params.restore_best_weights))

    model_summary = []
    model.summary(print_fn=lambda line: model_summary.append(line))
    if rank == 0:
        save_training_params_file(save_foldername, "\n".join(model_summary), model_container.get_config(),
                                  training_dataset_md5, c_weight, timestamp, training_params, optimizer)

    # every rank draws the SAME global batches (its slice of each is taken in Model._host_batch): an unseeded
    # run gets one OS-entropy seed from rank 0, not one per rank
    seed = parallel.shared_seed(training_params.seed) if parallel.world_size() > 1 else training_params.seed
    train_gen = data_gen.DataGenerator(train_images, train_labels, batch_size, training_params.aug_fn_args, training_params.aug_mode,
                                       training_params.aug_probs, training_params.aug_fly,
                                       model_container.get_preprocess_input_fn(), seed=seed)
    val_gen = data_gen.DataGenerator(val_images, val_labels, batch_size,
                                     training_params.aug_fn_args if aug_val_mode != "none" else [], aug_val_mode,
                                     training_params.aug_probs if aug_val_mode != "none" else (),
                                     training_params.aug_fly if aug_val_mode != "none" else False,
                                     model_container.get_preprocess_input_fn(),
                                     seed=None if seed is None else seed + 1)

    for name, gen in (("training", train_gen), ("validation", val_gen)):
        if batch_size > gen.get_total_samples():
            log.error(f"The batch size ({batch_size}) cannot be larger than the number of {name} samples "
                      f"({gen.get_total_samples()})")
            exit(1)
    log.info(f"Train generator total number of samples: {train_gen.get_total_samples()}")
    log.info(f"Validation generator total number of samples: {val_gen.get_total_samples()}")

    history = model.fit(x=train_gen, validation_data=val_gen, epochs=training_params.epochs,
                        callbacks=callbacks_list, verbose=1)
    return SimpleTrainingResult(model, save_foldername, history, savemodel.saved)


class SimpleTrainingResult:
    """What the reference leaves on disk, also handed back to the caller (the reference returns None)."""

    def __init__(self, model, save_foldername, history, checkpoints):
        self.model, self.save_foldername, self.history, self.checkpoints = model, save_foldername, history.history, checkpoints


train = train_model  # north_star alias
