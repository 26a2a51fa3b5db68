"""``evaluate_model``: the reference's evaluation workflow
(oct_image_segmentation_models/evaluation/evaluation.py:73-448, savers :451-700, aggregation :722-941).

The forward pass is batched on the GPU (device arg-max, 1 B/px back to the host) instead of one
``predict`` call per image; everything after it -- one-hot, boundary maps, Dice
metrics, optional graph search, per-image result files, dataset aggregates -- is the reference's host logic
re-stated.  Under ``torchrun`` the test set is sharded by contiguous index range (no collective); rank 0
aggregates.  PNG plots and surface-distance metrics are out of scope."""
from __future__ import annotations

import logging as log
import os
import time
import warnings
from pathlib import Path
from typing import List, Optional

import numpy as np

from .. import parallel
from ..common import (EVALUATION_METRIC_AVERAGE_SURFACE_DISTANCE, EVALUATION_METRIC_DICE_CLASSES,
                      EVALUATION_METRIC_DICE_MACRO, EVALUATION_METRIC_DICE_MICRO,
                      EVALUATION_METRIC_HAUSDORFF_DISTANCE, custom_metrics, dataset_loader as dl, h5io)
from ..common import utils as common_utils
from ..min_path_processing import graph_search, utils
from ..models import get_model_class
from ..min_path_processing.pool import SegmentPool
from .evaluation_parameters import EvaluationParameters
from .pipeline import BatchedPredictor

EVALUATION_RESULTS_FILENAME = "evaluation_results.hdf5"
GS_EVALUATION_RESULTS_FILENAME = "gs_evaluation_results.hdf5"
OVERALL_EVALUATION_RESULTS_FILENAME_HDF5 = "overall_evaluation_results.hdf5"
OVERALL_EVALUATION_RESULTS_FILENAME_CSV = "overall_evaluation_results.csv"


class EvaluationOutput:
    def __init__(self, image, image_name, image_segments, image_output_dir, predicted_labels, categorical_pred,
                 boundary_maps, gs_pred_segs, errors, mean_abs_err, mean_err, abs_err_sd, err_sd,
                 dice_classes=None, dice_macro=None, dice_micro=None) -> None:
        self.image = image
        self.image_name = image_name
        self.image_segments = image_segments
        self.image_output_dir = image_output_dir
        self.predicted_labels = predicted_labels
        self.categorical_pred = categorical_pred
        self.boundary_maps = boundary_maps
        self.gs_pred_segs = gs_pred_segs
        self.errors = errors
        self.mean_abs_err = mean_abs_err
        self.mean_err = mean_err
        self.abs_err_sd = abs_err_sd
        self.err_sd = err_sd
        self.dice_classes, self.dice_macro, self.dice_micro = dice_classes, dice_macro, dice_micro


def _dice_metrics(metrics, num_classes, label_onehot_hw, categorical_pred, transposed=False):
    """Dice classes / macro / micro exactly as evaluation.py:175-208 (and :335-375 for the graph-search maps,
    where both operands live in the transposed (W,H) frame)."""
    axes = (2, 1, 0) if transposed else (2, 0, 1)
    label_class_first = np.expand_dims(np.transpose(label_onehot_hw, axes=axes), axis=0)
    dc = dm = dmi = None
    if EVALUATION_METRIC_DICE_CLASSES in metrics:
        dc = custom_metrics.soft_dice_class(label_class_first, categorical_pred)
    if EVALUATION_METRIC_DICE_MACRO in metrics:
        f = custom_metrics.dice_coef_macro(is_y_true_sparse=False, num_classes=num_classes)
        lab = np.expand_dims(np.transpose(label_onehot_hw, axes=[1, 0, 2]) if transposed else label_onehot_hw, axis=0)
        dm = np.array(f(lab, np.transpose(categorical_pred, axes=[0, 2, 3, 1])))
    if EVALUATION_METRIC_DICE_MICRO in metrics:
        f = custom_metrics.dice_coef_micro(is_y_true_sparse=False, num_classes=num_classes)
        dmi = np.array(f(label_class_first, categorical_pred))
    return dc, dm, dmi


def evaluate_model(eval_params: EvaluationParameters) -> List[EvaluationOutput]:
    for m in (EVALUATION_METRIC_AVERAGE_SURFACE_DISTANCE, EVALUATION_METRIC_HAUSDORFF_DISTANCE):
        if m in eval_params.metrics:
            log.error(f"Metric '{m}' needs the un-vendored surface-distance package and is outside the accelerated path.")
            exit(1)
    rank, _, _ = parallel.init()
    world = parallel.world_size()

    data = dl.open_dataset(eval_params.test_dataset_path)
    eval_images, eval_labels, eval_image_names = dl.load_testing_data(data)
    n_images = eval_images.shape[0]
    if not eval_image_names:
        eval_image_names = [Path(f"image_{i}") for i in range(n_images)]
    eval_image_output_dirs = [eval_params.save_foldername / Path(f"image_{i}") for i in range(n_images)]

    eval_segments = np.swapaxes(utils.generate_boundary(np.squeeze(eval_labels, axis=3), axis=1), 0, 1)
    num_classes = eval_params.num_classes
    if rank == 0:
        os.makedirs(eval_params.save_foldername, exist_ok=True)
        save_eval_config_file(eval_params)
    parallel.barrier()

    try:
        model_class = get_model_class(eval_params.loaded_model.name)
    except ValueError as e:
        log.error(e)
        exit(1)
    model_class(**eval_params.model_config)  # validates the stored config exactly as the reference does

    lo, hi = parallel.shard_range(n_images, rank, world)
    eval_outputs: List[EvaluationOutput] = []
    bs = max(1, min(int(eval_params.batch_size), max(hi - lo, 1)))
    # BASELINE configs[4] path (evaluation/pipeline.py): the worker pool for the host min-path post-process is started
    # BEFORE the first GPU call of this process; the forward is a hipGraph replay at the configured batch with pinned,
    # double-buffered uint8 upload / download; a batch's graph search runs on the pool, not image by image on one core
    pool = None
    if eval_params.graph_search and hi > lo:
        pool = SegmentPool(eval_images.shape[1:3], eval_params.gsgrad, getattr(eval_params, "gs_workers", None))
    batches = ()
    if hi > lo and eval_images.dtype == np.uint8:
        engine = eval_params.loaded_model._ensure_engine(bs, False)
        batches = BatchedPredictor(engine, bs, want_maps=True, bg_ilm=True, bg_csi=False).run(eval_images[lo:hi])
    elif hi > lo:      # non-uint8 datasets: x / 255 on the host (Model.predict_labels), same outputs, no overlap
        def _plain():
            for r0 in range(0, hi - lo, bs):
                r1 = min(r0 + bs, hi - lo)
                lm, dm = eval_params.loaded_model.predict_labels(eval_images[lo + r0:lo + r1], batch_size=bs, want_maps=True,
                                                                 bg_ilm=True, bg_csi=False)
                yield r0, r1, lm, dm
        batches = _plain()
    t_prev = time.time()
    for rb0, rb1, label_maps, dev_maps in batches:
        b0, b1 = lo + rb0, lo + rb1
        predict_time = (time.time() - t_prev) / (b1 - b0)
        gs_batch = pool.segment(dev_maps, eval_segments[b0:b1]) if pool is not None else None
        for ind in range(b0, b1):
            eval_image, eval_image_name = eval_images[ind], eval_image_names[ind]
            eval_seg, eval_image_output_dir = eval_segments[ind], eval_image_output_dirs[ind]
            eval_label = common_utils.to_categorical(eval_labels[ind], num_classes)        # (H,W,C)
            os.makedirs(eval_image_output_dir, exist_ok=True)
            predicted_labels = label_maps[ind - b0:ind - b0 + 1].astype(np.int64)          # (1,H,W)
            categorical_pred = common_utils.labels_to_categorical(predicted_labels, num_classes)
            boundary_maps = dev_maps[ind - b0:ind - b0 + 1]   # == convert_predictions_to_maps_semantic(categorical_pred), on device
            dice_classes, dice_macro, dice_micro = _dice_metrics(eval_params.metrics, num_classes, eval_label, categorical_pred)

            predicted_labels = np.squeeze(predicted_labels, axis=0)
            categorical_pred = np.squeeze(categorical_pred, axis=0)
            boundary_maps = np.squeeze(boundary_maps, axis=0)
            _save_image_evaluation_results(eval_params, eval_image, eval_image_name, predicted_labels, categorical_pred,
                                           eval_label, eval_seg, dice_classes, dice_macro, dice_micro, predict_time,
                                           eval_image_output_dir)

            gs_pred_segs = errors = mean_abs_err = mean_err = abs_err_sd = err_sd = None
            if eval_params.graph_search:
                eval_image_t = np.transpose(eval_image, axes=[1, 0, 2])
                start_graph_time = time.time()
                gs_pred_segs, errors = gs_batch[ind - b0]          # == graph_search.segment_maps(boundary_maps_t, eval_seg, grid)
                reconstructed_maps = common_utils.create_area_mask(eval_image_t.shape, gs_pred_segs)
                reconstructed_maps = np.expand_dims(common_utils.to_categorical(reconstructed_maps, num_classes), axis=0)
                [gs_eval_label, reconstructed_maps] = common_utils.perform_argmax(reconstructed_maps)
                gs_dc, gs_dm, gs_dmi = _dice_metrics(eval_params.metrics, num_classes, eval_label, reconstructed_maps,
                                                     transposed=True)
                gs_eval_label = np.transpose(np.squeeze(gs_eval_label))
                graph_time = time.time() - start_graph_time
                mean_abs_err, mean_err, abs_err_sd, err_sd = graph_search.calculate_overall_errors(errors)
                _save_graph_based_evaluation_results(eval_params, eval_image_name, gs_eval_label, gs_pred_segs, gs_dc,
                                                     gs_dm, gs_dmi, errors, mean_abs_err, mean_err, abs_err_sd, err_sd,
                                                     graph_time, eval_image_output_dir)
            eval_outputs.append(EvaluationOutput(
                image=eval_image, image_name=eval_image_name, image_segments=eval_seg,
                image_output_dir=eval_image_output_dir, predicted_labels=predicted_labels,
                categorical_pred=categorical_pred, boundary_maps=boundary_maps, gs_pred_segs=gs_pred_segs, errors=errors,
                mean_abs_err=mean_abs_err, mean_err=mean_err, abs_err_sd=abs_err_sd, err_sd=err_sd,
                dice_classes=dice_classes, dice_macro=dice_macro, dice_micro=dice_micro))
        t_prev = time.time()
    if pool is not None:
        pool.close()
    parallel.barrier()
    if rank == 0:
        _calc_overall_dataset_errors(eval_params, eval_image_names)
    return eval_outputs


def _save_image_evaluation_results(eval_params, eval_image, image_name, predicted_labels, categorical_pred, eval_labels,
                                   eval_segs, dice_classes, dice_macro, dice_micro, predict_time, output_dir):
    with open(output_dir / "input_image_name.txt", "w") as f:
        f.write(str(image_name))
    np.savetxt(output_dir / Path("predicted_segmentation_map.csv"), predicted_labels, fmt="%d", delimiter=",")
    ds = {}
    if eval_params.save_params.categorical_pred is True:
        ds["categorical_pred"] = categorical_pred.astype("uint8")
    if eval_params.save_params.predicted_labels is True:
        ds["predicted_segmentation_map"] = predicted_labels.astype("uint8")
    ds["raw_image"] = eval_image.astype("uint8")
    eval_labels = np.argmax(eval_labels, axis=2)
    ds["eval_labels"] = eval_labels.astype("uint8")
    np.savetxt(output_dir / Path("ground_truth_segmentation_map.csv"), eval_labels, fmt="%d", delimiter=",")
    ds["raw_segs"] = eval_segs.astype("uint16")
    if dice_classes is not None:
        ds[EVALUATION_METRIC_DICE_CLASSES] = np.squeeze(dice_classes).astype("float64")
    if dice_macro is not None:
        ds[EVALUATION_METRIC_DICE_MACRO] = np.expand_dims(dice_macro, axis=0).astype("float64")
    if dice_micro is not None:
        ds[EVALUATION_METRIC_DICE_MICRO] = np.expand_dims(dice_micro, axis=0).astype("float64")
    attrs = {"model_filename": np.array(str(eval_params.model_path), dtype="S1000"),
             "image_name": np.array(str(image_name), dtype="S1000"),
             "timestamp": np.array(common_utils.get_timestamp(), dtype="S1000"),
             "predict_time": np.array(predict_time)}
    h5io.save(output_dir / Path(EVALUATION_RESULTS_FILENAME), ds, attrs)


def _save_graph_based_evaluation_results(eval_params, image_name, gs_eval_label, gs_pred_segs, gs_dice_classes,
                                         gs_dice_macro, gs_dice_micro, errors, mean_abs_err, mean_err, abs_err_sd,
                                         err_sd, graph_time, output_dir):
    np.savetxt(output_dir / Path("gs_boundaries.csv"), gs_pred_segs, delimiter=",", fmt="%d")
    np.savetxt(output_dir / Path("gs_predicted_segmentation_map.csv"), gs_eval_label, fmt="%d", delimiter=",")
    ds = {"gs_pred_segs": gs_pred_segs.astype("uint16"), "errors": errors.astype("float64"),
          "mean_abs_err": mean_abs_err.astype("float64"), "mean_err": mean_err.astype("float64"),
          "abs_err_sd": abs_err_sd.astype("float64"), "err_sd": err_sd.astype("float64"),
          "gs_predicted_labels": gs_eval_label.astype("uint8")}
    if gs_dice_classes is not None:
        ds[EVALUATION_METRIC_DICE_CLASSES] = np.squeeze(gs_dice_classes).astype("float64")
    if gs_dice_macro is not None:
        ds[EVALUATION_METRIC_DICE_MACRO] = np.expand_dims(gs_dice_macro, axis=0).astype("float64")
    if gs_dice_micro is not None:
        ds[EVALUATION_METRIC_DICE_MICRO] = np.expand_dims(gs_dice_micro, axis=0).astype("float64")
    attrs = {"model_filename": np.array(str(eval_params.model_path), dtype="S1000"),
             "image_name": np.array(str(image_name), dtype="S1000"),
             "timestamp": np.array(common_utils.get_timestamp(), dtype="S1000"),
             "graph_time": np.array(graph_time)}
    h5io.save(output_dir / Path(GS_EVALUATION_RESULTS_FILENAME), ds, attrs)


def save_eval_config_file(eval_params: EvaluationParameters):
    p = eval_params.test_dataset_path
    md5_path = p if Path(p).exists() else Path(str(p) + ".npz")
    attrs = {"model_filename": np.array(str(eval_params.model_path), dtype="S1000"),
             "mlflow_tracking_uri": np.array(str(eval_params.mlflow_tracking_uri), dtype="S1000"),
             "test_dataset_path": np.array(str(eval_params.test_dataset_path), dtype="S1000"),
             "test_dataset_md5": np.array(common_utils.md5(md5_path), dtype="S1000"),
             "gsgrad": np.array(eval_params.gsgrad)}
    h5io.save(eval_params.save_foldername / Path("eval_params.hdf5"), {}, attrs)


def _calc_overall_dataset_errors(eval_params: EvaluationParameters, eval_image_names: List[Path]):
    """Mean / sd over images of every per-image metric, re-read from the per-image result files
    (evaluation.py:722-941): ``inf -> nan``, ``nanmean`` / ``nanstd`` over axis 0; boundary-error statistics."""
    output_dir, metrics = eval_params.save_foldername, eval_params.metrics
    dir_list = [Path(output_dir) / Path(f"image_{i}") for i in range(len(eval_image_names))]

    def stack(files, name):
        return np.concatenate([np.expand_dims(f[name], axis=0) for f in files], axis=0)

    files = [h5io.load(d / Path(EVALUATION_RESULTS_FILENAME)) for d in dir_list]
    gs_files = [h5io.load(d / Path(GS_EVALUATION_RESULTS_FILENAME)) for d in dir_list] if eval_params.graph_search else []
    out = {"image_names": np.array([str(n) for n in eval_image_names], dtype="S1000")}
    lines = []

    def save_metric(metric_name: str, metric: np.ndarray):
        out[metric_name] = metric.copy()
        metric = metric.astype(np.float64)
        metric[metric == np.inf] = np.nan
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", category=RuntimeWarning)
            mean_metric, sd_metric = np.nanmean(metric, axis=0), np.nanstd(metric, axis=0)
        out[f"mean_{metric_name}"], out[f"sd_{metric_name}"] = mean_metric, sd_metric
        lines.append(f"Mean {metric_name}," + ",".join(f"{e:.7f}" for e in np.atleast_1d(mean_metric)))
        lines.append(f"SD {metric_name}," + ",".join(f"{e:.7f}" for e in np.atleast_1d(sd_metric)))

    for m in (EVALUATION_METRIC_DICE_CLASSES, EVALUATION_METRIC_DICE_MACRO, EVALUATION_METRIC_DICE_MICRO):
        if m in metrics:
            save_metric(m, stack(files, m))
    if eval_params.graph_search:
        for m in (EVALUATION_METRIC_DICE_CLASSES, EVALUATION_METRIC_DICE_MACRO, EVALUATION_METRIC_DICE_MICRO):
            if m in metrics:
                save_metric(f"gs_{m}", stack(gs_files, m))
        errors = stack(gs_files, "errors")
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", category=RuntimeWarning)
            mean_abs_errors_samples = np.nanmean(np.abs(errors), axis=2)
            mean_errors_samples = np.nanmean(errors, axis=2)
            out.update({
                "mean_abs_errors_cols": np.nanmean(np.abs(errors), axis=0),
                "mean_abs_errors_samples": mean_abs_errors_samples,
                "mean_abs_errors": np.nanmean(mean_abs_errors_samples, axis=0),
                "sd_abs_errors": np.nanstd(mean_abs_errors_samples, axis=0),
                "median_abs_errors": np.nanmedian(mean_abs_errors_samples, axis=0),
                "sd_abs_errors_samples": np.nanstd(np.abs(errors), axis=2),
                "mean_errors_cols": np.nanmean(errors, axis=0),
                "mean_errors_samples": mean_errors_samples,
                "mean_errors": np.nanmean(mean_errors_samples, axis=0),
                "sd_errors": np.nanstd(mean_errors_samples, axis=0),
                "median_errors": np.nanmedian(mean_errors_samples, axis=0),
                "errors": errors})
        for title, key in (("Mean abs errors", "mean_abs_errors"), ("Mean errors", "mean_errors"),
                           ("Median absolute errors", "median_abs_errors"), ("SD abs errors", "sd_abs_errors"),
                           ("SD errors", "sd_errors")):
            lines.append(f"{title}," + ",".join(f"{e:.7f}" for e in out[key]))
    h5io.save(output_dir / Path(OVERALL_EVALUATION_RESULTS_FILENAME_HDF5), out)
    with open(output_dir / Path(OVERALL_EVALUATION_RESULTS_FILENAME_CSV), "w") as f:
        f.write("\n".join(lines) + "\n")
    return out


eval_model = evaluate_model  # README / north_star alias
