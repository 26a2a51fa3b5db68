"""``predict``: the reference's prediction workflow
(oct_image_segmentation_models/prediction/prediction.py:48-186, savers :189-329) -- the evaluation stack
without labels/metrics.  Batched device forward with device arg-max; host tail re-stated; PNGs out of scope."""
from __future__ import annotations

import logging as log
import os
import time
from pathlib import Path
from typing import List, Union

import numpy as np

from .. import parallel
from ..common import h5io, utils
from ..evaluation.pipeline import BatchedPredictor
from ..min_path_processing import graph_search  # noqa: F401  (re-exported: callers build graph structures through it)
from ..min_path_processing.pool import SegmentPool
from ..models import get_model_class
from .prediction_parameters import PredictionParams


class PredictionOutput:
    def __init__(self, image: np.ndarray, image_name: Path, image_output_dir: Path, predicted_labels: np.ndarray,
                 categorical_pred: np.ndarray, boundary_maps: np.ndarray, gs_pred_segs: Union[np.ndarray, None]) -> None:
        self.image = image
        self.image_name = image_name
        self.image_output_dir = image_output_dir
        self.predicted_labels = predicted_labels
        self.categorical_pred = categorical_pred
        self.boundary_maps = boundary_maps
        self.gs_pred_segs = gs_pred_segs


def predict(predict_params: PredictionParams) -> List[PredictionOutput]:
    rank, _, _ = parallel.init()
    world = parallel.world_size()
    dataset = predict_params.dataset
    images = np.asarray(dataset.images)
    if rank == 0:
        os.makedirs(predict_params.config_output_dir, exist_ok=True)
        save_predict_config_file(predict_params)
    try:
        model_class = get_model_class(predict_params.loaded_model.name)
    except ValueError as e:
        log.error(e)
        exit(1)
    model_class(**predict_params.model_config)
    num_classes = predict_params.num_classes

    outputs: List[PredictionOutput] = []
    lo, hi = parallel.shard_range(len(images), rank, world)
    bs = max(1, min(int(predict_params.batch_size), max(hi - lo, 1)))
    # same device pipeline as evaluate_model (evaluation/pipeline.py): worker pool first, then the hipGraph replay with
    # pinned double-buffered uint8 transfers; non-uint8 images take the host x / 255 path of Model.predict_labels
    pool = None
    if predict_params.graph_search and hi > lo:
        pool = SegmentPool(images.shape[1:3], 1, getattr(predict_params, "gs_workers", None))
    if hi > lo and images.dtype == np.uint8:
        engine = predict_params.loaded_model._ensure_engine(bs, False)
        batches = BatchedPredictor(engine, bs, want_maps=True, bg_ilm=True, bg_csi=False).run(images[lo:hi])
    else:
        def _plain():
            for r0 in range(0, hi - lo, bs):
                r1 = min(r0 + bs, hi - lo)
                lm, dm = predict_params.loaded_model.predict_labels(images[lo + r0:lo + r1], batch_size=bs, want_maps=True,
                                                                    bg_ilm=True, bg_csi=False)
                yield r0, r1, lm, dm
        batches = _plain()
    t0 = time.time()
    for rb0, rb1, label_maps, dev_maps in batches:
        b0, b1 = lo + rb0, lo + rb1
        predict_time = (time.time() - t0) / (b1 - b0)
        gs_batch = pool.segment(dev_maps, None) if pool is not None else None
        for i in range(b0, b1):
            predict_image, image_name, image_output_dir = images[i], dataset.image_names[i], Path(dataset.image_output_dirs[i])
            os.makedirs(image_output_dir, exist_ok=True)
            log.info(f"Inferring image {i}: {image_name}")
            start_convert_time = time.time()
            predicted_labels = label_maps[i - b0:i - b0 + 1].astype(np.int64)
            categorical_pred = utils.labels_to_categorical(predicted_labels, num_classes)
            boundary_maps = dev_maps[i - b0:i - b0 + 1]   # == convert_predictions_to_maps_semantic(categorical_pred), on device
            convert_time = time.time() - start_convert_time
            predicted_labels = np.squeeze(predicted_labels, axis=0)
            categorical_pred = np.squeeze(categorical_pred, axis=0)
            boundary_maps = np.squeeze(boundary_maps, axis=0)
            save_image_prediction_results(predict_params, predict_image, image_name, predicted_labels, categorical_pred,
                                          boundary_maps, predict_time, convert_time, image_output_dir)
            gs_pred_segs = None
            if predict_params.graph_search:
                predict_image_t = np.transpose(predict_image, axes=[1, 0, 2])
                start_graph_time = time.time()
                gs_pred_segs = gs_batch[i - b0][0]               # == graph_search.segment_maps(boundary_maps_t, None, grid)
                reconstructed_maps = utils.create_area_mask(predict_image_t.shape, gs_pred_segs)
                reconstructed_maps = np.expand_dims(utils.to_categorical(reconstructed_maps, num_classes), axis=0)
                [gs_prediction_label, reconstructed_maps] = utils.perform_argmax(reconstructed_maps)
                gs_prediction_label = np.transpose(np.squeeze(gs_prediction_label))
                graph_time = time.time() - start_graph_time
                save_graph_based_prediction_results(predict_params, image_name, gs_prediction_label, gs_pred_segs,
                                                    graph_time, image_output_dir)
            outputs.append(PredictionOutput(image=predict_image, image_name=image_name, image_output_dir=image_output_dir,
                                            predicted_labels=predicted_labels, categorical_pred=categorical_pred,
                                            boundary_maps=boundary_maps, gs_pred_segs=gs_pred_segs))
        t0 = time.time()
    if pool is not None:
        pool.close()
    parallel.barrier()
    return outputs


def save_predict_config_file(predict_params: PredictionParams):
    attrs = {"model_filename": np.array(str(predict_params.model_path), dtype="S1000"),
             "error_col_inc_range": np.array((predict_params.col_error_range[0], predict_params.col_error_range[-1]))}
    h5io.save(predict_params.config_output_dir / Path("prediction_params.hdf5"), {}, attrs)


def save_image_prediction_results(pred_params, predict_image, image_name, predicted_labels, categorical_pred,
                                  boundary_maps, predict_time, convert_time, output_dir):
    ds = {}
    if pred_params.save_params.categorical_pred is True:
        ds["categorical_pred"] = categorical_pred.astype("uint8")
    np.savetxt(output_dir / Path("segmentation_map.csv"), predicted_labels, fmt="%d", delimiter=",")
    if pred_params.save_params.predicted_labels is True:
        ds["predicted_labels"] = predicted_labels.astype("uint8")
    if pred_params.save_params.boundary_maps is True:
        ds["boundary_maps"] = boundary_maps.astype("uint8")
    ds["raw_image"] = predict_image.astype("uint8")
    attrs = {"model_filename": np.array(str(pred_params.model_path), dtype="S1000"),
             "image_name": np.array(str(image_name), dtype="S1000"),
             "timestamp": np.array(utils.get_timestamp(), dtype="S1000"),
             "predict_time": np.array(predict_time), "convert_time": convert_time}
    h5io.save(output_dir / Path("prediction_info.hdf5"), ds, attrs)


def save_graph_based_prediction_results(predict_params, image_name, gs_prediction_label, gs_pred_segs, graph_time,
                                        output_dir):
    np.savetxt(output_dir / Path("gs_boundaries.csv"), gs_pred_segs, delimiter=",", fmt="%d")
    np.savetxt(output_dir / Path("gs_segmentation_map.csv"), gs_prediction_label, fmt="%d", delimiter=",")
    ds = {"gs_pred_segs": gs_pred_segs.astype("uint16"), "gs_predicted_labels": gs_prediction_label.astype("uint8")}
    attrs = {"model_filename": np.array(str(predict_params.model_path), dtype="S1000"),
             "image_name": np.array(str(image_name), dtype="S1000"),
             "timestamp": np.array(utils.get_timestamp(), dtype="S1000"), "graph_time": np.array(graph_time)}
    h5io.save(output_dir / Path("graph_search_prediction_info.hdf5"), ds, attrs)
