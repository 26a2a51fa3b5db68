"""``EvaluationParameters`` / ``EvaluationSaveParams`` with the reference's constructor contract
(oct_image_segmentation_models/evaluation/evaluation_parameters.py:12-85); the model is loaded in the
constructor, as there."""
from __future__ import annotations

import logging as log
from pathlib import Path
from typing import List, Optional

from ..common import EVALUATION_METRICS, utils


class EvaluationSaveParams:
    def __init__(self, predicted_labels: bool = True, categorical_pred: bool = False, png_images: bool = True,
                 boundary_maps: bool = True) -> None:
        self.predicted_labels = predicted_labels
        self.categorical_pred = categorical_pred
        self.png_images = png_images      # accepted for compatibility; PNG plotting is out of scope
        self.boundary_maps = boundary_maps


class EvaluationParameters:
    def __init__(self, model_path: Path, mlflow_tracking_uri: Optional[str], mlflow_run_uuid: Optional[str],
                 test_dataset_path: Path, save_foldername: Path, save_params: EvaluationSaveParams,
                 graph_search: bool, metrics: List[str], gsgrad=1, dice_errors: bool = True, binarize: bool = True,
                 bg_ilm: bool = True, bg_csi: bool = False, batch_size: int = 32):
        self.model_path = Path(model_path)
        self.mlflow_tracking_uri = mlflow_tracking_uri
        self.mlflow_run_uuid = mlflow_run_uuid
        self.test_dataset_path = Path(test_dataset_path)
        self.binarize = binarize
        self.save_params = save_params
        self.graph_search = graph_search
        if not set(metrics).issubset(EVALUATION_METRICS):
            log.error(f"Some of the provided metrics are invalid. Provided metrics: {metrics}.")
            exit(1)
        self.metrics = metrics
        self.gsgrad = gsgrad
        self.dice_errors = dice_errors
        self.bg_ilm = bg_ilm
        self.bg_csi = bg_csi
        self.save_foldername = Path(save_foldername)
        self.batch_size = batch_size   # extension: device batch (the reference predicts one image per call)
        self.loaded_model, self.model_config = utils.load_model_and_config(
            self.model_path, mlflow_tracking_uri=mlflow_tracking_uri, mlflow_run_uuid=mlflow_run_uuid)
        self.num_classes = self.loaded_model.output.shape[-1]
