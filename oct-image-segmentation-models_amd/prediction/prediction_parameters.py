"""``PredictionParams`` / ``PredictionSaveParams`` with the reference's constructor contract
(oct_image_segmentation_models/prediction/prediction_parameters.py:12-63)."""
from __future__ import annotations

from pathlib import Path
from typing import Union

from ..common import utils
from ..common.dataset import Dataset


class PredictionSaveParams:
    def __init__(self, predicted_labels: bool = True, categorical_pred: bool = False, png_images: bool = True,
                 boundary_maps: bool = True) -> None:
        self.predicted_labels = predicted_labels
        self.categorical_pred = categorical_pred
        self.png_images = png_images      # accepted for compatibility; PNG plotting is out of scope
        self.boundary_maps = boundary_maps


class PredictionParams:
    def __init__(self, model_path: Path, mlflow_tracking_uri: Union[str, None], mlflow_run_uuid: Union[str, None],
                 dataset: Dataset, config_output_dir: Path, save_params: PredictionSaveParams,
                 graph_search: bool = False, trim_maps: bool = False, trim_ref_ind: int = 0,
                 trim_window: tuple = (0, 0), col_error_range: tuple = None, batch_size: int = 32) -> None:
        self.model_path = Path(model_path)
        self.mlflow_tracking_uri = mlflow_tracking_uri
        self.mlflow_run_uuid = mlflow_run_uuid
        self.dataset = dataset
        self.loaded_model, self.model_config = utils.load_model_and_config(
            self.model_path, mlflow_tracking_uri=mlflow_tracking_uri, mlflow_run_uuid=mlflow_run_uuid)
        self.num_classes = self.loaded_model.output.shape[-1]
        self.config_output_dir = Path(config_output_dir)
        self.save_params = save_params
        self.graph_search = graph_search
        self.trim_maps = trim_maps
        self.trim_ref_ind = trim_ref_ind
        self.trim_window = trim_window
        self.batch_size = batch_size
        self.col_error_range = col_error_range
        if col_error_range is None:
            self.col_error_range = range(dataset.images[0].shape[1])  # image_width
