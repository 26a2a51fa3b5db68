"""Plugin base class -- same contract as the reference's ``BaseModel``
(oct_image_segmentation_models/models/base_model.py:8-36)."""
from __future__ import annotations

from typing import Callable


class BaseModel:
    def __init__(self, *, input_channels: int, num_classes: int, image_height: int, image_width: int):
        self.input_channels = int(input_channels)
        self.num_classes = int(num_classes)
        self.image_height = int(image_height)
        self.image_width = int(image_width)

    def build_model(self):
        raise NotImplementedError("Must be implemented in subclasses.")

    def get_config(self) -> dict:
        return {
            "input_channels": self.input_channels,
            "num_classes": self.num_classes,
            "image_height": self.image_height,
            "image_width": self.image_width,
        }

    def get_preprocess_input_fn(self) -> Callable:
        raise NotImplementedError("Must be implemented in subclasses.")
