"""``Dataset`` holder used by ``predict`` (oct_image_segmentation_models/common/dataset.py:10-32)."""
from __future__ import annotations

from pathlib import Path
from typing import List

import numpy as np


class Dataset:
    def __init__(self, images: np.ndarray, image_names: List[Path], image_output_dirs: List[Path]):
        if not (len(images) == len(image_names) == len(image_output_dirs)):
            raise ValueError("images, image_names and image_output_dirs must have the same length")
        self.images = images
        self.image_names = image_names
        self.image_output_dirs = image_output_dirs
