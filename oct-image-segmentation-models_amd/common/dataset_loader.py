"""Dataset key contract of the reference (oct_image_segmentation_models/common/dataset_loader.py:9-33):
``train_images``/``train_labels``/``val_images``/``val_labels`` and ``test_images``/``test_labels``/
``test_images_source``.  ``hdf5_data_file`` is anything indexable by key (an ``h5py.File`` or the dict
returned by ``h5io.load``)."""
from __future__ import annotations

from pathlib import Path
from typing import List, Tuple

import numpy as np

from . import h5io


def open_dataset(path):
    """Open a dataset file: real HDF5 through h5py when available, else the .npz twin with the same keys."""
    return h5io.load(path)


def load_training_data(hdf5_data_file):
    return hdf5_data_file["train_images"][:], hdf5_data_file["train_labels"][:]


def load_validation_data(hdf5_data_file):
    return hdf5_data_file["val_images"][:], hdf5_data_file["val_labels"][:]


def load_testing_data(hdf5_data_file) -> Tuple[np.ndarray, np.ndarray, List[Path]]:
    test_images = hdf5_data_file["test_images"][:]
    test_labels = hdf5_data_file["test_labels"][:]
    src = hdf5_data_file.get("test_images_source")
    names = [] if src is None else [Path(x.decode("ascii") if isinstance(x, (bytes, np.bytes_)) else str(x)) for x in src]
    return test_images, test_labels, names
