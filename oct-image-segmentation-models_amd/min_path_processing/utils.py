"""``generate_boundary`` (reference: min_path_processing/utils.py:4-18): label map -> first row of each
region, per column.  Boundaries belong to the first pixel of the "next" region."""
import numpy as np


def generate_boundary(img_array, axis=0):
    num_classes = int(np.amax(img_array))
    return np.array([np.argmax(img_array == i, axis=axis) for i in range(1, num_classes + 1)])
