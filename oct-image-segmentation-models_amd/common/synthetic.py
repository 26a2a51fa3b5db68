"""Seeded synthetic OCT-like B-scans for benchmarks and plumbing runs: there is no network
for datasets, so ``bench.py`` and the examples feed the engine with these.

Labels follow the area-mask convention of the reference's ``create_area_mask``
(common/dataset_construction.py:694-706): C-1 boundaries split each A-scan (image column) into C stacked
regions numbered 0..C-1 from the top; images are a per-region grey level plus Gaussian speckle."""
from __future__ import annotations

import numpy as np


def make_scans(n: int, height: int, width: int, num_classes: int, seed: int = 1234):
    """Returns (images uint8 (n,H,W,1), labels uint8 (n,H,W,1)); every class occurs in every scan."""
    rng = np.random.default_rng(seed)
    nb = num_classes - 1
    col = np.arange(width, dtype=np.float64)
    row = np.arange(height, dtype=np.float64)[:, None]
    level = np.linspace(40.0, 200.0, num_classes)
    images = np.empty((n, height, width, 1), np.uint8)
    labels = np.empty((n, height, width, 1), np.uint8)
    for i in range(n):
        depth = np.sort(rng.uniform(0.15, 0.85, nb)) * height
        amp = rng.uniform(0.01, 0.05, nb) * height
        wavelength = rng.uniform(width / 12.0, width / 3.0, nb)
        phase = rng.uniform(0.0, 2.0 * np.pi, nb)
        curves = depth[:, None] + amp[:, None] * np.sin(col[None, :] / wavelength[:, None] + phase[:, None])
        curves = np.clip(np.sort(curves, axis=0), 1.0, height - 2.0)
        area = (row[None, :, :] >= curves[:, None, :]).sum(axis=0).astype(np.uint8)
        speckle = rng.normal(0.0, 25.0, (height, width))
        images[i, :, :, 0] = np.clip(level[area] + speckle, 0, 255).astype(np.uint8)
        labels[i, :, :, 0] = area
    return images, labels
