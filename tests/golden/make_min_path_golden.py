"""Generates tests/golden/min_path_golden.npz by running the REAL reference implementation of the host
post-process (``oct_image_segmentation_models.min_path_processing``, numpy + heapq only -- it imports in the
build container) on seeded synthetic inputs.  The reference itself never travels; only the vectors do.

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference python tests/golden/make_min_path_golden.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oct_image_segmentation_models.min_path_processing import graph_search as gs  # noqa: E402  (reference)
from oct_image_segmentation_models.min_path_processing import utils as gsu  # noqa: E402  (reference)
from oracle import unet_numpy as on  # noqa: E402  (boundary-map construction; utils.py of the reference needs TF)


def adjacency(graph):
    adj = -np.ones((len(graph), 4), np.int32)
    for i, nb in enumerate(graph):
        adj[i, :len(nb)] = nb
    return adj


def main():
    out = {}
    rng = np.random.default_rng(2024)
    cases = [("a", 32, 48, 4), ("b", 40, 64, 3), ("c", 24, 20, 5)]
    for tag, H, W, C in cases:
        _, labels = on.synth_scans(2, H, W, C, seed=hash(tag) % 1000 if False else ord(tag))
        lab = labels[..., 0]
        truths = np.swapaxes(gsu.generate_boundary(lab, axis=1), 0, 1)        # (n, C-1, W)  evaluation.py:86-88
        out[f"{tag}_labels"] = lab
        out[f"{tag}_generate_boundary"] = truths
        graph = gs.create_graph_structure((W, H))
        out[f"{tag}_graph_adj"] = adjacency(graph)
        cat = np.transpose(np.eye(C, dtype=np.float32)[lab], (0, 3, 1, 2))
        clean = on.convert_predictions_to_maps_semantic(cat)                  # (n, C-1, H, W) uint8
        noisy = clean.copy()
        salt = rng.uniform(size=noisy.shape) < 0.02
        noisy[salt] = rng.integers(0, 256, salt.sum()).astype(np.uint8)
        grey = rng.integers(0, 256, clean.shape).astype(np.uint8)             # dense random weights: many ties broken
        empty = np.zeros_like(clean)
        for kind, maps in (("clean", clean), ("noisy", noisy), ("grey", grey), ("empty", empty)):
            maps_t = np.transpose(maps[0], (0, 2, 1)).copy()                  # (C-1, W, H)  evaluation.py:292
            preds, errors, norm = gs.segment_maps(maps_t, truths[0], graph)
            out[f"{tag}_{kind}_maps_t"] = maps_t
            out[f"{tag}_{kind}_pred"] = preds
            out[f"{tag}_{kind}_errors"] = errors
            stats = gs.calculate_overall_errors(errors)
            out[f"{tag}_{kind}_stats"] = np.stack(stats)
        # a truth row containing zeros / nan exercises calc_errors' invalid handling
        t = truths[0].astype(np.float64).copy(); t[0, :3] = 0; t[-1, 5] = np.nan
        out[f"{tag}_calc_errors_truth"] = t
        out[f"{tag}_calc_errors"] = np.stack([gs.calc_errors(out[f"{tag}_clean_pred"][k], t[k]) for k in range(C - 1)])
    g2 = gs.create_graph_structure((6, 5), max_grad=2)
    out["grad2_graph_adj"] = -np.ones((len(g2), 6), np.int32)
    for i, nb in enumerate(g2):
        out["grad2_graph_adj"][i, :len(nb)] = nb
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "min_path_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in list(out.items())[:6]}, "...", len(out), "arrays")


if __name__ == "__main__":
    main()
