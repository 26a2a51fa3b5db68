"""Min-path boundary delineation on the host (stays on host per north_star).

Restatement of the reference's ``min_path_processing/graph_search.py`` (:5-105 Dijkstra, :108-225 graph
structure, :337-357 appended columns, :360-428 delineation, :479-516 errors, :519-572 ``segment_maps``,
:575-589 overall errors) with identical edge weights and heap tie-breaking, pinned by golden vectors captured
from the reference (``tests/golden/min_path_golden.npz``).  The grid graph is implicit (``GridGraph``) instead
of a Python list of lists rebuilt per image (0.34 s/image in the reference), and the search runs
in ``liboct_minpath.so`` (C++, ``csrc/minpath.cpp``) when built; the pure-Python body is the same algorithm."""
from __future__ import annotations

import ctypes as C
import os
from heapq import heappop, heappush

import numpy as np

_HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB_PATH = os.path.join(_HERE, "liboct_minpath.so")
_lib = None


def _native():
    global _lib
    if _lib is None and os.path.exists(_LIB_PATH):
        l = C.CDLL(_LIB_PATH)
        l.oct_minpath_delineate.restype = C.c_int
        l.oct_minpath_delineate.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        _lib = l
    return _lib


class GridGraph:
    """Implicit ``create_graph_structure`` result: (width+2) x height vertices, index = col + row*graph_width."""

    def __init__(self, shape, max_grad=1):
        self.graph_width = int(shape[0]) + 2
        self.graph_height = int(shape[1])
        self.max_grad = int(max_grad)

    def __len__(self):
        return self.graph_width * self.graph_height

    def __getitem__(self, node):
        gw, gh, mg = self.graph_width, self.graph_height, self.max_grad
        i, j = divmod(int(node), gw)
        right, down = (j + 1) + i * gw, j + (i + 1) * gw
        up = [(j + 1) + (i - g) * gw for g in range(1, mg + 1) if i - g >= 0]
        dn = [(j + 1) + (i + g) * gw for g in range(1, mg + 1) if i + g <= gh - 1]
        if i == gh - 1:
            return [] if j == gw - 1 else [right] + up
        if i == 0:
            if j == gw - 1:
                return [down]
            return [right, down] + dn if j == 0 else [right] + dn
        if j == gw - 1:
            return [down]
        return [right, down] + up + dn if j == 0 else [right] + up + dn


def create_graph_structure(shape, max_grad=1):
    """``shape`` = (width, height) of the transposed image (a 3rd channel entry is ignored, as the reference
    is called with ``eval_image_t.shape``)."""
    return GridGraph(shape, max_grad)


def run_dijkstras(prob_map, start_ind, graph_structure):
    gw = prob_map.shape[0]
    max_ind = prob_map.shape[0] * prob_map.shape[1] - 1
    shortest_paths = [None] * (max_ind + 1)
    candidates_q = [(0, 0, 0, start_ind, 0)]
    add_count = 1
    while candidates_q:
        path_len, _, _, v, a = heappop(candidates_q)
        if shortest_paths[v] is not None:
            continue
        shortest_paths[v] = (path_len, a)
        if v == max_ind:
            break
        cur_col, cur_row = v % gw, v // gw
        cur_prob = prob_map[cur_col][cur_row]
        for i, n in enumerate(graph_structure[v]):
            if shortest_paths[n] is not None:
                continue
            n_col, n_row = n % gw, n // gw
            edge_len = 2 - (cur_prob + prob_map[n_col][n_row])   # the reference's np.max(x, 0) is not a clamp
            prio = 0 if (n_col == cur_col and n_row == cur_row + 1) else i + 1
            heappush(candidates_q, (path_len + edge_len, prio, add_count, n, v))
            add_count += 1
    return [0 if x is None else x for x in shortest_paths]


def append_firstlast_cols(prob_map):
    ones = np.ones((1, prob_map.shape[1]))
    return np.concatenate((ones, prob_map, ones), axis=0)


def delineate_boundary(prob_map, graph_structure):
    prob_map = np.ascontiguousarray(append_firstlast_cols(prob_map), dtype=np.float64)
    map_width, map_height = prob_map.shape
    lib = _native()
    if lib is not None and isinstance(graph_structure, GridGraph):
        delin = np.zeros(map_width - 2, dtype=np.float64)
        rc = lib.oct_minpath_delineate(prob_map.ctypes.data, map_width, map_height, graph_structure.max_grad,
                                       delin.ctypes.data)
        if rc != 0:
            raise RuntimeError("oct_minpath_delineate failed")
        return delin
    shortest_paths = run_dijkstras(prob_map, 0, graph_structure)
    node_ind = map_width * map_height - 1
    node_coord = (node_ind % map_width, node_ind // map_width)
    prev_node_ind = shortest_paths[node_ind][1]
    node_order_coords = []
    while node_coord != (0, 0):
        node_order_coords.append(node_coord)
        node_coord = (prev_node_ind % map_width, prev_node_ind // map_width)
        prev_node_ind = shortest_paths[prev_node_ind][1]
    delin = np.zeros((map_width - 2))
    for coord in node_order_coords:
        if coord[0] != 0 and coord[0] != map_width - 1:
            delin[coord[0] - 1] = coord[1]
    return delin


def calc_errors(prediction, truth):
    truth = np.asarray(truth, dtype=np.float64)
    error = np.asarray(prediction).astype("float64") - truth
    error[np.isnan(truth) | (truth <= 0)] = np.nan
    return error


def segment_maps(prob_maps, truths, graph_structure):
    """uint8 maps (num_maps, width, height) -> (predictions uint16 (num_maps, width), errors float64, maps/255)."""
    prob_maps = prob_maps / 255
    num_maps, width = prob_maps.shape[0], prob_maps.shape[1]
    predictions = np.zeros((num_maps, width), dtype="uint16")
    errors = np.zeros((num_maps, width), dtype="float64")
    for map_ind in range(num_maps):
        prediction = delineate_boundary(prob_maps[map_ind], graph_structure)
        predictions[map_ind, :] = prediction
        if truths is not None:
            errors[map_ind:, ] = calc_errors(prediction, truths[map_ind, :])   # as the reference (Appendix D.5)
    return (predictions, errors, prob_maps)


def calculate_overall_errors(errors):
    with np.errstate(all="ignore"):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", category=RuntimeWarning)
            return [np.nanmean(np.abs(errors), axis=1), np.nanmean(errors, axis=1),
                    np.nanstd(np.abs(errors), axis=1), np.nanstd(errors, axis=1)]
