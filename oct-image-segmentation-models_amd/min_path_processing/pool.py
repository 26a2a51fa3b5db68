"""Process pool for the host min-path post-process (BASELINE configs[4]; SURVEY 8d "host min_path_processing timed
1 core and N-process pool").  One task = one B-scan: its (C-1) boundary maps -> ``graph_search.segment_maps``
(reference min_path_processing/graph_search.py:519-572; callers evaluation.py:289-315, prediction.py:134-143).

Workers are SPAWNED (fresh interpreters: safe next to a process that already holds a GPU context) and import only
numpy + the native ``liboct_minpath.so`` -- never torch.  Results are exactly ``segment_maps`` of the same inputs, in
input order."""
from __future__ import annotations

import multiprocessing as mp
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import graph_search

_graph = None


def _worker_init(shape_t: Tuple[int, ...], gsgrad: int) -> None:
    global _graph
    _graph = graph_search.create_graph_structure(shape_t, gsgrad)     # implicit grid: built once per worker


def _segment_one(task):
    maps_hw, truths = task                                              # (C-1, H, W) uint8, (C-1, W) or None
    maps_t = np.ascontiguousarray(np.transpose(maps_hw, axes=[0, 2, 1]))
    pred, errors, _ = graph_search.segment_maps(maps_t, truths, _graph)
    return pred, errors


def default_workers() -> int:
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(16, n))                                           # a GPU box's CPU share for one GPU is 16


class SegmentPool:
    """``segment(maps, truths)``: (n, C-1, H, W) uint8 boundary maps (+ optional (n, C-1, W) truths) ->
    [(predictions uint16 (C-1, W), errors float64 (C-1, W)), ...].  ``workers <= 1`` runs inline."""

    def __init__(self, image_shape_hw: Sequence[int], gsgrad: int = 1, workers: Optional[int] = None):
        self.shape_t = (int(image_shape_hw[1]), int(image_shape_hw[0]))   # the graph search works on the (W, H) view
        self.gsgrad = int(gsgrad)
        self.workers = default_workers() if workers is None else int(workers)
        self._pool = None
        if self.workers > 1:
            self._pool = mp.get_context("spawn").Pool(self.workers, initializer=_worker_init,
                                                      initargs=(self.shape_t, self.gsgrad))
        else:
            _worker_init(self.shape_t, self.gsgrad)

    def segment_async(self, maps: np.ndarray, truths: Optional[np.ndarray] = None):
        tasks = [(maps[i], None if truths is None else truths[i]) for i in range(maps.shape[0])]
        if self._pool is None:
            res = [_segment_one(t) for t in tasks]
            return _Done(res)
        chunk = max(1, len(tasks) // (4 * self.workers))
        return self._pool.map_async(_segment_one, tasks, chunksize=chunk)

    def segment(self, maps: np.ndarray, truths: Optional[np.ndarray] = None) -> List[Tuple[np.ndarray, np.ndarray]]:
        return self.segment_async(maps, truths).get()

    def close(self) -> None:
        if self._pool is not None:
            self._pool.close(); self._pool.join(); self._pool = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


class _Done:
    def __init__(self, res): self._res = res
    def get(self, timeout=None): return self._res
