"""Process pool for the host min-path post-process (BASELINE configs[4]; SURVEY 8d "host min_path_processing timed
1 core and N-process pool").  One task = one B-scan: its (C-1) boundary maps -> ``graph_search.segment_maps``
(reference min_path_processing/graph_search.py:519-572; callers evaluation.py:289-315, prediction.py:134-143).

Workers are SPAWNED (fresh interpreters: safe next to a process that already holds a GPU context) and import only
numpy + the native ``liboct_minpath.so`` -- never torch.  Results are exactly ``segment_maps`` of the same inputs, in
input order.

Spawned workers re-import ``__main__``: a script that calls ``evaluate_model`` / ``predict`` at top level without an
``if __name__ == "__main__":`` guard would re-run itself in every worker and the parent would wait forever.  So: the
pool is only started from a real main process (inside a spawned child the work runs inline), worker start-up is probed
once with a timeout (workers that cannot start -- no guard, ``liboct_minpath.so`` not loadable -- make the pool fall
back to inline execution with a warning instead of hanging), ``get()`` of a batch has a timeout with the same inline
fallback, and under ``torchrun`` the default worker count is divided by LOCAL_WORLD_SIZE."""
from __future__ import annotations

import atexit
import logging
import multiprocessing as mp
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import graph_search

log = logging.getLogger(__name__)
_graph = None
START_TIMEOUT_S = float(os.environ.get("OCT_GS_POOL_START_TIMEOUT", "60"))     # worker start-up probe
TASK_TIMEOUT_S = float(os.environ.get("OCT_GS_POOL_TASK_TIMEOUT", "600"))       # one batch of maps


def _worker_init(shape_t: Tuple[int, ...], gsgrad: int) -> None:
    global _graph
    _graph = graph_search.create_graph_structure(shape_t, gsgrad)     # implicit grid: built once per worker


_views = {}          # worker side: path -> read-only memmap of a batch of maps (a few most recent batches)


def _batch_view(path: str, shape: Tuple[int, ...]) -> np.ndarray:
    v = _views.get(path)
    if v is None:
        while len(_views) >= 4:
            _views.pop(next(iter(_views)))
        v = _views[path] = np.memmap(path, dtype=np.uint8, mode="r", shape=tuple(shape))
    return v


def _segment_one(task):
    if len(task) == 4:                                                  # (file of the whole batch in /dev/shm, index, shape, truths)
        path, i, shape, truths = task
        maps_hw = _batch_view(path, shape)[i]
    else:
        maps_hw, truths = task                                          # (C-1, H, W) uint8, (C-1, W) or None
    maps_t = np.ascontiguousarray(np.transpose(maps_hw, axes=[0, 2, 1]))
    pred, errors, _ = graph_search.segment_maps(maps_t, truths, _graph)
    return pred, errors


def default_workers() -> int:
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    n = max(1, min(16, n))                                              # a GPU box's CPU share for one GPU is 16
    try:                                                                # one pool per rank: share the cores
        n = max(1, n // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1"))))
    except ValueError:
        pass
    return n


def _probe(_):
    return os.getpid()


def _in_spawned_child() -> bool:
    return mp.current_process().name != "MainProcess" or mp.parent_process() is not None


_SHM_DIR = "/dev/shm"
_live_files = set()          # batch files not yet removed (removed when their batch is collected; all of them at exit)


def _remove_file(path: Optional[str]) -> None:
    if path:
        _live_files.discard(path)
        try:
            os.unlink(path)
        except OSError:
            pass


def _remove_all_files() -> None:
    for p in list(_live_files):
        _remove_file(p)


atexit.register(_remove_all_files)


class SegmentPool:
    """``segment(maps, truths)``: (n, C-1, H, W) uint8 boundary maps (+ optional (n, C-1, W) truths) ->
    [(predictions uint16 (C-1, W), errors float64 (C-1, W)), ...].  ``workers <= 1`` runs inline."""

    _seq = 0

    def __init__(self, image_shape_hw: Sequence[int], gsgrad: int = 1, workers: Optional[int] = None):
        self.shape_t = (int(image_shape_hw[1]), int(image_shape_hw[0]))   # the graph search works on the (W, H) view
        self.gsgrad = int(gsgrad)
        self.workers = default_workers() if workers is None else int(workers)
        self._pool = None
        if self.workers > 1 and _in_spawned_child():
            log.warning("SegmentPool created inside a worker process: running the min-path post-process inline")
            self.workers = 1
        if self.workers > 1:
            pool = mp.get_context("spawn").Pool(self.workers, initializer=_worker_init, initargs=(self.shape_t, self.gsgrad))
            try:                        # workers that die while bootstrapping are respawned forever: find out now
                pool.map_async(_probe, range(self.workers)).get(timeout=START_TIMEOUT_S)
                self._pool = pool
            except Exception as e:      # mp.TimeoutError, or the initializer's own error
                pool.terminate(); pool.join()
                log.warning("min-path worker pool did not start (%s: %s) -- is the calling script guarded by "
                            "`if __name__ == '__main__':`? -- running inline", type(e).__name__, e)
                self.workers = 1
        if self._pool is None:
            _worker_init(self.shape_t, self.gsgrad)

    def segment_async(self, maps: np.ndarray, truths: Optional[np.ndarray] = None):
        tasks = [(maps[i], None if truths is None else truths[i]) for i in range(maps.shape[0])]
        if self._pool is None:
            res = [_segment_one(t) for t in tasks]
            return _Done(res)
        chunk = max(1, len(tasks) // (4 * self.workers))
        # The maps of a device batch are 33 MB at 128 x 2 x 256 x 512: pickled through the pool's pipe they cost the
        # submitting thread 20-30 ms per batch (the thread that also drives the GPU pipeline).  Written ONCE to a file in
        # /dev/shm (page cache: a memcpy) and mapped read-only by the workers, a task is a path and an index.
        path = None
        if maps.dtype == np.uint8 and maps.nbytes >= (1 << 20) and os.path.isdir(_SHM_DIR) and os.access(_SHM_DIR, os.W_OK):
            try:
                SegmentPool._seq += 1
                path = os.path.join(_SHM_DIR, f"oct_gs_{os.getpid()}_{SegmentPool._seq}.u8")
                np.ascontiguousarray(maps).tofile(path)
                _live_files.add(path)
            except OSError:
                path = None
        if path is None:
            return _Pending(self, self._pool.map_async(_segment_one, tasks, chunksize=chunk), tasks)
        ptasks = [(path, i, tuple(maps.shape), None if truths is None else truths[i]) for i in range(maps.shape[0])]
        return _Pending(self, self._pool.map_async(_segment_one, ptasks, chunksize=chunk), tasks, path)

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


class _Pending:
    """A batch in flight on the pool; ``get`` falls back to inline execution if the pool does not answer in time."""

    def __init__(self, owner: SegmentPool, handle, tasks, path: Optional[str] = None):
        self._owner, self._h, self._tasks, self._path = owner, handle, tasks, path

    def get(self, timeout: Optional[float] = None):
        try:
            return self._h.get(timeout=TASK_TIMEOUT_S if timeout is None else timeout)
        except mp.TimeoutError:
            log.warning("min-path worker pool did not answer within the timeout: finishing this batch inline")
            if _graph is None:
                _worker_init(self._owner.shape_t, self._owner.gsgrad)
            return [_segment_one(t) for t in self._tasks]
        finally:
            _remove_file(self._path); self._path = None      # (workers that still map it keep their pages until they move on)

    def __del__(self):
        _remove_file(getattr(self, "_path", None))
