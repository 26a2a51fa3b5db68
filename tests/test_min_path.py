"""Host post-process (min_path_processing) pinned by golden vectors captured from the REAL reference
(tests/golden/make_min_path_golden.py).  Bit-exact: integer/index work."""
import os

import numpy as np
import pytest

import __graft_entry__ as ge

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "min_path_golden.npz"))
CASES = [("a", 32, 48, 4), ("b", 40, 64, 3), ("c", 24, 20, 5)]


@pytest.fixture(scope="module")
def gs():
    ge.build()
    from oct_image_segmentation_models_amd.min_path_processing import graph_search
    assert graph_search._native() is not None, "liboct_minpath.so was not built"
    return graph_search


def test_generate_boundary_matches_reference():
    from oct_image_segmentation_models_amd.min_path_processing import utils
    for tag, H, W, C in CASES:
        got = np.swapaxes(utils.generate_boundary(G[f"{tag}_labels"], axis=1), 0, 1)
        assert np.array_equal(got, G[f"{tag}_generate_boundary"])


def test_graph_structure_matches_reference(gs):
    for tag, H, W, C in CASES:
        g = gs.create_graph_structure((W, H))
        adj = G[f"{tag}_graph_adj"]
        assert len(g) == adj.shape[0]
        for v in range(len(g)):
            assert g[v] == [int(n) for n in adj[v] if n >= 0], v
    g2 = gs.create_graph_structure((6, 5), max_grad=2)
    for v in range(len(g2)):
        assert g2[v] == [int(n) for n in G["grad2_graph_adj"][v] if n >= 0]


@pytest.mark.parametrize("native", [True, False])
@pytest.mark.parametrize("kind", ["clean", "noisy", "grey", "empty"])
def test_segment_maps_matches_reference(gs, kind, native, monkeypatch):
    if not native:
        monkeypatch.setattr(gs, "_native", lambda: None)   # pure-Python body of the same algorithm
    for tag, H, W, C in CASES:
        if not native and tag == "b" and kind in ("grey", "empty"):
            continue  # keep the CPU suite short: the Python body is slow on dense maps
        graph = gs.create_graph_structure((W, H))
        preds, errors, norm = gs.segment_maps(G[f"{tag}_{kind}_maps_t"], G[f"{tag}_generate_boundary"][0], graph)
        assert preds.dtype == np.uint16 and np.array_equal(preds, G[f"{tag}_{kind}_pred"]), (tag, kind)
        assert np.array_equal(errors, G[f"{tag}_{kind}_errors"], equal_nan=True)
        stats = np.stack(gs.calculate_overall_errors(errors))
        assert np.allclose(stats, G[f"{tag}_{kind}_stats"], rtol=0, atol=1e-12, equal_nan=True)


def test_calc_errors_invalid_truths(gs):
    for tag, H, W, C in CASES:
        t = G[f"{tag}_calc_errors_truth"]
        got = np.stack([gs.calc_errors(G[f"{tag}_clean_pred"][k], t[k]) for k in range(C - 1)])
        assert np.array_equal(got, G[f"{tag}_calc_errors"], equal_nan=True)


def test_clean_maps_reproduce_truth_boundaries(gs):
    # property at a larger size than the goldens: a clean boundary map delineates to the truth rows exactly
    from oracle import unet_numpy as on
    from oct_image_segmentation_models_amd.min_path_processing import utils
    from oct_image_segmentation_models_amd.common import utils as cu
    H, W, C = 128, 256, 4
    _, labels = on.synth_scans(1, H, W, C, seed=3)
    lab = labels[..., 0]
    truths = np.swapaxes(utils.generate_boundary(lab, axis=1), 0, 1)[0]
    maps = cu.convert_predictions_to_maps_semantic(cu.labels_to_categorical(lab, C))[0]
    preds, errors, _ = gs.segment_maps(np.transpose(maps, (0, 2, 1)), truths, gs.create_graph_structure((W, H)))
    assert np.array_equal(preds, truths) and np.all(errors == 0)


def test_segment_pool_equals_inline_segment_maps():
    """min_path_processing/pool.py (BASELINE configs[4] host stage): spawned workers return exactly what
    graph_search.segment_maps returns inline, in input order, with and without ground-truth boundaries."""
    from oct_image_segmentation_models_amd.min_path_processing import graph_search
    from oct_image_segmentation_models_amd.min_path_processing.pool import SegmentPool
    n, H, W = 5, 48, 96
    maps = np.zeros((n, 2, H, W), np.uint8)
    truths = np.zeros((n, 2, W))
    for i in range(n):
        for c in range(2):
            rows = (H * (c + 1) // 3 + 3 * np.sin(np.arange(W) / 7.0 + i)).astype(int)
            maps[i, c, rows, np.arange(W)] = 255
            maps[i, c, (rows + 5) % H, np.arange(W)] = 90            # a weaker competing ridge
            truths[i, c] = rows
    grid = graph_search.create_graph_structure((W, H), 1)
    want = [graph_search.segment_maps(np.transpose(maps[i], (0, 2, 1)), truths[i], grid)[:2] for i in range(n)]
    with SegmentPool((H, W), 1, workers=2) as pool:
        got = pool.segment(maps, truths)
        got_async = pool.segment_async(maps[:2]).get()
    with SegmentPool((H, W), 1, workers=1) as solo:
        got1 = solo.segment(maps, truths)
    for i in range(n):
        for g in (got, got1):
            assert np.array_equal(g[i][0], want[i][0]) and np.array_equal(g[i][1], want[i][1], equal_nan=True)
        assert np.array_equal(want[i][0][0], truths[i, 0].astype(np.uint16))   # the bright ridge is found
    assert np.array_equal(got_async[1][0], want[1][0])


def test_segment_pool_passes_large_batches_through_shared_memory_files():
    """A device batch of boundary maps (>= 1 MB) reaches the workers as ONE file in /dev/shm they map read-only, not as
    pickled arrays; same results, and the file is gone once the batch has been collected."""
    import glob
    import os
    from oct_image_segmentation_models_amd.min_path_processing import graph_search, pool as gspool
    if not (os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK)):
        pytest.skip("/dev/shm not writable")
    n, H, W = 6, 256, 512
    maps = np.zeros((n, 2, H, W), np.uint8)
    for i in range(n):
        for c in range(2):
            rows = (H * (c + 1) // 3 + 9 * np.sin(np.arange(W) / 31.0 + i)).astype(int)
            maps[i, c, rows, np.arange(W)] = 255
    assert maps.nbytes >= 1 << 20
    grid = graph_search.create_graph_structure((W, H), 1)
    want = [graph_search.segment_maps(np.transpose(maps[i], (0, 2, 1)), None, grid)[:2] for i in range(n)]
    mine = os.path.join("/dev/shm", f"oct_gs_{os.getpid()}_*.u8")
    with gspool.SegmentPool((H, W), 1, workers=2) as pool:
        if pool.workers < 2:
            pytest.skip("worker pool unavailable here")
        job = pool.segment_async(maps)
        assert len(glob.glob(mine)) == 1                      # the batch sits in one file while it is in flight
        got = job.get()
        assert glob.glob(mine) == []
        again = pool.segment(maps[::-1].copy())               # a second batch: workers move on to the new file
    for i in range(n):
        assert np.array_equal(got[i][0], want[i][0]) and np.array_equal(got[i][1], want[i][1], equal_nan=True)
        assert np.array_equal(again[i][0], want[n - 1 - i][0])
    assert glob.glob(mine) == []


def test_pool_started_from_an_unguarded_script_falls_back_inline_instead_of_hanging(tmp_path):
    """Spawned workers re-import ``__main__``; a top-level script without the ``__main__`` guard (here: a script fed on
    stdin, which a worker cannot even open) can never bring a worker up.  The pool must notice (start-up probe with a
    timeout), say so, and run the post-process inline -- same results."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {root!r})\n"
        "import oct_image_segmentation_models_amd\n"
        "from oct_image_segmentation_models_amd.min_path_processing.pool import SegmentPool\n"
        "maps = np.zeros((3, 2, 24, 40), np.uint8); maps[:, 0, 7, :] = 255; maps[:, 1, 15, :] = 255\n"
        "p = SegmentPool((24, 40), workers=2)\n"
        "r = p.segment(maps)\n"
        "print('WORKERS', p.workers, 'ROWS', r[0][0][:, 0].tolist(), flush=True)\n"
        "p.close()\n")
    env = dict(os.environ, OCT_GS_POOL_START_TIMEOUT="5")
    out = subprocess.run([sys.executable, "-"], input=code.encode(), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                         env=env, timeout=120)
    assert out.returncode == 0
    assert b"WORKERS 1 ROWS [7, 15]" in out.stdout, out.stdout[-300:]
