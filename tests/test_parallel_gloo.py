"""World-size-2 ``gloo`` tests of the data-parallel path on CPU.  The arithmetic backend in
these tests is the oracle (test infrastructure); what is under test is the DP plumbing in ``parallel.py``:
batch sharding, loss scaling 1/R, ONE sum-all-reduce of the flat gradient buffer, moving-stat averaging."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import unet_numpy as on


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tmpdir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from oct_image_segmentation_models_amd import parallel
    r, lr, w = parallel.init("gloo")
    assert (r, w) == (rank, world) and parallel.world_size() == world

    cfg = on.UNetConfig(num_classes=3, start_neurons=4, pool_layers=1)
    params, state = on.init_params(cfg, seed=0, randomize_bn=True)
    G = 4                                          # global batch, 2 per rank
    images, labels = on.synth_scans(G, 16, 32, 3, seed=3)
    x = on.preprocess_u8(images, np.float64)
    mask = np.ones((G, 8, 16, 8))
    lo, hi = parallel.shard_batch(G, rank, world)
    assert (lo, hi) == (rank * 2, rank * 2 + 2)
    # rank-local step: per-rank BN statistics, loss scaled by 1/world
    _, cache = on.forward(cfg, params, state, x[lo:hi], training=True, dropout_mask=mask[lo:hi])
    loss, grads = on.backward(cfg, params, cache, labels[lo:hi], macro=True, loss_scale=1.0 / world)
    flat = torch.from_numpy(on.flatten_grads(grads).copy())
    parallel.allreduce_gradients(flat)             # the ONE exchange step of the path
    new_state = torch.from_numpy(on.flatten_state(on.updated_moving_stats(cfg, state, cache)))
    avg_state = parallel.average_moving_stats(new_state)
    tmax = parallel.max_over_ranks(float(rank + 1))
    parallel.barrier()
    np.savez(os.path.join(tmpdir, f"r{rank}.npz"), g=flat.numpy(), loss=loss, state=new_state.numpy(),
             avg=avg_state.numpy(), tmax=tmax)
    dist.destroy_process_group()


def test_dp_two_ranks_equals_mean_of_rank_losses(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "r0.npz"); r1 = np.load(tmp_path / "r1.npz")
    assert np.array_equal(r0["g"], r1["g"])        # every rank holds the same reduced gradient
    assert r0["tmax"] == 2.0 and r1["tmax"] == 2.0
    assert np.allclose(r0["avg"], (r0["state"] + r1["state"]) / 2) and np.array_equal(r0["avg"], r1["avg"])
    # single-process reference: gradient of mean over ranks of the per-rank (per-rank-BN) macro-Dice losses
    cfg = on.UNetConfig(num_classes=3, start_neurons=4, pool_layers=1)
    params, state = on.init_params(cfg, seed=0, randomize_bn=True)
    images, labels = on.synth_scans(4, 16, 32, 3, seed=3)
    x = on.preprocess_u8(images, np.float64)
    tot = None; losses = []
    for lo in (0, 2):
        _, cache = on.forward(cfg, params, state, x[lo:lo + 2], training=True, dropout_mask=np.ones((2, 8, 16, 8)))
        loss, grads = on.backward(cfg, params, cache, labels[lo:lo + 2], macro=True, loss_scale=0.5)
        g = on.flatten_grads(grads); tot = g if tot is None else tot + g; losses.append(loss)
    assert np.allclose(r0["g"], tot, rtol=0, atol=1e-15)
    assert abs(r0["loss"] - losses[0]) < 1e-15 and abs(r1["loss"] - losses[1]) < 1e-15


def _seed_worker(rank, world, port, tmpdir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from oct_image_segmentation_models_amd import parallel
    from oct_image_segmentation_models_amd.common.data_generator import DataGenerator
    parallel.init("gloo")
    seed = parallel.shared_seed(None)              # unseeded run: rank 0's OS-entropy draw, broadcast
    stamp = parallel.broadcast_object(f"stamp-from-rank-{rank}")
    rng = np.random.default_rng(0)
    images = rng.integers(0, 256, (12, 4, 6, 1)).astype(np.uint8)
    labels = rng.integers(0, 3, (12, 4, 6, 1)).astype(np.uint8)
    g = DataGenerator(images, labels, 4, [], "none", (), False, None, seed=seed)
    taken = []
    for epoch in range(2):
        for _ in range(len(g)):
            X, lab = g.next_batch_u8(parallel.shard_batch(g.batch_size, rank, world))   # this rank's slice of the GLOBAL
            assert X.shape[0] == g.batch_size // world and lab.shape[0] == X.shape[0]   # batch, as Model._host_batch takes it
            taken.append(X)
        g.on_epoch_end()
    np.savez(os.path.join(tmpdir, f"s{rank}.npz"), seed=seed, taken=np.stack(taken), stamp=stamp)
    dist.destroy_process_group()


def test_dp_ranks_share_one_shuffle_and_partition_each_global_batch(tmp_path):
    """with seed=None every rank used to shuffle differently, so rank slices duplicated / dropped samples."""
    port = _free_port()
    mp.spawn(_seed_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    s0 = np.load(tmp_path / "s0.npz"); s1 = np.load(tmp_path / "s1.npz")
    assert int(s0["seed"]) == int(s1["seed"])
    assert str(s0["stamp"]) == str(s1["stamp"]) == "stamp-from-rank-0"
    rng = np.random.default_rng(0)
    images = rng.integers(0, 256, (12, 4, 6, 1)).astype(np.uint8)
    key = lambda a: a.reshape(a.shape[0], -1).sum(1).tolist()      # noqa: E731  (image sums identify samples here)
    assert len(set(key(images))) == 12
    per_epoch = s0["taken"].shape[0] // 2
    for e in range(2):
        got = []
        for b in range(per_epoch):
            got += key(s0["taken"][e * per_epoch + b]) + key(s1["taken"][e * per_epoch + b])
        assert sorted(got) == sorted(key(images))   # 3 global batches of 4 = every sample exactly once per epoch


def test_shard_helpers():
    from oct_image_segmentation_models_amd import parallel
    assert [parallel.shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert [parallel.shard_range(2, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]   # ragged / empty
    assert parallel.shard_batch(256, 3, 8) == (96, 128)
    with pytest.raises(ValueError):
        parallel.shard_batch(10, 0, 4)
    assert parallel.env_rank() == (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)),
                                   int(os.environ.get("WORLD_SIZE", 1)))
