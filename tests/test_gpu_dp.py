"""GPU test of the data-parallel path with the HIP engine as the arithmetic backend (SURVEY 8 row a14 / 8e;
reference: tf.distribute.MirroredStrategy, training/training.py:185-188,243).

A one-GPU box cannot host two RCCL ranks, so the two ranks of this test share cuda:0 and exchange through ``gloo``
(the collective is ``torch.distributed`` either way; ``parallel.py`` is backend-agnostic).  What is checked:

* the reduced gradient of a 2-rank step (per-rank batch halves, per-rank BN statistics, per-rank dropout stream, loss
  scaled by 1/2, tail segment all-reduced on the side stream behind the engine's event, encoder segment after backward)
  equals the sum of the two half-batch gradients computed by ONE process with the same engines;
* ``Model.fit`` under 2 ranks leaves bit-identical parameters on both ranks after several optimizer steps, and the
  ranks consumed disjoint halves of the same global batches.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import unet_numpy as on

pytestmark = pytest.mark.gpu

H, W, C, G = 32, 64, 3, 4          # global batch 4 = 2 per rank
CFG = dict(input_channels=1, num_classes=C, image_height=H, image_width=W, start_neurons=8, pool_layers=2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _half_gradient(rank, images, labels):
    """One rank's share of the step, exactly as bench.py / Model.fit run it (minus the collective)."""
    from oct_image_segmentation_models_amd.engine import UNetEngine
    eng = UNetEngine(device="cuda:0", max_batch=G // 2, training=True, seed=1000 + rank, init_seed=0, **CFG)
    lo, hi = rank * (G // 2), (rank + 1) * (G // 2)
    x = torch.from_numpy(images[lo:hi]).cuda(); lab = torch.from_numpy(labels[lo:hi, ..., 0].copy()).cuda()
    eng.set_dropout_step(5)
    eng.forward(x, training=True, labels=lab, want_probs=False)
    loss4 = eng.loss_dice()
    return eng, x, lab, loss4


def _worker(rank, world, port, tmpdir):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from oct_image_segmentation_models_amd import optimizers, parallel
    from oct_image_segmentation_models_amd.common import custom_losses, custom_metrics
    from oct_image_segmentation_models_amd.common.data_generator import DataGenerator
    from oct_image_segmentation_models_amd.models import get_model_class
    parallel.init("gloo")
    torch.cuda.set_device(0)
    images, labels = on.synth_scans(G, H, W, C, seed=3)

    # ---- (a) one step through GradReducer with the side-stream overlap ----
    eng, x, lab, loss4 = _half_gradient(rank, images, labels)
    red = parallel.GradReducer(eng, overlap=True)
    assert red.overlap and 0 < red.off < eng.n_params
    red.backward_and_reduce(lab, macro=True, loss_scale=1.0 / world)
    torch.cuda.synchronize()
    g_overlap = eng.grads.cpu().numpy().copy()
    # the same step with ONE all-reduce after backward gives the same bits
    red.close()
    eng.set_dropout_step(5)
    eng.forward(x, training=True, labels=lab, want_probs=False); eng.loss_dice()
    parallel.GradReducer(eng, overlap=False).backward_and_reduce(lab, macro=True, loss_scale=1.0 / world)
    torch.cuda.synchronize()
    g_flat = eng.grads.cpu().numpy().copy()

    # ---- (b) Model.fit, 3 optimizer steps per epoch x 2 epochs on a 12-scan set, global batch 4 ----
    tr_i, tr_l = on.synth_scans(12, H, W, C, seed=11)
    model = get_model_class("unet")(**{k: v for k, v in CFG.items()}).build_model()
    model.config["seed"] = 3
    loss = custom_losses.custom_loss_objects["dice_loss_macro"]["function"](num_classes=C, is_y_true_sparse=True)
    metric = custom_metrics.training_monitor_metric_objects["dice_coef_macro"](True, C)
    model.compile(optimizer=optimizers.Adam(learning_rate=2e-3), loss=loss, metrics=[metric])
    gen = DataGenerator(tr_i, tr_l, G, [], "none", (), False, None, seed=parallel.shared_seed(None))
    hist = model.fit(x=gen, epochs=2, verbose=0)
    torch.cuda.synchronize()
    np.savez(os.path.join(tmpdir, f"r{rank}.npz"), g_overlap=g_overlap, g_flat=g_flat, loss=loss4.cpu().numpy(),
             params=model.engine.params.cpu().numpy(), state=model.engine.state.cpu().numpy(),
             hist=np.array(hist.history["loss"]), off=red.off)
    torch.distributed.destroy_process_group()


def test_two_rank_step_equals_sum_of_half_batch_gradients_and_fit_keeps_ranks_in_sync(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "r0.npz"); r1 = np.load(tmp_path / "r1.npz")
    # every rank holds the same reduced gradient; overlap and flat forms agree bit for bit
    assert np.array_equal(r0["g_overlap"], r1["g_overlap"]) and np.array_equal(r0["g_flat"], r1["g_flat"])
    assert np.array_equal(r0["g_overlap"], r0["g_flat"])
    # single-process reference with the same engines: sum of the two half-batch gradients at loss_scale 1/2
    images, labels = on.synth_scans(G, H, W, C, seed=3)
    tot, losses = None, []
    for rank in range(2):
        eng, x, lab, loss4 = _half_gradient(rank, images, labels)
        eng.backward(lab, macro=True, loss_scale=0.5)
        torch.cuda.synchronize()
        g = eng.grads.cpu().numpy().astype(np.float32)
        tot = g if tot is None else tot + g
        losses.append(loss4.cpu().numpy())
    assert np.array_equal(r0["g_overlap"], tot)                       # a + b in f32, either order
    assert np.array_equal(r0["loss"], losses[0]) and np.array_equal(r1["loss"], losses[1])
    assert np.abs(tot).max() > 0 and np.isfinite(tot).all()
    # Model.fit: identical parameters on both ranks after 6 steps, loss history identical (mean over ranks), finite
    assert np.array_equal(r0["params"], r1["params"])
    assert np.array_equal(r0["hist"], r1["hist"]) and np.isfinite(r0["hist"]).all() and len(r0["hist"]) == 2
    assert not np.array_equal(r0["state"], r1["state"])                # BN moving statistics stay per replica


def _rccl_one_rank(port, q):
    """Child process: a ONE-rank RCCL communicator (backend "nccl" = RCCL on ROCm) driving the exchange step."""
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        import torch.distributed as dist
        from oct_image_segmentation_models_amd import parallel
        from oct_image_segmentation_models_amd.engine import UNetEngine
        parallel.init("nccl", force=True)
        assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
        assert parallel.collective_active()
        info = parallel.describe()
        images, labels = on.synth_scans(G, H, W, C, seed=3)
        x = torch.from_numpy(images).cuda(); lab = torch.from_numpy(labels[..., 0].copy()).cuda()
        out = {}
        for mode in ("plain", "flat", "overlap"):
            eng = UNetEngine(device="cuda:0", max_batch=G, training=True, seed=1000, init_seed=0, **CFG)
            eng.set_dropout_step(5)
            eng.forward(x, training=True, labels=lab, want_probs=False); eng.loss_dice()
            if mode == "plain":
                eng.backward(lab, macro=True, loss_scale=1.0)                 # no collective at all
            else:
                red = parallel.GradReducer(eng, overlap=(mode == "overlap"))
                assert red.active and red.overlap == (mode == "overlap")
                red.backward_and_reduce(lab, macro=True, loss_scale=1.0)      # RCCL all-reduce(s) of the flat buffer
                red.close()
            torch.cuda.synchronize()
            out[mode] = eng.grads.cpu().numpy().copy()
        # a timed loop: the two collectives per step really run, step after step, on their streams
        eng = UNetEngine(device="cuda:0", max_batch=G, training=True, seed=1000, init_seed=0, **CFG)
        red = parallel.GradReducer(eng, overlap=True)
        for _ in range(5):
            eng.forward(x, training=True, labels=lab, want_probs=False); eng.loss_dice()
            red.backward_and_reduce(lab, macro=True, loss_scale=1.0); eng.adam_step(lr=1e-3)
        torch.cuda.synchronize()
        ok = bool(torch.isfinite(eng.params).all())
        parallel.shutdown()
        q.put(("ok", info, out, ok))
    except Exception as e:       # noqa: BLE001  -- reported to the parent
        import traceback
        q.put(("error", traceback.format_exc(), None, False))


def test_rccl_exchange_step_runs_on_one_gpu():
    """SURVEY 8 row a14 / 8e.  RCCL itself, not gloo: ``init_process_group("nccl", world_size=1)`` on the one GPU of this
    box, then the exchange step exactly as the 8-GPU run issues it -- tail event -> all-reduce of grads[off:] on the side
    stream -> all-reduce of grads[:off] -> join.  A one-rank SUM all-reduce is the identity: gradients must equal the
    no-collective backward BIT FOR BIT, with and without the overlap."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_one_rank, args=(_free_port(), q))
    p.start()
    try:
        status, info, out, ok = q.get(timeout=300)
    finally:
        p.join(timeout=60)
        if p.is_alive():
            p.kill()
    assert status == "ok", info
    assert info["backend"] == "nccl" and info["world_size"] == 1 and info["ranks"][0]["device"] == "cuda:0", info
    assert np.abs(out["plain"]).max() > 0 and np.isfinite(out["plain"]).all()
    assert np.array_equal(out["plain"], out["flat"]) and np.array_equal(out["plain"], out["overlap"])
    assert ok
