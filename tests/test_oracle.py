"""CPU tests of the oracle itself: numpy restatement vs independent torch/autograd
restatement, analytic known answers, finite differences.  (PARITY UNPINNED vs
TensorFlow: see oracle/unet_numpy.py header.)"""
import numpy as np
import pytest
import torch

from oracle import unet_numpy as on
from oracle import unet_torch as ot


def small_cfg(**kw):
    d = dict(input_channels=1, num_classes=3, start_neurons=4, pool_layers=2, conv_layers=2)
    d.update(kw)
    return on.UNetConfig(**d)


def test_param_counts_match_survey():
    # 487403 trainable + 1712 moving at C=3; 489124 total at C=4
    assert on.param_count(on.UNetConfig(num_classes=3)) == (487403, 1712)
    t, s = on.param_count(on.UNetConfig(num_classes=4))
    assert t + s == 489124
    t, s = on.param_count(on.UNetConfig(num_classes=3, pool_layers=5))
    assert (t, t + s) == (1948267, 1951771)
    assert len(build := on.build_plan(on.UNetConfig())) == 23 and sum(c.has_bn for c in build) == 22


def test_same_padding_side_2x2():
    # delta image locates the pad side: 2x2 'same' pads bottom/right only
    x = np.zeros((1, 4, 4, 1)); x[0, 3, 3, 0] = 1.0
    k = np.arange(1, 5, dtype=np.float64).reshape(2, 2, 1, 1)
    out = on.conv2d_same(x, k, np.zeros(1))
    # out[y,x] = sum k[ky,kx] * in[y+ky, x+kx]; delta at (3,3) seen by (3,3)->k00, (2,3)->k10, (3,2)->k01, (2,2)->k11
    assert out[0, 3, 3, 0] == 1 and out[0, 2, 3, 0] == 3 and out[0, 3, 2, 0] == 2 and out[0, 2, 2, 0] == 4
    assert out.sum() == 10


def test_bn_constant_tensor_gives_beta():
    z = np.full((2, 4, 4, 3), 7.0)
    y, mean, var, _, _ = on.batchnorm_train(z, np.array([1., 2., 3.]), np.array([.1, .2, .3]), 1e-3)
    assert np.allclose(y, np.array([.1, .2, .3])) and np.allclose(var, 0) and np.allclose(mean, 7)


def test_dice_known_answers():
    y = on.one_hot(np.array([[[0, 1], [2, 1]]]), 3, np.float64)
    # perfect prediction: score=(2T+s)/(2T+s)=1 per class => loss 0
    assert abs(on.dice_loss_macro(y, y)) < 1e-12 and abs(on.dice_loss_micro(y, y)) < 1e-12
    assert abs(on.dice_coef_macro(y, y) - 1) < 1e-12 and abs(on.dice_coef_micro(y, y) - 1) < 1e-12
    # empty class predicted nowhere and present nowhere => score = s/s = 1
    y2 = on.one_hot(np.zeros((1, 2, 2), int), 3, np.float64)
    assert abs(on.dice_loss_macro(y2, y2)) < 1e-12
    # uniform prediction 1/3: per class I=T/3, P=4/3
    p = np.full((1, 2, 2, 3), 1 / 3)
    T = y.sum(axis=(1, 2))[0]
    expect = 1 - np.mean((2 * T / 3 + 1e-5) / (T + 4 / 3 + 1e-5))
    assert abs(on.dice_loss_macro(y, p) - expect) < 1e-12


@pytest.mark.parametrize("training", [False, True])
@pytest.mark.parametrize("C", [3, 4])
def test_forward_numpy_vs_torch(training, C):
    cfg = small_cfg(num_classes=C)
    params, state = on.init_params(cfg, seed=1, randomize_bn=True)
    rng = np.random.default_rng(0)
    x = rng.uniform(0, 1, (2, 16, 32, 1))
    mask = (rng.uniform(size=(2, 4, 8, 16)) > 0.5).astype(np.float64)
    probs, _ = on.forward(cfg, params, state, x, training=training, dropout_mask=mask)
    tp, ts = ot.to_torch(params, state)
    tprobs = ot.forward(cfg, tp, ts, torch.tensor(x), training=training, dropout_mask=torch.tensor(mask))
    assert probs.shape == (2, 16, 32, C)
    assert np.abs(probs - tprobs.numpy()).max() < 1e-12
    assert np.allclose(probs.sum(-1), 1)


@pytest.mark.parametrize("macro", [True, False])
def test_backward_numpy_vs_autograd(macro):
    cfg = small_cfg()
    params, state = on.init_params(cfg, seed=2, randomize_bn=True)
    rng = np.random.default_rng(3)
    x = rng.uniform(0, 1, (2, 16, 32, 1))
    labels = rng.integers(0, 3, (2, 16, 32, 1)).astype(np.uint8)
    mask = (rng.uniform(size=(2, 4, 8, 16)) > 0.5).astype(np.float64)
    probs, cache = on.forward(cfg, params, state, x, training=True, dropout_mask=mask)
    loss, grads = on.backward(cfg, params, cache, labels, macro=macro, loss_scale=0.5)
    tloss, tprobs, tgrads = ot.loss_and_grads(cfg, params, state, x, labels, macro=macro,
                                              dropout_mask_np=mask, loss_scale=0.5)
    assert abs(loss - tloss) < 1e-12
    for g, tg in zip(grads, tgrads):
        for k in g:
            # conv bias ahead of a BN has an analytically-zero gradient (rounding noise only)
            scale = max(1e-6, np.abs(tg[k]).max())
            assert np.abs(g[k] - tg[k]).max() / scale < 1e-8, k


def test_backward_finite_difference():
    cfg = small_cfg(start_neurons=2, pool_layers=1)
    params, state = on.init_params(cfg, seed=5, randomize_bn=True)
    rng = np.random.default_rng(7)
    x = rng.uniform(0, 1, (1, 8, 8, 1))
    labels = rng.integers(0, 3, (1, 8, 8, 1)).astype(np.uint8)
    mask = np.ones((1, 4, 4, 4))
    _, cache = on.forward(cfg, params, state, x, training=True, dropout_mask=mask)
    _, grads = on.backward(cfg, params, cache, labels)
    y = on.one_hot(labels, 3, np.float64)
    flat = on.flatten_params(params)
    gflat = on.flatten_grads(grads)
    for idx in rng.choice(flat.size, 12, replace=False):
        e = 1e-6
        vals = []
        for s in (+e, -e):
            f = flat.copy(); f[idx] += s
            pr, _ = on.forward(cfg, on.unflatten_params(cfg, f), state, x, training=True, dropout_mask=mask)
            vals.append(on.dice_loss_macro(y, pr))
        fd = (vals[0] - vals[1]) / (2 * e)
        assert abs(fd - gflat[idx]) < 1e-6 * max(1, abs(fd)), (idx, fd, gflat[idx])


def test_flatten_roundtrip_and_keras_order():
    cfg = small_cfg()
    params, state = on.init_params(cfg, seed=0, randomize_bn=True)
    flat = on.flatten_params(params)
    back = on.unflatten_params(cfg, flat)
    assert all(np.array_equal(a[k], b[k]) for a, b in zip(params, back) for k in a)
    sflat = on.flatten_state(state)
    sback = on.unflatten_state(cfg, sflat)
    assert all(np.array_equal(a[k], b[k]) for a, b in zip(state, sback) for k in a)
    wl = on.keras_weight_list(params, state)
    nconv = len(params); nbn = len(state)
    assert len(wl) == 2 * nconv + 4 * nbn
    assert wl[0].shape == (3, 3, 1, 4) and wl[2].shape == (4,) and wl[-2].shape == (1, 1, 4, 3)


def test_moving_stats_update():
    cfg = small_cfg()
    params, state = on.init_params(cfg, seed=0)
    x = np.random.default_rng(0).uniform(0, 1, (2, 8, 8, 1))
    _, cache = on.forward(cfg, params, state, x, training=True, dropout_mask=np.ones((2, 2, 2, 16)))
    new = on.updated_moving_stats(cfg, state, cache)
    n = 2 * 8 * 8
    assert np.allclose(new[0]["moving_mean"], 0.01 * cache[0]["mean"])
    assert np.allclose(new[0]["moving_var"], 0.99 + 0.01 * cache[0]["var"] * n / (n - 1))


def test_adam_matches_keras_formula_first_step():
    th, m, v = np.array([1.0]), np.zeros(1), np.zeros(1)
    th2, m2, v2 = on.adam_step(th, np.array([0.5]), m, v, 1)
    # t=1: m=0.05, v=0.00025*... lr_t = lr*sqrt(1-b2)/(1-b1); update = lr_t*m/(sqrt(v)+eps)
    lr_t = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
    assert np.allclose(th2, 1.0 - lr_t * 0.05 / (np.sqrt(0.001 * 0.25) + 1e-7))


def test_postprocess_boundary_maps():
    lab = np.zeros((1, 8, 4), int); lab[0, 3:, :] = 1; lab[0, 6:, :] = 2
    probs = np.eye(3)[lab]
    am, cat = on.perform_argmax(probs)
    assert np.array_equal(am, lab) and cat.shape == (1, 3, 8, 4)
    maps = on.convert_predictions_to_maps_semantic(cat)
    assert maps.shape == (1, 2, 8, 4) and maps.dtype == np.uint8
    # central difference x2 minus the rolled copy leaves one 255 response on the
    # first row of the lower region (rows 3 and 6 here), every column
    assert np.array_equal(np.nonzero(maps[0, 0][:, 0])[0], [3]) and maps[0, 0][3, 0] == 255
    assert np.array_equal(np.nonzero(maps[0, 1][:, 0])[0], [6]) and maps[0, 1][6, 0] == 255


def test_synth_scans_all_classes():
    im, lab = on.synth_scans(3, 64, 128, 4, seed=1)
    assert im.shape == (3, 64, 128, 1) and im.dtype == np.uint8 and lab.dtype == np.uint8
    assert len(np.unique(lab)) == 4


def test_oracle_reproduces_committed_golden():
    """The committed fixture (tests/golden/unet_golden.npz, made by make_unet_golden.py) pins the oracle."""
    import os
    from tests.helpers import dropout_keep_mask
    from tests.golden.make_unet_golden import CFG, B, H, W
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "unet_golden.npz"))
    cfg = on.UNetConfig(**CFG)
    params = on.unflatten_params(cfg, G["params_flat"].astype(np.float64))
    state = on.unflatten_state(cfg, G["state_flat"].astype(np.float64))
    x = on.preprocess_u8(G["images"], np.float64)
    mask = dropout_keep_mask(int(G["dropout_seed"]), int(G["dropout_step"]), (B, H >> 2, W >> 2, 16)).astype(np.float64)
    pi, _ = on.forward(cfg, params, state, x, training=False)
    pt, cache = on.forward(cfg, params, state, x, training=True, dropout_mask=mask)
    assert np.abs(pi - G["probs_infer"]).max() < 1e-13 and np.abs(pt - G["probs_train"]).max() < 1e-13
    _, grads = on.backward(cfg, params, cache, G["labels"], macro=True)
    assert np.abs(on.flatten_grads(grads) - G["grads_macro"]).max() < 1e-13
    assert np.abs(on.flatten_state(on.updated_moving_stats(cfg, state, cache)) - G["state_after"]).max() < 1e-13


def test_focal_dice_loss_known_answer_and_gradient():
    """focal_dice_loss (custom_losses.py:98-178): hand-computed value on a 2-pixel case, gradient vs finite
    differences and vs torch autograd of the same formula."""
    import torch
    p = np.array([[[[0.7, 0.2, 0.1], [0.25, 0.5, 0.25]]]], np.float64)       # (1,1,2,3)
    lab = np.array([[[[0], [2]]]], np.uint8)
    cw = np.array([1.0, 2.0, 3.0])
    f = (1.0 * 0.3 ** 2 * -np.log(0.7) + 3.0 * 0.75 ** 2 * -np.log(0.25)) / 2
    assert abs(on.focal_loss_mean(lab, p, 2.0, cw) - f) < 1e-12
    y = on.one_hot(lab, 3, np.float64)
    assert abs(on.focal_dice_loss(lab, p, 3, 2.0, cw, 0.3, True) - (0.3 * f + 0.7 * on.dice_loss_macro(y, p))) < 1e-12
    rng = np.random.default_rng(0)
    z = rng.normal(size=(2, 4, 5, 3)); p = np.exp(z) / np.exp(z).sum(-1, keepdims=True)
    lab = rng.integers(0, 3, (2, 4, 5, 1)).astype(np.uint8)
    for gamma, w in ((2.0, cw), (1.5, None)):
        g = on.focal_loss_grad(lab, p, gamma, w)
        for idx in [(0, 1, 2, int(lab[0, 1, 2, 0])), (1, 3, 4, int(lab[1, 3, 4, 0])), (1, 0, 0, (int(lab[1, 0, 0, 0]) + 1) % 3)]:
            q = p.copy(); q[idx] += 1e-6; r = p.copy(); r[idx] -= 1e-6
            fd = (on.focal_loss_mean(lab, q, gamma, w) - on.focal_loss_mean(lab, r, gamma, w)) / 2e-6
            assert abs(fd - g[idx]) < 1e-6 * max(1.0, abs(fd))
        tp = torch.tensor(p, requires_grad=True)
        tl = torch.tensor(lab[..., 0].astype(np.int64))
        py = tp.gather(-1, tl[..., None])[..., 0].clamp(1e-7, 1 - 1e-7)
        tw = torch.ones_like(py) if w is None else torch.tensor(w)[tl]
        (tw * (1 - py) ** gamma * -py.log()).sum().div(tl.numel()).backward()
        assert np.abs(tp.grad.numpy() - g).max() < 1e-12
