"""BASELINE batch sizes on the GPU: configs[1] is timed at batch 32 and configs[4] at batch 128, so
the kernel variants those grids select are exercised here at exactly those sizes, through size-independent properties
(the oracle cannot run 32 full-size scans in seconds):

* training step, B=32, 256x512: finite loss and gradients; the device Dice losses equal the closed forms evaluated on
  the returned probabilities; gradient buffer deterministic across two identical steps; kernel selection logged;
* inference, B=128: a captured hipGraph replay == a plain forward == four B=32 forwards, bit for bit (inference is
  independent of batch composition), arg-max consistent with the probabilities."""
import numpy as np
import pytest
import torch

from oracle import unet_numpy as on

pytestmark = pytest.mark.gpu

H, W, C = 256, 512, 3


def scans(n, seed):
    from oct_image_segmentation_models_amd.common.synthetic import make_scans
    img, lab = make_scans(8, H, W, C, seed=seed)
    reps = (n + 7) // 8
    return np.tile(img, (reps, 1, 1, 1))[:n], np.tile(lab, (reps, 1, 1, 1))[:n]


def test_train_step_at_batch_32_full_size():
    from oct_image_segmentation_models_amd.engine import UNetEngine
    B = 32
    eng = UNetEngine(device="cuda:0", input_channels=1, num_classes=C, image_height=H, image_width=W, max_batch=B,
                     training=True, seed=5, init_seed=1)
    img, lab = scans(B, 11)
    img[8:] = np.roll(img[8:], 17, axis=2)                      # not 4 identical groups of 8
    x = torch.from_numpy(img).cuda(); l = torch.from_numpy(lab[..., 0].copy()).cuda()
    eng.set_dropout_step(2)
    eng.profile_begin()
    probs, _ = eng.forward(x, training=True, labels=l)
    loss4 = eng.loss_dice().cpu().numpy()
    eng.backward(l, macro=True)
    ents = eng.profile_end()
    g1 = eng.grads.clone()
    fams = sorted({e["kernel"].split("<")[0] for e in ents})
    print("kernel families at B=32 256x512:", fams)
    assert len(ents) > 40 and all(e["total_ms"] > 0 for e in ents)
    # routing at the configuration the bench times: the thin layers on conv_bt_k -- the 8-channel
    # launches in the two-pixel form --, the wide ones on conv_bx_k / conv_dwbx_k; none of the round-1 forward kernels
    names = {e["kernel"] for e in ents}
    assert {"conv_bt_k", "conv_bx_k", "conv_dwbx_k"} <= set(fams)
    assert sum(1 for n in names if n.startswith("conv_bt_k") and ",2px" in n) >= 5, sorted(names)
    # the BN-backward transform is applied on load: at most the one 16 -> 32 block keeps the stand-alone pass
    assert sum(e["launches"] for e in ents if e["kernel"].startswith("bn_bwd_apply")) <= 1
    assert sum(1 for n in names if n.endswith(",gb>")) >= 10, sorted(names)
    assert not ({"conv_igemm_k", "conv_igemm_p_k", "conv_pair8_k", "conv_thin8_k"} & set(fams)), fams
    p = probs.cpu().numpy().astype(np.float64)
    assert np.isfinite(p).all() and np.abs(p.sum(-1) - 1).max() < 1e-5
    y = on.one_hot(lab, C, np.float64)
    assert abs(loss4[0] - on.dice_loss_macro(y, p)) < 1e-5 and abs(loss4[1] - on.dice_loss_micro(y, p)) < 1e-5
    assert abs(loss4[2] - on.dice_coef_macro(y, p)) < 1e-4 and abs(loss4[3] - on.dice_coef_micro(y, p)) < 1e-4
    g = g1.cpu().numpy()
    assert np.isfinite(g).all() and np.abs(g).max() > 0
    for L in eng.layers:                                        # every tensor of every layer received a gradient
        n = L["kh"] * L["kw"] * L["cin"] * L["cout"]
        assert np.abs(g[L["kernel_off"]:L["kernel_off"] + n]).max() > 0, L["name"]
    # the same step again: identical bits (no atomics, fixed summation order)
    eng.set_dropout_step(2)
    eng.forward(x, training=True, labels=l, want_probs=False); eng.loss_dice(); eng.backward(l, macro=True)
    assert torch.equal(eng.grads, g1)
    # one Adam step lowers nothing to NaN and changes every parameter tensor
    p0 = eng.params.clone(); eng.adam_step(lr=1e-3)
    assert torch.isfinite(eng.params).all() and (eng.params != p0).float().mean() > 0.9


def test_inference_at_batch_128_graph_replay_equals_chunked_forwards():
    from oct_image_segmentation_models_amd.engine import UNetEngine
    B = 128
    eng = UNetEngine(device="cuda:0", input_channels=1, num_classes=C, image_height=H, image_width=W, max_batch=B,
                     training=False, seed=5, init_seed=1)
    # non-trivial moving statistics so BN inference coefficients matter
    rng = np.random.default_rng(0)
    wl = eng.get_weights()
    i = 0
    for L in eng.layers:
        i += 2
        if L["has_bn"]:
            c = L["cout"]
            wl[i] = rng.uniform(0.5, 1.5, c).astype(np.float32); wl[i + 1] = rng.normal(0, 0.1, c).astype(np.float32)
            wl[i + 2] = rng.normal(0, 0.1, c).astype(np.float32); wl[i + 3] = rng.uniform(0.5, 1.5, c).astype(np.float32)
            i += 4
    eng.set_weights(wl)
    img, _ = scans(B, 21)
    for k in range(B):
        img[k] = np.roll(img[k], 3 * k, axis=1)                 # 128 distinct scans
    x = torch.from_numpy(img).cuda()
    full, am = eng.forward(x, training=False, want_argmax=True)
    full = full.clone(); am = am.clone()
    for lo in range(0, B, 32):
        part, pam = eng.forward(x[lo:lo + 32].contiguous(), training=False, want_argmax=True)
        assert torch.equal(part, full[lo:lo + 32]) and torch.equal(pam, am[lo:lo + 32])
    xb = torch.zeros_like(x)
    gp, gam = eng.graph_capture(xb, want_probs=True, want_argmax=True)
    xb.copy_(x); eng.graph_launch(); torch.cuda.synchronize()
    assert torch.equal(gp, full) and torch.equal(gam, am)
    srt = torch.sort(full, dim=-1).values
    tie = (srt[..., -1] - srt[..., -2]) < 1e-6
    assert torch.equal(gam.long()[~tie], full.argmax(-1)[~tie])
    assert torch.isfinite(full).all() and (full.sum(-1) - 1).abs().max() < 1e-5
