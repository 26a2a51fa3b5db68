"""BASELINE batch sizes on the GPU: configs[1] is timed at batch 32 and configs[4] at batch 128, so
the kernel variants those grids select are exercised here at exactly those sizes: against the fp64 torch oracle
(``test_bench_configuration_matches_the_fp64_oracle``: the numpy oracle cannot run 32 full-size scans in seconds, the
multi-threaded torch one can) and through size-independent properties:

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



def test_bench_configuration_matches_the_fp64_oracle():
    """BASELINE configs[1] exactly as ``bench.py`` times it -- 256x512x1, pool_layers 4, 3 classes, batch 32, fp32,
    default options (so: the grid-size dependent kernel selection, persistent block counts and statistic-row counts of
    B = 32) -- against ``oracle/unet_torch.py`` in fp64 (reference: training/training.py:401-407).  Forward: every
    layer's pre-BN output z and the loss.  Backward: every layer's dz from the head downwards and every gradient tensor.
    An fp32 path can only differ from an fp64 one at ReLU kinks (|BN pre-activation| below fp32 rounding flips a mask
    and the flip spreads over a few pixels per layer), so backward tensors are gated by quantiles: at least 99.9 % of the
    elements within 5e-4 of the tensor's scale and a kernel-gradient relative L2 error of at most 1e-3."""
    from oct_image_segmentation_models_amd.engine import UNetEngine
    from oracle import unet_torch as ot
    B, P = 32, 4
    torch.set_num_threads(min(16, torch.get_num_threads() or 16))
    cfg = on.UNetConfig(num_classes=C, start_neurons=8, pool_layers=P)
    params, state = on.init_params(cfg, seed=7, dtype=np.float32, randomize_bn=True)
    eng = UNetEngine(device="cuda:0", input_channels=1, num_classes=C, image_height=H, image_width=W, max_batch=B,
                     training=True, seed=5, init_seed=1)
    eng.set_weights(on.keras_weight_list(params, state))
    img, lab = scans(B, 31)
    for k in range(B):
        img[k] = np.roll(img[k], 5 * k, axis=1); lab[k] = np.roll(lab[k], 5 * k, axis=1)     # 32 distinct scans
    x = torch.from_numpy(img).cuda(); l = torch.from_numpy(lab[..., 0].copy()).cuda()
    eng.set_dropout_step(3)
    mask = eng.dropout_mask(B).cpu().numpy().astype(np.float64)
    eng.forward(x, training=True, labels=l, want_probs=False)
    loss4 = eng.loss_dice().cpu().numpy()
    eng.backward(l, macro=True, loss_scale=1.0)
    g = eng.grads.cpu().numpy().astype(np.float64)
    nb = len(eng.layers) - 1

    # ---- the oracle: one fp64 forward + autograd backward over the same 32 scans ----
    tp, ts = ot.to_torch(params, state, dtype=torch.float64, requires_grad=True)
    zs = []
    probs = ot.forward(cfg, tp, ts, torch.tensor(on.preprocess_u8(img, np.float64)), training=True,
                       dropout_mask=torch.tensor(mask), collect_z=zs)
    y = torch.nn.functional.one_hot(torch.tensor(lab[..., 0].astype("int64")), C).double()
    loss = ot.dice_loss(y, probs, macro=True)
    loss.backward()
    assert abs(float(loss4[0]) - float(loss)) < 1e-5, (float(loss4[0]), float(loss))

    # forward, per layer
    for li in range(nb):
        zr = zs[li].detach().permute(0, 2, 3, 1).numpy()
        z = eng.debug_activation(li, 0)[:B].cpu().numpy()
        err = np.abs(z - zr).max() / np.abs(zr).max()
        assert err < 1e-4, f"layer {li} {eng.layers[li]['name']}: z differs by {err:.2e} of its scale"
    # backward, per layer from the head downwards
    worst_q = 0.0
    for li in range(nb - 1, -1, -1):
        dzr = zs[li].grad.permute(0, 2, 3, 1).numpy()
        dz = eng.debug_dz(li)[:B].cpu().numpy()
        scale = np.abs(dzr).max()
        bad = float((np.abs(dz - dzr) > 5e-4 * scale).mean())
        worst_q = max(worst_q, bad)
        assert bad <= 1e-3, f"layer {li} {eng.layers[li]['name']}: {bad:.2e} of dz beyond 5e-4 of its scale"
        assert np.linalg.norm(dz - dzr) <= 2e-3 * np.linalg.norm(dzr), eng.layers[li]["name"]
    # every gradient tensor
    for L_, p_ in zip(eng.layers, tp):
        n = L_["kh"] * L_["kw"] * L_["cin"] * L_["cout"]; c = L_["cout"]
        gk, rk = g[L_["kernel_off"]:L_["kernel_off"] + n], p_["kernel"].grad.numpy().ravel()
        assert np.linalg.norm(gk - rk) <= 1e-3 * np.linalg.norm(rk), f"{L_['name']}.kernel"
        kscale = np.abs(rk).max()
        pieces = [("bias", L_["bias_off"])] + ([("gamma", L_["gamma_off"]), ("beta", L_["beta_off"])] if L_["has_bn"] else [])
        for key, off in pieces:
            rv = p_[key].grad.numpy().ravel()
            # a conv bias ahead of a BN has an analytically zero gradient: judge it on the kernel's scale
            scale = max(np.abs(rv).max(), kscale if key == "bias" else 0.0, 1e-12)
            assert np.abs(g[off:off + c] - rv).max() <= 1e-3 * scale, f"{L_['name']}.{key}"
    print(f"B=32 oracle parity: loss {float(loss4[0]):.6f} vs {float(loss):.6f}; worst dz outlier fraction {worst_q:.2e}")


def _bf16_round(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(torch.bfloat16).double().numpy()


def _bf16_ulp(a):
    return 2.0 ** (np.floor(np.log2(np.maximum(np.abs(a), 1e-30))) - 7)


def test_configs2_bf16_at_batch_64_full_size():
    """BASELINE configs[2] at its size: 512x1024x1, pool_layers 5, batch 64, bf16 activations + MFMA operands with fp32
    accumulation / BN statistics / parameters.  Properties (finite, closed-form Dice on the returned probabilities,
    every tensor gets a gradient, two identical steps give identical bits) plus the layer-local "one bf16 rounding" check
    on two sampled layers: the stored z of a layer recomputed in fp64 from the STORED input tensor under the
    bf16-operand model (activation and weights rounded once, exact products, wide accumulation)."""
    from oct_image_segmentation_models_amd.common.synthetic import make_scans
    from oct_image_segmentation_models_amd.engine import UNetEngine
    B, Hc, Wc, P = 64, 512, 1024, 5
    eng = UNetEngine(device="cuda:0", input_channels=1, num_classes=C, image_height=Hc, image_width=Wc, max_batch=B,
                     training=True, seed=5, init_seed=2, pool_layers=P, dtype="bfloat16")
    img8, lab8 = make_scans(8, Hc, Wc, C, seed=41)
    img = np.concatenate([np.roll(img8, 9 * k, axis=2) for k in range(B // 8)]); lab = np.concatenate([np.roll(lab8, 9 * k, axis=2) for k in range(B // 8)])
    x = torch.from_numpy(img).cuda(); l = torch.from_numpy(lab[..., 0].copy()).cuda()
    eng.set_dropout_step(2)
    probs, _ = eng.forward(x, training=True, labels=l)
    loss4 = eng.loss_dice().cpu().numpy()
    eng.backward(l, macro=True)
    g1 = eng.grads.clone()
    assert torch.isfinite(probs).all() and (probs.sum(-1) - 1).abs().max() < 1e-5
    y = torch.nn.functional.one_hot(l.long(), C).double(); pd = probs.double()
    I = (y * pd).sum(dim=(1, 2)); D = y.sum(dim=(1, 2)) + pd.sum(dim=(1, 2))
    assert abs(float(1 - ((2 * I + 1e-5) / (D + 1e-5)).mean()) - float(loss4[0])) < 1e-5
    g = g1.cpu().numpy()
    assert np.isfinite(g).all()
    for L_ in eng.layers:
        n = L_["kh"] * L_["kw"] * L_["cin"] * L_["cout"]
        assert np.abs(g[L_["kernel_off"]:L_["kernel_off"] + n]).max() > 0, L_["name"]

    # ---- layer-local: z of a SRC_PREV layer from the stored z of its predecessor, on a crop of one image ----
    wl = eng.get_weights()
    kernels, k = {}, 0
    for li, L_ in enumerate(eng.layers):
        kernels[li] = (wl[k], wl[k + 1]); k += 6 if L_["has_bn"] else 2
    names = [L_["name"] for L_ in eng.layers]
    for name, b, y0, x0 in (("enc0.conv1", 5, 100, 300), (f"dec{P - 1}.conv1", 37, 8, 640)):
        li = names.index(name); hh, ww = 24, 40
        zp = eng.debug_activation(li - 1, 0)[b, y0 - 1:y0 + hh + 1, x0 - 1:x0 + ww + 1].cpu().double().numpy()
        rec = eng.debug_bn_record(li - 1).cpu().numpy()
        a32 = np.maximum(rec[0] * zp.astype(np.float32) + rec[1], np.float32(0))        # the consumer's relu(a z + b), fp32
        act = _bf16_round(a32)
        w = _bf16_round(kernels[li][0]); bias = kernels[li][1].astype(np.float64)
        zr = np.zeros((hh, ww, w.shape[3]))
        for ky in range(3):
            for kx in range(3):
                zr += np.einsum("yxc,cm->yxm", act[ky:ky + hh, kx:kx + ww], w[ky, kx])
        zr += bias
        zh = eng.debug_activation(li, 0)[b, y0:y0 + hh, x0:x0 + ww].cpu().double().numpy()
        d = np.abs(zh - _bf16_round(zr)); ulp = _bf16_ulp(zr)
        assert (d <= ulp).mean() >= 0.999 and (d == 0).mean() >= 0.98, (name, float((d <= ulp).mean()), float((d == 0).mean()))

    # ---- the same step again: identical bits ----
    eng.set_dropout_step(2)
    eng.forward(x, training=True, labels=l, want_probs=False); eng.loss_dice(); eng.backward(l, macro=True)
    assert torch.equal(eng.grads, g1)
