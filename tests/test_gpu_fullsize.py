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
    # the BN-backward transform is applied on load; only the four half-resolution blocks whose thin-kernel instantiation
    # would spill keep the stand-alone pass (enc1.conv1, enc2.conv0, dec2.conv0, dec2.conv1)
    applied = {e["layer"] for e in ents if e["kernel"].startswith("bn_bwd_apply")}
    assert applied <= {"enc1.conv1", "enc2.conv0", "dec2.conv0", "dec2.conv1"}, applied
    assert sum(1 for n in names if ",gb" in n) >= 10, sorted(names)
    # ... and the full-resolution 3x3 layers reduce their backward-weights inside the backward-data launches
    assert {e["layer"] for e in ents if e["kernel"].endswith(",dw>")} == {"enc0.conv1", "dec3.conv0", "dec3.conv1"}
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
    layer's pre-BN output z (1e-4 of its scale) and the loss (1e-5).  Backward: every layer's dz from the head downwards
    and every gradient tensor.

    An fp32 path can only differ from the fp64 one through rounding, but on a ReLU net rounding is amplified at kinks: an
    element whose BN pre-activation is within fp32 rounding of zero flips its mask, is then wrong by its whole value, and
    the flip spreads (3x3 per conv, and through the BN-backward means to every element) as the gradient travels down.
    How much of that is inherent is MEASURED, not assumed: the same oracle run in fp32 (torch CPU, a different summation
    order, same inputs) against its own fp64 run gives the yardstick; the HIP path must stay within 3x of it (plus a
    floor), layer by layer -- quantile (share of elements beyond 5e-4 of the tensor's scale) and relative L2 -- and every
    kernel gradient within max(1e-3, 3x yardstick) relative L2."""
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
    names = [L_["name"] for L_ in eng.layers]

    def oracle(dtype):
        """(loss, [z per layer, NHWC], [dz per layer], [grad dict per layer]) of one forward + autograd backward"""
        tp, ts = ot.to_torch(params, state, dtype=dtype, requires_grad=True)
        zs = []
        npdt = np.float64 if dtype == torch.float64 else np.float32
        probs = ot.forward(cfg, tp, ts, torch.tensor(on.preprocess_u8(img, npdt)), training=True,
                           dropout_mask=torch.tensor(mask.astype(npdt)), collect_z=zs)
        y = torch.nn.functional.one_hot(torch.tensor(lab[..., 0].astype("int64")), C).to(dtype)
        loss = ot.dice_loss(y, probs, macro=True)
        loss.backward()
        z = [t.detach().permute(0, 2, 3, 1).numpy() for t in zs]
        dz = [t.grad.permute(0, 2, 3, 1).numpy().astype(np.float64) for t in zs]
        gr = [{k: v.grad.numpy().astype(np.float64).ravel() for k, v in p.items()} for p in tp]
        return float(loss.detach()), z, dz, gr

    loss64, z64, dz64, g64 = oracle(torch.float64)
    assert abs(float(loss4[0]) - loss64) < 1e-5, (float(loss4[0]), loss64)
    for li in range(nb):                                    # forward, per layer
        zh = eng.debug_activation(li, 0)[:B].cpu().numpy()
        err = np.abs(zh - z64[li]).max() / np.abs(z64[li]).max()
        assert err < 1e-4, f"layer {li} {names[li]}: z differs by {err:.2e} of its scale"
    del z64
    loss32, _, dz32, g32 = oracle(torch.float32)            # the yardstick: fp32 torch vs fp64 torch

    def dev(a, ref):
        scale = np.abs(ref).max()
        return float((np.abs(a - ref) > 5e-4 * scale).mean()), float(np.linalg.norm(a - ref) / np.linalg.norm(ref))

    rows = []
    for li in range(nb - 1, -1, -1):                        # backward, per layer from the head downwards
        qh, lh = dev(eng.debug_dz(li)[:B].cpu().numpy(), dz64[li])
        qy, ly = dev(dz32[li], dz64[li])
        rows.append((names[li], qh, qy, lh, ly))
    print("dz vs the fp64 oracle: layer, share of elements beyond 5e-4 of scale (HIP | fp32 torch), relative L2 (HIP | fp32 torch)")
    for r in rows:
        print(f"   {r[0]:12s} {r[1]:9.2e} {r[2]:9.2e}   {r[3]:9.2e} {r[4]:9.2e}")
    krows = []
    for li, L_ in enumerate(eng.layers):
        n = L_["kh"] * L_["kw"] * L_["cin"] * L_["cout"]
        gk = g[L_["kernel_off"]:L_["kernel_off"] + n]
        krows.append((L_["name"], float(np.linalg.norm(gk - g64[li]["kernel"]) / np.linalg.norm(g64[li]["kernel"])),
                      float(np.linalg.norm(g32[li]["kernel"] - g64[li]["kernel"]) / np.linalg.norm(g64[li]["kernel"]))))
    print("kernel gradients, relative L2 vs fp64 (HIP | fp32 torch):")
    for r in krows:
        print(f"   {r[0]:12s} {r[1]:9.2e} {r[2]:9.2e}")
    for name, qh, qy, lh, ly in rows:
        assert qh <= 3 * qy + 1e-4, f"{name}: {qh:.2e} of dz beyond 5e-4 of its scale (fp32 yardstick {qy:.2e})"
        assert lh <= 3 * ly + 1e-3, f"{name}: dz relative L2 {lh:.2e} (fp32 yardstick {ly:.2e})"
    for name, eh, ey in krows:
        assert eh <= max(1e-3, 3 * ey), f"{name}.kernel: relative L2 {eh:.2e} (fp32 yardstick {ey:.2e})"
    for li, L_ in enumerate(eng.layers):                    # bias / gamma / beta on the layer's kernel-gradient scale
        c = L_["cout"]
        kscale = np.abs(g64[li]["kernel"]).max()
        for key, off in [("bias", L_["bias_off"])] + ([("gamma", L_["gamma_off"]), ("beta", L_["beta_off"])] if L_["has_bn"] else []):
            rv = g64[li][key]
            scale = max(np.abs(rv).max(), kscale if key == "bias" else 0.0, 1e-12)
            eh = np.abs(g[off:off + c] - rv).max() / scale
            ey = np.abs(g32[li][key] - rv).max() / scale
            assert eh <= max(1e-3, 3 * ey), f"{L_['name']}.{key}: {eh:.2e} of scale (fp32 yardstick {ey:.2e})"


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


def test_statistics_finalized_in_the_producing_launch_agree_with_the_finalize_kernels_over_many_steps():
    """Option "fuse_bn_finalize": the BatchNorm records of the thin layers are written by the LAST block of the launch that
    emits the partial rows (write-through rows + arrival counter, csrc/kernels_fin.hpp) instead of by a bn_*_finalize launch.  A hand-off between
    workgroups of one launch that went wrong would show as a stale / partial sum in SOME step under load, so: 25 training
    steps at the benched configuration (B = 32, every CU busy, 768-block persistent grids), each compared with an engine
    that runs the separate finalize kernels on identical inputs and parameters -- records and moving statistics to fp32
    rounding (the two routes add the same rows in different orders), and the finalize launches of those layers are gone."""
    from oct_image_segmentation_models_amd import _hip
    from oct_image_segmentation_models_amd.engine import UNetEngine
    B = 32
    img, lab = scans(B, 51)
    x = torch.from_numpy(img).cuda(); l = torch.from_numpy(lab[..., 0].copy()).cuda()
    engs = {}
    try:
        for fused in (1, 0):
            _hip.set_option("fuse_bn_finalize", fused)
            engs[fused] = UNetEngine(device="cuda:0", input_channels=1, num_classes=C, image_height=H, image_width=W,
                                     max_batch=B, training=True, seed=5, init_seed=3)
    finally:
        _hip.set_option("fuse_bn_finalize", 0)          # (the default: the route measured 0.5-1 % slower per step)
    nb = len(engs[1].layers) - 1
    for step in range(25):
        recs = {}
        for fused, eng in engs.items():
            eng.set_dropout_step(step)
            if step == 0:
                eng.profile_begin()
            eng.forward(x, training=True, labels=l, want_probs=False); eng.loss_dice(); eng.backward(l, macro=True)
            if step == 0:
                fins = [e for e in eng.profile_end() if e["kernel"].startswith("bn_") and "finalize" in e["kernel"]]
                n = sum(e["launches"] for e in fins)
                assert (n <= 2 * nb - 18) if fused else (n == 2 * nb), (fused, n)        # >= 18 of the 44 launches are gone
            recs[fused] = ([eng.debug_bn_record(li).clone() for li in range(nb)], eng.state.clone(), eng.grads.clone())
            eng.adam_step(lr=1e-3)
        engs[0].params.copy_(engs[1].params)          # keep the two engines on identical parameters
        engs[0].state.copy_(engs[1].state)
        for li in range(nb):
            a, b = recs[1][0][li].double(), recs[0][0][li].double()
            scale = b.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)
            assert ((a - b).abs() / scale).max() < 2e-5, (step, engs[1].layers[li]["name"], ((a - b).abs() / scale).amax(dim=1))
        assert (recs[1][1] - recs[0][1]).abs().max() < 1e-6 * max(1.0, float(recs[0][1].abs().max()))
        g1, g0 = recs[1][2].double(), recs[0][2].double()
        assert (g1 - g0).norm() <= 1e-4 * g0.norm(), step
