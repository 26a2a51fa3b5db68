"""Analytic known-answer tests of the HIP path that do NOT go through the oracle: each pins one
of the Keras-2.9 semantics SURVEY Appendix B could not verify against TensorFlow, directly on the device kernels,
with hand-set weights and closed-form expected values.

* B.1  Conv2D 2x2 "same" pads bottom/right (0 before, 1 after) and is a cross-correlation (no kernel flip);
       B.5 UpSampling2D is nearest (out[i,j] = in[i//2, j//2]); B.4 MaxPooling2D 2x2 stride 2.
* B.2  BatchNormalization in training mode maps a constant tensor to beta, and its moving statistics move by
       (1 - momentum) towards (batch mean, batch variance = 0).
* B.7 / custom_losses.py:47-81  softmax of equal logits is uniform, and the Dice losses of a uniform prediction
       have the closed form (2 n_c / C + s) / (n_c + N / C + s).
Reference graph: /root/reference/oct_image_segmentation_models/models/unet.py:20-57,106-153."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

H, W, SN = 16, 32, 4


def engine(C=3, training=False, B=1):
    from oct_image_segmentation_models_amd.engine import UNetEngine
    return UNetEngine(device="cuda:0", input_channels=1, num_classes=C, image_height=H, image_width=W, start_neurons=SN,
                      pool_layers=1, conv_layers=1, max_batch=B, training=training, seed=1, dropout_rate=0.0)


def identity_weights(eng, up_kernel=None):
    """Every 3x3 conv passes channel 0 through its centre tap, every BN is the identity on moving statistics
    (gamma 1, beta 0, mean 0, var 1 - eps), the 2x2 up-conv maps channel 0 -> channel 0 with ``up_kernel``."""
    ws = []
    for L in eng.layers:
        k = np.zeros((L["kh"], L["kw"], L["cin"], L["cout"]), np.float32)
        if L["kh"] == 3:
            k[1, 1, 0, 0] = 1.0
        elif L["kh"] == 2 and up_kernel is not None:
            k[:, :, 0, 0] = up_kernel
        ws += [k, np.zeros(L["cout"], np.float32)]
        if L["has_bn"]:
            c = L["cout"]
            ws += [np.ones(c, np.float32), np.zeros(c, np.float32), np.zeros(c, np.float32), np.full(c, 1.0 - 1e-3, np.float32)]
    return ws


def test_up_conv_pads_bottom_right_nearest_upsampling_and_no_kernel_flip():
    eng = engine()
    names = [L["name"] for L in eng.layers]
    assert names == ["enc0.conv0", "mid.conv0", "dec0.up", "dec0.conv0", "head"]
    wk = np.array([[1.0, 2.0], [4.0, 8.0]], np.float32)        # w[ky][kx], all sums of subsets distinct
    eng.set_weights(identity_weights(eng, wk))
    y0, x0 = 6, 10                                             # even: the 2x2 pool window / upsampled block is aligned
    img = np.zeros((1, H, W, 1), np.uint8); img[0, y0, x0, 0] = 255
    eng.forward(torch.from_numpy(img).cuda(), training=False)
    torch.cuda.synchronize()
    z_enc = eng.debug_activation(0, 0)[0, :, :, 0].cpu().numpy()
    z_mid = eng.debug_activation(1, 0)[0, :, :, 0].cpu().numpy()
    z_up = eng.debug_activation(2, 0)[0, :, :, 0].cpu().numpy()
    want = np.zeros((H, W), np.float32); want[y0, x0] = 1.0
    assert np.allclose(z_enc, want, atol=1e-6)                                   # u8 255 -> 1.0, centre tap
    wm = np.zeros((H // 2, W // 2), np.float32); wm[y0 // 2, x0 // 2] = 1.0
    assert np.allclose(z_mid, wm, atol=1e-6)                                     # 2x2/stride-2 max pool keeps the delta
    # nearest upsampling gives ones on {y0, y0+1} x {x0, x0+1}; out[y, x] = sum_{ky,kx} w[ky, kx] * in[y + ky, x + kx]
    # (pad bottom/right, cross-correlation).  Pad top/left would put w[0,0] at (y0+2, x0+2); a flipped kernel would
    # put w[0,0] instead of w[1,1] at (y0-1, x0-1).
    exp = np.zeros((H, W), np.float32)
    for y in range(y0 - 1, y0 + 2):
        for x in range(x0 - 1, x0 + 2):
            exp[y, x] = sum(wk[ky, kx] for ky in range(2) for kx in range(2)
                            if y0 <= y + ky <= y0 + 1 and x0 <= x + kx <= x0 + 1)
    assert exp[y0 - 1, x0 - 1] == 8.0 and exp[y0, x0] == 15.0 and exp[y0 + 1, x0 + 1] == 1.0 and exp[y0 + 2, x0 + 2] == 0.0
    assert np.allclose(z_up, exp, atol=1e-5), (z_up[y0 - 2:y0 + 4, x0 - 2:x0 + 4], exp[y0 - 2:y0 + 4, x0 - 2:x0 + 4])
    # bottom/right image border: a delta in the last pooled cell must not read past the edge
    img2 = np.zeros((1, H, W, 1), np.uint8); img2[0, H - 2, W - 2, 0] = 255
    eng.forward(torch.from_numpy(img2).cuda(), training=False)
    torch.cuda.synchronize()
    z2 = eng.debug_activation(2, 0)[0, :, :, 0].cpu().numpy()
    assert np.allclose(z2[H - 1, W - 1], wk[0, 0], atol=1e-5) and np.allclose(z2[H - 2, W - 2], wk.sum(), atol=1e-5)


def test_batchnorm_training_maps_a_constant_tensor_to_beta_and_moves_the_moving_stats():
    B = 2
    eng = engine(training=True, B=B)
    ws = identity_weights(eng, np.ones((2, 2), np.float32))
    # first block: z = 0.75 * x + 0.125 everywhere (centre tap only, so no border effect); gamma 1.7, beta -0.3 / 0.4
    ws[0][1, 1, 0, :] = 0.75; ws[1][:] = 0.125
    ws[2][:] = 1.7; ws[3][:] = np.array([-0.3, 0.4, 0.0, 0.25], np.float32)
    ws[4][:] = 0.5; ws[5][:] = 2.0                               # moving mean / variance before the step
    eng.set_weights(ws)
    img = np.full((B, H, W, 1), 204, np.uint8)                   # 204 / 255 = 0.8
    lab = torch.zeros((B, H, W), dtype=torch.uint8, device="cuda")
    eng.forward(torch.from_numpy(img).cuda(), training=True, labels=lab, want_probs=False)
    torch.cuda.synchronize()
    zc = np.float32(0.75) * np.float32(204 / 255.0) + np.float32(0.125)
    z = eng.debug_activation(0, 0)[:B].cpu().numpy()
    assert np.allclose(z, zc, atol=1e-6)
    rec = eng.debug_bn_record(0).cpu().numpy()                  # rows a, b: the consumer applies relu(a * z + b)
    y = rec[0] * zc + rec[1]
    assert np.allclose(y, ws[3], atol=2e-4), y                  # (z - mean) * rstd = 0  ->  y = beta  (rstd = eps^-1/2 ~ 31.6)
    assert np.allclose(rec[2], zc, atol=1e-6)                   # batch mean
    assert np.allclose(rec[3], 1.0 / np.sqrt(1e-3), rtol=1e-4)  # rstd of a zero-variance batch
    st = eng.state.cpu().numpy()
    mm = st[eng.layers[0]["moving_mean_off"]:eng.layers[0]["moving_mean_off"] + SN]
    mv = st[eng.layers[0]["moving_var_off"]:eng.layers[0]["moving_var_off"] + SN]
    assert np.allclose(mm, 0.99 * 0.5 + 0.01 * zc, atol=1e-6)   # momentum 0.99 (Appendix B.2)
    assert np.allclose(mv, 0.99 * 2.0, atol=1e-6)               # + 0.01 * 0 (biased or Bessel-corrected: both 0)


@pytest.mark.parametrize("C", [3, 4])
def test_uniform_softmax_and_dice_closed_form(C):
    B = 2
    eng = engine(C=C, training=True, B=B)
    ws = identity_weights(eng, np.ones((2, 2), np.float32))
    ws[-2][:] = 0.0; ws[-1][:] = 0.0                             # head: zero kernel, zero bias -> equal logits
    eng.set_weights(ws)
    rng = np.random.default_rng(C)
    img = rng.integers(0, 256, (B, H, W, 1)).astype(np.uint8)
    lab = rng.integers(0, C, (B, H, W)).astype(np.uint8)
    lab[1, :, : W // 2] = 0                                      # unequal class counts per sample
    probs, _ = eng.forward(torch.from_numpy(img).cuda(), training=True, labels=torch.from_numpy(lab).cuda())
    loss4 = eng.loss_dice(smooth=1e-5).cpu().numpy()
    assert torch.allclose(probs, torch.full_like(probs, 1.0 / C), rtol=0, atol=1e-7)   # softmax(0, ..., 0) = 1/C
    s, N = 1e-5, H * W
    n = np.stack([(lab == c).reshape(B, -1).sum(1) for c in range(C)], axis=1).astype(np.float64)      # (B, C)
    macro = 1.0 - ((2.0 * n / C + s) / (n + N / C + s)).mean()                       # custom_losses.py:65-81
    micro = 1.0 - (2.0 * n.sum() / C + s) / (n.sum() + B * N + s)                    # custom_losses.py:47-62 (sum p = B*N)
    assert abs(loss4[0] - macro) < 1e-6 and abs(loss4[1] - micro) < 1e-6, (loss4, macro, micro)
    # training monitors threshold p > 0.5: nothing passes at p = 1/C  ->  macro metric = eps / (n_c + eps), micro = 0
    eps = 1e-5
    assert abs(loss4[2] - (eps / (n + eps)).mean()) < 1e-6 and loss4[3] == 0.0
