"""CPU oracle for the OCT U-Net hot path -- TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of the algorithm the reference delegates to
TensorFlow 2.9 / Keras for its U-Net forward/backward path.  It is the checker
the HIP path is compared against; nothing under ``oracle/`` is imported by the
product package (only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may use it).

PARITY UNPINNED: the arithmetic of this path lives in the third-party
dependency ``tensorflow==2.9.0`` (reference ``pyproject.toml:31``), which is not
vendored under /root/reference and is not installed in the build container, and
the reference ships no tests, golden vectors or fixtures for this path
(SURVEY.md section 8c).  The oracle therefore restates the *published* Keras
layer semantics anchored on the reference's own call
sites, and is cross-checked against an independent torch-CPU/autograd
restatement (``oracle/unet_torch.py``) and analytic known answers in
``tests/``.

Reference call sites restated here (paths relative to
``/root/reference/oct_image_segmentation_models``):

* topology, op order, concat order, defaults ......... models/unet.py:20-57,106-153
* preprocess x/255 ................................... models/unet.py:87-91
* dice_loss_micro / dice_loss_macro .................. common/custom_losses.py:47-81
* dice_coef_micro / dice_coef_macro / soft_dice_class  common/custom_metrics.py:18-100
* argmax / one-hot / boundary maps ................... common/utils.py:73-168
* num_classes, to_categorical, compile ............... training/training.py:176-227,243-266

All tensors are NHWC (Keras ``channels_last``, training.py:164-165).  All
functions take/return numpy arrays and compute in the dtype of their inputs
(float64 for the definitional oracle, float32 to mimic device rounding).
"""
from __future__ import annotations

import dataclasses
from typing import Dict, List, Optional, Tuple

import numpy as np

# ----------------------------------------------------------------------------
# configuration / layer plan
# ----------------------------------------------------------------------------


@dataclasses.dataclass
class UNetConfig:
    """Constructor arguments of ``UNet`` (models/unet.py:62-85) + Keras defaults."""

    input_channels: int = 1
    num_classes: int = 3
    start_neurons: int = 8
    pool_layers: int = 4
    conv_layers: int = 2
    enc_kernel: Tuple[int, int] = (3, 3)
    dec_kernel: Tuple[int, int] = (2, 2)
    bn_eps: float = 1e-3  # keras BatchNormalization default (Appendix B.2)
    bn_momentum: float = 0.99
    dropout_rate: float = 0.5  # models/unet.py:130
    # TF-2.9 fused BN feeds the Bessel-corrected batch variance into the moving
    # average (documented assumption) -- isolated switch.
    bn_unbiased_moving_var: bool = True


@dataclasses.dataclass
class ConvSpec:
    """One Conv2D(+BN+ReLU) node of the graph in Keras creation order."""

    name: str
    kh: int
    kw: int
    cin: int
    cout: int
    level: int  # resolution level of the OUTPUT (0 = full resolution)
    has_bn: bool
    src: str  # "input" | "prev" | "pool" | "up" | "concat" | "head"
    skip_from: int = -1  # conv index whose output is the skip half of a concat


def build_plan(cfg: UNetConfig) -> List[ConvSpec]:
    """Conv nodes in creation order -- models/unet.py:106-153."""
    sn, P, L = cfg.start_neurons, cfg.pool_layers, cfg.conv_layers
    ekh, ekw = cfg.enc_kernel
    dkh, dkw = cfg.dec_kernel
    plan: List[ConvSpec] = []
    cin = cfg.input_channels
    enc_last: List[int] = []
    for i in range(P):  # encoder, unet.py:113-121 -> unet_enc_block :32-38
        size = sn * (2 ** i)
        for j in range(L):
            src = "input" if (i == 0 and j == 0) else ("pool" if j == 0 else "prev")
            plan.append(ConvSpec(f"enc{i}.conv{j}", ekh, ekw, cin, size, i, True, src))
            cin = size
        enc_last.append(len(plan) - 1)
    size = sn * (2 ** P)  # bottleneck, unet.py:123-129
    for j in range(L):
        src = ("pool" if P > 0 else "input") if j == 0 else "prev"
        plan.append(ConvSpec(f"mid.conv{j}", ekh, ekw, cin, size, P, True, src))
        cin = size
    for i in range(P):  # decoder, unet.py:132-140 -> unet_dec_block :47-57
        lvl = P - 1 - i
        size = sn * (2 ** lvl)
        plan.append(ConvSpec(f"dec{i}.up", dkh, dkw, cin, size, lvl, True, "up"))
        cin = 2 * size  # concatenate([x, concat_map]) unet.py:52
        for j in range(L):
            if j == 0:
                plan.append(ConvSpec(f"dec{i}.conv{j}", ekh, ekw, cin, size, lvl, True,
                                     "concat", skip_from=enc_last[lvl]))
            else:
                plan.append(ConvSpec(f"dec{i}.conv{j}", ekh, ekw, cin, size, lvl, True, "prev"))
            cin = size
    plan.append(ConvSpec("head", 1, 1, cin, cfg.num_classes, 0, False, "head"))  # unet.py:142-147
    return plan


def param_count(cfg: UNetConfig) -> Tuple[int, int]:
    """(trainable, BN-moving-state) counts; 487403 / 1712 at the default config, C=3."""
    t = s = 0
    for c in build_plan(cfg):
        t += c.kh * c.kw * c.cin * c.cout + c.cout
        if c.has_bn:
            t += 2 * c.cout
            s += 2 * c.cout
    return t, s


def init_params(cfg: UNetConfig, seed: int = 0, dtype=np.float64,
                randomize_bn: bool = False) -> Tuple[List[Dict[str, np.ndarray]], List[Dict[str, np.ndarray]]]:
    """Keras initialisers (Appendix B.1/B.2): glorot_uniform kernel, zero bias,
    gamma=1, beta=0, moving_mean=0, moving_var=1.  ``randomize_bn`` perturbs
    bias/gamma/beta/moving stats so tests exercise every term."""
    rng = np.random.default_rng(seed)
    params, state = [], []
    for c in build_plan(cfg):
        fan_in, fan_out = c.kh * c.kw * c.cin, c.kh * c.kw * c.cout
        lim = np.sqrt(6.0 / (fan_in + fan_out))
        p = {"kernel": rng.uniform(-lim, lim, (c.kh, c.kw, c.cin, c.cout)).astype(dtype),
             "bias": np.zeros(c.cout, dtype)}
        if randomize_bn:
            p["bias"] = rng.normal(0, 0.1, c.cout).astype(dtype)
        if c.has_bn:
            p["gamma"] = np.ones(c.cout, dtype)
            p["beta"] = np.zeros(c.cout, dtype)
            st = {"moving_mean": np.zeros(c.cout, dtype), "moving_var": np.ones(c.cout, dtype)}
            if randomize_bn:
                # includes a few negative gammas: max-pool does not commute with BN then
                p["gamma"] = rng.normal(1.0, 0.3, c.cout).astype(dtype)
                p["beta"] = rng.normal(0, 0.2, c.cout).astype(dtype)
                st["moving_mean"] = rng.normal(0, 0.1, c.cout).astype(dtype)
                st["moving_var"] = rng.uniform(0.5, 1.5, c.cout).astype(dtype)
            state.append(st)
        params.append(p)
    return params, state


# flat-buffer exchange layouts -------------------------------------------------

def flatten_params(params: List[Dict[str, np.ndarray]]) -> np.ndarray:
    """Trainable buffer layout of the C ABI: per conv [kernel(HWIO), bias, gamma, beta]."""
    out = []
    for p in params:
        out += [p["kernel"].ravel(), p["bias"].ravel()]
        if "gamma" in p:
            out += [p["gamma"].ravel(), p["beta"].ravel()]
    return np.concatenate(out)


def unflatten_params(cfg: UNetConfig, flat: np.ndarray) -> List[Dict[str, np.ndarray]]:
    params, o = [], 0
    for c in build_plan(cfg):
        n = c.kh * c.kw * c.cin * c.cout
        p = {"kernel": flat[o:o + n].reshape(c.kh, c.kw, c.cin, c.cout).copy()}
        o += n
        p["bias"] = flat[o:o + c.cout].copy(); o += c.cout
        if c.has_bn:
            p["gamma"] = flat[o:o + c.cout].copy(); o += c.cout
            p["beta"] = flat[o:o + c.cout].copy(); o += c.cout
        params.append(p)
    assert o == flat.size
    return params


def flatten_state(state: List[Dict[str, np.ndarray]]) -> np.ndarray:
    return np.concatenate([np.concatenate([s["moving_mean"], s["moving_var"]]) for s in state])


def unflatten_state(cfg: UNetConfig, flat: np.ndarray) -> List[Dict[str, np.ndarray]]:
    st, o = [], 0
    for c in build_plan(cfg):
        if c.has_bn:
            st.append({"moving_mean": flat[o:o + c.cout].copy(),
                       "moving_var": flat[o + c.cout:o + 2 * c.cout].copy()})
            o += 2 * c.cout
    assert o == flat.size
    return st


def keras_weight_list(params, state) -> List[np.ndarray]:
    """``model.get_weights()`` order (Appendix B.9): per layer in creation order,
    Conv2D -> [kernel, bias]; BatchNormalization -> [gamma, beta, moving_mean, moving_var]."""
    out, bi = [], 0
    for p in params:
        out += [p["kernel"], p["bias"]]
        if "gamma" in p:
            out += [p["gamma"], p["beta"], state[bi]["moving_mean"], state[bi]["moving_var"]]
            bi += 1
    return out


# ----------------------------------------------------------------------------
# primitive ops (forward)
# ----------------------------------------------------------------------------

def same_pad(k: int) -> Tuple[int, int]:
    """TF 'same' padding at stride 1: k-1 total, floor((k-1)/2) before, rest
    after => 3x3: (1,1); 2x2: (0,1).  Appendix B.1."""
    return (k - 1) // 2, (k - 1) - (k - 1) // 2


def conv2d_same(x: np.ndarray, kernel: np.ndarray, bias: np.ndarray) -> np.ndarray:
    """Conv2D(strides 1, padding 'same', dilation 1): cross-correlation, HWIO
    kernel, plus bias -- models/unet.py:27."""
    kh, kw, cin, cout = kernel.shape
    B, H, W, _ = x.shape
    (pt, pb), (pl, pr) = same_pad(kh), same_pad(kw)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    out = np.zeros((B, H, W, cout), dtype=x.dtype)
    for ky in range(kh):
        for kx in range(kw):
            out += xp[:, ky:ky + H, kx:kx + W, :] @ kernel[ky, kx]
    return out + bias


def batchnorm_train(z, gamma, beta, eps):
    """BatchNormalization, training=True: batch mean and BIASED variance over
    (N,H,W) -- models/unet.py:21, Appendix B.2."""
    mean = z.mean(axis=(0, 1, 2))
    var = ((z - mean) ** 2).mean(axis=(0, 1, 2))
    rstd = 1.0 / np.sqrt(var + eps)
    xhat = (z - mean) * rstd
    return gamma * xhat + beta, mean, var, rstd, xhat


def batchnorm_infer(z, gamma, beta, mm, mv, eps):
    return gamma * (z - mm) / np.sqrt(mv + eps) + beta


def relu(x):
    return np.maximum(x, 0)


def maxpool2x2(x):
    """MaxPooling2D(pool_size=(2,2)): stride 2, 'valid' -- models/unet.py:37."""
    B, H, W, C = x.shape
    return x[:, :H // 2 * 2, :W // 2 * 2, :].reshape(B, H // 2, 2, W // 2, 2, C).max(axis=(2, 4))


def upsample2x(x):
    """UpSampling2D() nearest, size (2,2): out[i,j] = in[i//2, j//2] -- models/unet.py:42."""
    return x.repeat(2, axis=1).repeat(2, axis=2)


def softmax(z):
    e = np.exp(z - z.max(axis=-1, keepdims=True))
    return e / e.sum(axis=-1, keepdims=True)


def one_hot(labels: np.ndarray, C: int, dtype) -> np.ndarray:
    """to_categorical on (N,H,W[,1]) labels -- training.py:225-227."""
    lab = labels.reshape(labels.shape[0], labels.shape[1], labels.shape[2]).astype(np.int64)
    return np.eye(C, dtype=dtype)[lab]


# ----------------------------------------------------------------------------
# loss + metrics (custom_losses.py / custom_metrics.py)
# ----------------------------------------------------------------------------

def dice_loss_macro(y, p, smooth=1e-5):
    """custom_losses.py:65-81: per (b,c) soft Dice over H,W; 1 - mean."""
    I = (y * p).sum(axis=(1, 2)); T = y.sum(axis=(1, 2)); Pp = p.sum(axis=(1, 2))
    return 1.0 - ((2.0 * I + smooth) / (T + Pp + smooth)).mean()


def dice_loss_micro(y, p, smooth=1e-5):
    """custom_losses.py:47-62: one global ratio."""
    I = (y * p).sum(); T = y.sum(); Pp = p.sum()
    return 1.0 - (2.0 * I + smooth) / (T + Pp + smooth)


def dice_loss_grad(y, p, macro: bool, smooth=1e-5):
    """dL/dp, closed form."""
    if macro:
        B, _, _, C = p.shape
        I = (y * p).sum(axis=(1, 2), keepdims=True)
        D = y.sum(axis=(1, 2), keepdims=True) + p.sum(axis=(1, 2), keepdims=True) + smooth
        return -(2.0 * y * D - (2.0 * I + smooth)) / (D * D) / (B * C)
    I = (y * p).sum(); D = y.sum() + p.sum() + smooth
    return -(2.0 * y * D - (2.0 * I + smooth)) / (D * D)


FOCAL_EPS = 1e-7   # keras backend epsilon: the probabilities are clipped to [eps, 1 - eps] before the logarithm


def focal_loss_mean(labels, p, gamma=2.0, class_weight=None, clip_modulation=False):
    """Focal half of ``SparseCategoricalFocalDiceLoss.call`` (custom_losses.py:98-160): the per-pixel sparse
    categorical focal loss  cw[y] * (1 - p_y)^gamma * (-log p_y)  of the third-party ``focal-loss==0.0.7``
    (``SparseCategoricalFocalLoss``, probabilities clipped to [eps, 1-eps]) summed and divided by the number of label
    elements, custom_losses.py:150-153.  PARITY UNPINNED (third-party package absent; published formula restated)."""
    lab = np.asarray(labels).reshape(p.shape[:-1]).astype(np.int64)
    raw = np.take_along_axis(p, lab[..., None], axis=-1)[..., 0]
    py = np.clip(raw, FOCAL_EPS, 1.0 - FOCAL_EPS)
    w = 1.0 if class_weight is None else np.asarray(class_weight, p.dtype)[lab]
    # ``clip_modulation``: whether (1 - p)^gamma sees the clipped probability too -- unverifiable here (package absent);
    # the engine exposes the same switch (oct_set_option "focal_clip_modulation", default 0 = logarithm only)
    return float((w * (1.0 - (py if clip_modulation else raw)) ** gamma * -np.log(py)).sum() / lab.size)


def focal_loss_grad(labels, p, gamma=2.0, class_weight=None, clip_modulation=False):
    """d focal_loss_mean / d p (non-zero only at the true class).  Where the clip is active the logarithm is constant;
    with ``clip_modulation`` the modulation is too and the derivative vanishes there."""
    lab = np.asarray(labels).reshape(p.shape[:-1]).astype(np.int64)
    raw = np.take_along_axis(p, lab[..., None], axis=-1)[..., 0]
    py = np.clip(raw, FOCAL_EPS, 1.0 - FOCAL_EPS)
    w = 1.0 if class_weight is None else np.asarray(class_weight, p.dtype)[lab]
    inr = (raw >= FOCAL_EPS) & (raw <= 1.0 - FOCAL_EPS)
    q = 1.0 - raw
    d = w * (gamma * q ** (gamma - 1.0) * np.log(py) - np.where(inr, q ** gamma / py, 0.0)) / lab.size
    if clip_modulation:
        d = np.where(inr, d, 0.0)
    g = np.zeros_like(p)
    np.put_along_axis(g, lab[..., None], d[..., None], axis=-1)
    return g


def focal_dice_loss(labels, p, num_classes, gamma=2.0, class_weight=None, focal_loss_weight=0.5, dice_macro=True,
                    smooth=1e-5):
    """``focal_dice_loss`` (custom_losses.py:163-178):  w * focal + (1 - w) * dice."""
    y = one_hot(labels, num_classes, p.dtype)
    dice = dice_loss_macro(y, p, smooth) if dice_macro else dice_loss_micro(y, p, smooth)
    return focal_loss_weight * focal_loss_mean(labels, p, gamma, class_weight) + (1.0 - focal_loss_weight) * dice


def dice_coef_macro(y, p, eps=1e-5):
    """Training monitor, custom_metrics.py:48-77: hard threshold p>0.5."""
    ph = (p > 0.5).astype(p.dtype)
    I = (y * ph).sum(axis=(1, 2)); T = y.sum(axis=(1, 2)); Pp = ph.sum(axis=(1, 2))
    return ((2.0 * I + eps) / (T + Pp + eps)).mean()


def dice_coef_micro(y, p):
    """custom_metrics.py:18-45: no epsilon (0/0 -> nan, as the reference)."""
    ph = (p > 0.5).astype(p.dtype)
    with np.errstate(invalid="ignore", divide="ignore"):
        return 2.0 * (y * ph).sum() / (y.sum() + ph.sum())


def soft_dice_class(y_true, y_pred, eps=1e-5):
    """custom_metrics.py:86-100 (eval metric; inputs (b,c,H,W))."""
    axes = tuple(range(2, y_pred.ndim))
    inter = np.sum(y_pred * y_true, axis=axes)
    denom = np.sum(y_pred + y_true, axis=axes)
    return (2.0 * inter + eps) / (denom + eps)


# ----------------------------------------------------------------------------
# post-step (common/utils.py:80-168)
# ----------------------------------------------------------------------------

def perform_argmax(predictions: np.ndarray):
    """utils.py:80-112 with bin=True, channels_last: returns (argmax (n,H,W),
    one-hot (n,C,H,W) float32)."""
    C = predictions.shape[3]
    am = np.argmax(predictions, axis=3)
    cat = np.transpose(np.eye(C, dtype=np.float32)[am], (0, 3, 1, 2))
    return am, cat


def convert_predictions_to_maps_semantic(categorical_pred, bg_ilm=True, bg_csi=False):
    """utils.py:115-168: vertical-gradient boundary maps, uint8 (n,C-1,H,W)."""
    n, C, H, W = categorical_pred.shape
    out = np.zeros((n, C - 1, H, W), dtype=np.uint8)
    for s in range(n):
        for m in range(1, C):
            flip = (m == 1 and bg_ilm) or (m == C - 1 and bg_csi)
            cur = categorical_pred[s, m - 1 if flip else m].astype(np.float64)
            g = np.gradient(cur, axis=0)
            if flip:
                g = -g
            g[g < 0] = 0
            g *= 2
            g = g - np.roll(g, -1, axis=0)
            g[g < 0] = 0
            out[s, m - 1] = (g * 255).astype(np.uint8)
    return out


# ----------------------------------------------------------------------------
# whole-network forward / backward
# ----------------------------------------------------------------------------

def preprocess_u8(images_u8: np.ndarray, dtype) -> np.ndarray:
    """x/255.0 (models/unet.py:87-91); bit-identical float32 table per Appendix B.11."""
    lut = (np.arange(256) / 255.0).astype(np.float32)
    return lut[images_u8].astype(dtype)


def _bn_act(cfg, spec, p, st, z, training, cache_entry):
    if training:
        yb, mean, var, rstd, xhat = batchnorm_train(z, p["gamma"], p["beta"], cfg.bn_eps)
        cache_entry.update(mean=mean, var=var, rstd=rstd, xhat=xhat)
    else:
        yb = batchnorm_infer(z, p["gamma"], p["beta"], st["moving_mean"], st["moving_var"], cfg.bn_eps)
    y = relu(yb)
    cache_entry["y"] = y
    return y


def forward(cfg: UNetConfig, params, state, x: np.ndarray, training: bool = False,
            dropout_mask: Optional[np.ndarray] = None):
    """Graph of ``UNet.build_model`` (models/unet.py:106-153).

    x: (B,H,W,Cin) already preprocessed.  ``dropout_mask`` (B,h,w,c) of {0,1}
    is the keep-mask applied (x mask / (1-rate)) after the bottleneck when
    ``training`` -- the reference's TF RNG stream cannot be reproduced, so the
    mask is an input.  Returns (probs, cache).
    """
    plan = build_plan(cfg)
    P = cfg.pool_layers
    cache: List[dict] = [dict() for _ in plan]
    pooled: Dict[int, np.ndarray] = {}
    bn_idx = 0
    cur = x
    out_of: Dict[int, np.ndarray] = {}
    for li, spec in enumerate(plan):
        p = params[li]
        if spec.src in ("input", "prev"):
            inp = cur
        elif spec.src == "pool":
            inp = maxpool2x2(cur)
            cache[li]["pool_in"] = cur
        elif spec.src == "up":
            inp = upsample2x(cur)
        elif spec.src == "concat":
            inp = np.concatenate([cur, out_of[spec.skip_from]], axis=-1)  # [up, skip] unet.py:52
        elif spec.src == "head":
            inp = cur
        cache[li]["x"] = inp
        z = conv2d_same(inp, p["kernel"], p["bias"])
        cache[li]["z"] = z
        if spec.has_bn:
            st = state[bn_idx]; bn_idx += 1
            cur = _bn_act(cfg, spec, p, st, z, training, cache[li])
            out_of[li] = cur
            if spec.name == f"mid.conv{cfg.conv_layers - 1}" and training and cfg.dropout_rate > 0:
                # Dropout(0.5) unet.py:130 -- inverted scaling
                assert dropout_mask is not None, "training forward needs an explicit dropout keep-mask"
                cache[li]["drop"] = dropout_mask.astype(z.dtype) / (1.0 - cfg.dropout_rate)
                cur = cur * cache[li]["drop"]
        else:
            probs = softmax(z)
            cache[li]["probs"] = probs
    return probs, cache


def updated_moving_stats(cfg: UNetConfig, state, cache):
    """moving <- moving*momentum + batch*(1-momentum) (Appendix B.2)."""
    plan = build_plan(cfg)
    new, bi = [], 0
    m = cfg.bn_momentum
    for li, spec in enumerate(plan):
        if not spec.has_bn:
            continue
        z = cache[li]["z"]
        n = z.shape[0] * z.shape[1] * z.shape[2]
        var = cache[li]["var"]
        if cfg.bn_unbiased_moving_var and n > 1:
            var = var * (n / (n - 1.0))
        new.append({"moving_mean": state[bi]["moving_mean"] * m + cache[li]["mean"] * (1 - m),
                    "moving_var": state[bi]["moving_var"] * m + var * (1 - m)})
        bi += 1
    return new


def _conv_backward(x, kernel, dz):
    kh, kw, cin, cout = kernel.shape
    B, H, W, _ = x.shape
    (pt, pb), (pl, pr) = same_pad(kh), same_pad(kw)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    gxp = np.zeros_like(xp)
    dk = np.zeros_like(kernel)
    for ky in range(kh):
        for kx in range(kw):
            xs = xp[:, ky:ky + H, kx:kx + W, :]
            dk[ky, kx] = np.tensordot(xs, dz, axes=([0, 1, 2], [0, 1, 2]))
            gxp[:, ky:ky + H, kx:kx + W, :] += dz @ kernel[ky, kx].T
    return gxp[:, pt:pt + H, pl:pl + W, :], dk, dz.sum(axis=(0, 1, 2))


def _maxpool_backward(x, g):
    """Gradient to the first maximum of each 2x2 window (row-major scan)."""
    B, H, W, C = x.shape
    xw = x.reshape(B, H // 2, 2, W // 2, 2, C).transpose(0, 1, 3, 5, 2, 4).reshape(B, H // 2, W // 2, C, 4)
    idx = xw.argmax(axis=-1)
    gw = np.zeros_like(xw)
    np.put_along_axis(gw, idx[..., None], g[..., None], axis=-1)
    return gw.reshape(B, H // 2, W // 2, C, 2, 2).transpose(0, 1, 4, 2, 5, 3).reshape(B, H, W, C)


def backward(cfg: UNetConfig, params, cache, labels: np.ndarray, macro: bool = True,
             smooth: float = 1e-5, loss_scale: float = 1.0, focal=None, focal_clip_modulation=False):
    """Hand-derived reverse pass of ``forward(training=True)`` for the Dice losses, or -- with
    ``focal = (focal_loss_weight, gamma, class_weight or None)`` -- for ``focal_dice_loss``.
    Returns (loss, grads) with grads in the same structure as ``params``."""
    plan = build_plan(cfg)
    probs = cache[-1]["probs"]
    y = one_hot(labels, cfg.num_classes, probs.dtype)
    loss = dice_loss_macro(y, probs, smooth) if macro else dice_loss_micro(y, probs, smooth)
    dp = dice_loss_grad(y, probs, macro, smooth) * loss_scale
    if focal is not None:
        fw, gamma, cw = focal
        loss = fw * focal_loss_mean(labels, probs, gamma, cw, focal_clip_modulation) + (1.0 - fw) * loss
        dp = (1.0 - fw) * dp + fw * loss_scale * focal_loss_grad(labels, probs, gamma, cw, focal_clip_modulation)
    dz = probs * (dp - (probs * dp).sum(axis=-1, keepdims=True))  # softmax Jacobian
    grads: List[dict] = [dict() for _ in plan]
    g_out: Dict[int, np.ndarray] = {}  # gradient wrt the (activated) output of conv li

    def add(li, g):
        g_out[li] = g_out[li] + g if li in g_out else g

    for li in range(len(plan) - 1, -1, -1):
        spec, p, c = plan[li], params[li], cache[li]
        if spec.has_bn:
            g = g_out.pop(li)
            if "drop" in c:
                g = g * c["drop"]
            g = g * (c["y"] > 0)  # ReLU
            n = g.shape[0] * g.shape[1] * g.shape[2]
            dbeta = g.sum(axis=(0, 1, 2))
            dgamma = (g * c["xhat"]).sum(axis=(0, 1, 2))
            dz = p["gamma"] * c["rstd"] * (g - dbeta / n - c["xhat"] * dgamma / n)
            grads[li]["gamma"], grads[li]["beta"] = dgamma, dbeta
        c["dz"] = dz  # kept for layer-wise checks of the device path
        gx, dk, db = _conv_backward(c["x"], p["kernel"], dz)
        grads[li]["kernel"], grads[li]["bias"] = dk, db
        if spec.src == "input":
            continue
        if spec.src in ("prev", "head"):
            add(li - 1, gx)
        elif spec.src == "pool":
            add(li - 1, _maxpool_backward(c["pool_in"], gx))
        elif spec.src == "up":
            B, H, W, C = gx.shape
            add(li - 1, gx.reshape(B, H // 2, 2, W // 2, 2, C).sum(axis=(2, 4)))
        elif spec.src == "concat":
            cu = spec.cin // 2
            add(li - 1, gx[..., :cu])
            add(spec.skip_from, gx[..., cu:])
    return loss, grads


def flatten_grads(grads) -> np.ndarray:
    return flatten_params(grads)


# ----------------------------------------------------------------------------
# optimizers (Keras formulations, Appendix B.8)
# ----------------------------------------------------------------------------

def adam_step(theta, g, m, v, t: int, lr=1e-3, b1=0.9, b2=0.999, eps=1e-7):
    """Keras Adam: theta -= lr*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+eps)."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    return theta - lr_t * m / (np.sqrt(v) + eps), m, v


def sgd_step(theta, g, mom_buf, lr=1e-2, momentum=0.0):
    """Keras SGD: v = momentum*v - lr*g; theta += v."""
    mom_buf = momentum * mom_buf - lr * g
    return theta + mom_buf, mom_buf


# ----------------------------------------------------------------------------
# synthetic workload -- shared by tests and bench
# ----------------------------------------------------------------------------

def synth_scans(n: int, H: int, W: int, num_classes: int, seed: int = 1234):
    """Seeded synthetic OCT-like B-scans: C-1 smooth sinusoidal boundaries ->
    area mask (create_area_mask semantics, dataset_construction.py:694-706),
    image = per-class grey level + Gaussian speckle.  Returns (images u8
    (n,H,W,1), labels u8 (n,H,W,1))."""
    rng = np.random.default_rng(seed)
    C = num_classes
    cols = np.arange(W)
    rows = np.arange(H)[:, None]
    images = np.zeros((n, H, W, 1), np.uint8)
    labels = np.zeros((n, H, W, 1), np.uint8)
    grey = np.linspace(40, 200, C)
    for i in range(n):
        base = np.sort(rng.uniform(0.15 * H, 0.85 * H, C - 1))
        bs = []
        for k in range(C - 1):
            A = rng.uniform(0.01 * H, 0.05 * H)
            lam = rng.uniform(W / 12.0, W / 3.0)
            ph = rng.uniform(0, 2 * np.pi)
            bs.append(base[k] + A * np.sin(cols / lam + ph))
        bs = np.clip(np.sort(np.stack(bs, 0), axis=0), 1, H - 2)
        lab = np.zeros((H, W), np.uint8)
        for k in range(C - 1):
            lab += (rows >= bs[k][None, :]).astype(np.uint8)
        img = grey[lab] + rng.normal(0, 25, (H, W))
        images[i, :, :, 0] = np.clip(img, 0, 255).astype(np.uint8)
        labels[i, :, :, 0] = lab
    return images, labels
