"""Independent torch-CPU restatement of the U-Net hot path -- TEST INFRASTRUCTURE ONLY.

Second, independently written implementation of the same semantics as
``oracle/unet_numpy.py`` (functional torch ops + autograd instead of
hand-derived numpy gradients).  Used (a) in ``tests/`` to cross-check the numpy
oracle, and (b) by ``bench.py``'s ``cpu_baseline`` leg as the multi-threaded
(oneDNN) CPU port of the reference path, because the Keras/TensorFlow-2.9
reference itself is not installable here.  PARITY UNPINNED
against TensorFlow, see the header of ``oracle/unet_numpy.py``.

Follows /root/reference/oct_image_segmentation_models/models/unet.py:20-57,
106-153 and common/custom_losses.py:47-81.  Never imported by the product
package.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn.functional as F

from .unet_numpy import UNetConfig, build_plan, same_pad


def _conv_same(x, kernel_hwio, bias):
    # x: NCHW; kernel HWIO -> OIHW; explicit asymmetric TF "same" pad (2x2: 0 before, 1 after)
    kh, kw = kernel_hwio.shape[0], kernel_hwio.shape[1]
    (pt, pb), (pl, pr) = same_pad(kh), same_pad(kw)
    x = F.pad(x, (pl, pr, pt, pb))
    return F.conv2d(x, kernel_hwio.permute(3, 2, 0, 1).contiguous(), bias)


def forward(cfg: UNetConfig, params: List[dict], state: List[dict], x_nhwc: torch.Tensor,
            training: bool = False, dropout_mask: Optional[torch.Tensor] = None,
            collect_stats: Optional[list] = None, collect_z: Optional[list] = None):
    """params/state: lists of dicts of torch tensors (same structure as the numpy
    oracle).  Returns probabilities NHWC.  ``collect_z``: every conv block's pre-BN
    output z (NCHW) is appended, with its gradient retained -- after ``backward()``
    ``z.grad`` is that block's dz."""
    plan = build_plan(cfg)
    x = x_nhwc.permute(0, 3, 1, 2)
    outs = {}
    bi = 0
    for li, spec in enumerate(plan):
        p = params[li]
        if spec.src == "pool":
            x = F.max_pool2d(x, 2)
        elif spec.src == "up":
            x = F.interpolate(x, scale_factor=2, mode="nearest")
        elif spec.src == "concat":
            x = torch.cat([x, outs[spec.skip_from]], dim=1)
        z = _conv_same(x, p["kernel"], p["bias"])
        if collect_z is not None:
            if z.requires_grad:
                z.retain_grad()
            collect_z.append(z)
        if spec.has_bn:
            st = state[bi]; bi += 1
            if training:
                mean = z.mean(dim=(0, 2, 3))
                var = z.var(dim=(0, 2, 3), unbiased=False)
                if collect_stats is not None:
                    collect_stats.append((mean.detach(), var.detach(), z.numel() // z.shape[1]))
            else:
                mean, var = st["moving_mean"], st["moving_var"]
            zn = (z - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + cfg.bn_eps)
            x = F.relu(zn * p["gamma"][None, :, None, None] + p["beta"][None, :, None, None])
            outs[li] = x
            if training and spec.name == f"mid.conv{cfg.conv_layers - 1}" and cfg.dropout_rate > 0:
                x = x * dropout_mask.permute(0, 3, 1, 2) / (1.0 - cfg.dropout_rate)
        else:
            x = torch.softmax(z, dim=1)
    return x.permute(0, 2, 3, 1)


def dice_loss(y_onehot, p, macro: bool, smooth: float = 1e-5):
    if macro:
        I = (y_onehot * p).sum(dim=(1, 2)); D = y_onehot.sum(dim=(1, 2)) + p.sum(dim=(1, 2))
        return 1.0 - ((2.0 * I + smooth) / (D + smooth)).mean()
    I = (y_onehot * p).sum(); D = y_onehot.sum() + p.sum()
    return 1.0 - (2.0 * I + smooth) / (D + smooth)


def to_torch(params_np, state_np, dtype=torch.float64, requires_grad=False):
    params = [{k: torch.tensor(v, dtype=dtype, requires_grad=requires_grad) for k, v in p.items()}
              for p in params_np]
    state = [{k: torch.tensor(v, dtype=dtype) for k, v in s.items()} for s in state_np]
    return params, state


def loss_and_grads(cfg, params_np, state_np, x_np, labels_np, macro=True, dropout_mask_np=None,
                   dtype=torch.float64, loss_scale=1.0):
    """Autograd gradients in the numpy oracle's structure."""
    params, state = to_torch(params_np, state_np, dtype, requires_grad=True)
    x = torch.tensor(x_np, dtype=dtype)
    dm = None if dropout_mask_np is None else torch.tensor(dropout_mask_np, dtype=dtype)
    probs = forward(cfg, params, state, x, training=True, dropout_mask=dm)
    lab = torch.tensor(labels_np.reshape(labels_np.shape[:3]).astype("int64"))
    y = F.one_hot(lab, cfg.num_classes).to(dtype)
    loss = dice_loss(y, probs, macro)
    (loss * loss_scale).backward()
    grads = [{k: v.grad.numpy() for k, v in p.items()} for p in params]
    return float(loss.detach()), probs.detach().numpy(), grads
