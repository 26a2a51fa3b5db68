"""Generates tests/golden/unet_golden.npz: a small fixed U-Net case (inputs, weights, dropout mask and the fp64
oracle's outputs) so that the oracle itself is pinned against accidental change and the HIP path is compared
against COMMITTED numbers, not only against a live oracle run.

The vectors come from oracle/unet_numpy.py (fp64).  They are NOT TensorFlow outputs: TensorFlow 2.9 is not
installable in the build container, so parity with the Keras reference itself stays unpinned.

    python tests/golden/make_unet_golden.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import unet_numpy as on  # noqa: E402
from tests.helpers import dropout_keep_mask, relu_margin  # noqa: E402

CFG = dict(input_channels=1, num_classes=3, start_neurons=4, pool_layers=2, conv_layers=2)
B, H, W = 2, 16, 32


def main():
    cfg = on.UNetConfig(**CFG)
    params, state = on.init_params(cfg, seed=11, dtype=np.float32, randomize_bn=True)
    p64 = [{k: v.astype(np.float64) for k, v in p.items()} for p in params]
    s64 = [{k: v.astype(np.float64) for k, v in s.items()} for s in state]
    mask = dropout_keep_mask(4242, 7, (B, H >> 2, W >> 2, 16)).astype(np.float64)
    for seed in range(1, 500):       # a data seed with a safe ReLU margin (tight fp32-vs-fp64 comparison is then valid)
        images, labels = on.synth_scans(B, H, W, 3, seed=seed)
        x = on.preprocess_u8(images, np.float64)
        probs_t, cache = on.forward(cfg, p64, s64, x, training=True, dropout_mask=mask)
        if relu_margin(cfg, p64, cache) > 2e-5:
            break
    probs_i, _ = on.forward(cfg, p64, s64, x, training=False)
    out = {"images": images, "labels": labels, "params_flat": on.flatten_params(params), "state_flat": on.flatten_state(state),
           "dropout_seed": np.array(4242), "dropout_step": np.array(7), "probs_infer": probs_i, "probs_train": probs_t,
           "data_seed": np.array(seed)}
    y = on.one_hot(labels, 3, np.float64)
    out["metrics_train"] = np.array([on.dice_loss_macro(y, probs_t), on.dice_loss_micro(y, probs_t),
                                     on.dice_coef_macro(y, probs_t), on.dice_coef_micro(y, probs_t)])
    for macro in (True, False):
        loss, grads = on.backward(cfg, p64, cache, labels, macro=macro, loss_scale=1.0)
        out["grads_macro" if macro else "grads_micro"] = on.flatten_grads(grads)
    out["state_after"] = on.flatten_state(on.updated_moving_stats(cfg, s64, cache))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "unet_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "data seed", seed, "margin", relu_margin(cfg, p64, cache))


if __name__ == "__main__":
    main()
