#!/usr/bin/env python3
"""Search (CPU, oracle only) for a data seed whose training forward keeps every BN pre-activation > margin away
from the ReLU kink, for a tests/test_gpu_parity.py CASES entry.  usage: find_margin_seed.py B H W C sn P L in_ch"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import unet_numpy as on
from tests.helpers import dropout_keep_mask, relu_margin

B, H, W, C, sn, P, L, ic = map(int, sys.argv[1:9])
want = float(sys.argv[9]) if len(sys.argv) > 9 else 2.5e-5
cfg = on.UNetConfig(input_channels=ic, num_classes=C, start_neurons=sn, pool_layers=P, conv_layers=L)
params, state = on.init_params(cfg, seed=0, dtype=np.float32, randomize_bn=True)
p64 = [{k: v.astype(np.float64) for k, v in p.items()} for p in params]
s64 = [{k: v.astype(np.float64) for k, v in s.items()} for s in state]
mask = dropout_keep_mask(100, 3, (B, H >> P, W >> P, sn << P)).astype(np.float64)
best = (0, -1)
for seed in range(1, 400):
    images, labels = on.synth_scans(B, H, W, C, seed=seed)
    if ic > 1:
        images = np.random.default_rng(seed).integers(0, 256, (B, H, W, ic)).astype(np.uint8)
    _, cache = on.forward(cfg, p64, s64, on.preprocess_u8(images, np.float64), training=True, dropout_mask=mask)
    m = relu_margin(cfg, p64, cache)
    if m > best[0]:
        best = (m, seed); print(seed, m, flush=True)
    if m > want:
        break
