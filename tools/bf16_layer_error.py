"""Per-layer error of the bf16-storage mode against the fp64 oracle (diagnostic; run on the GPU box).

Prints, for every stored pre-BN tensor z, max|err|/max|z| and rms(err)/rms(z), so that smooth rounding accumulation
can be told from a layer that is simply wrong.  Usage: python tools/bf16_layer_error.py
"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import unet_numpy as on
import test_gpu_parity as T

for case in T.BF16_CASES:
    B, H, W, C, sn, P, L, ic = case
    for mode in ("bf16", "f32"):
        if mode == "bf16":
            cfg, eng, p64, s64 = T.make_bf16(B, H, W, C, sn, P, L, ic)
        else:
            cfg, eng, params, state = T.make(B, H, W, C, sn, P, L, ic)
            p64 = [{k: v.astype(np.float64) for k, v in p.items()} for p in params]
            s64 = [{k: v.astype(np.float64) for k, v in s.items()} for s in state]
        images, labels = T.data(B, H, W, C, ic, seed=T.MARGIN_SEED.get(case, 5))
        x = torch.from_numpy(images).cuda(); lab = torch.from_numpy(labels[..., 0].copy()).cuda()
        xin = on.preprocess_u8(images, np.float64)
        eng.set_dropout_step(T.DROP_STEP)
        mask = eng.dropout_mask(B).cpu().numpy().astype(np.float64)
        probs, _ = eng.forward(x, training=True, labels=lab)
        ref, cache = on.forward(cfg, p64, s64, xin, training=True, dropout_mask=mask)
        print(f"case {case} mode {mode}: probs max err {np.abs(probs.cpu().numpy() - ref).max():.3e}")
        for li, spec in enumerate(on.build_plan(cfg)[:-1]):
            z = eng.debug_activation(li, 0)[:B].cpu().numpy().astype(np.float64)
            r = cache[li]["z"]; e = z - r
            # error of merely rounding the oracle's own tensor to bf16 (the floor for a stored bf16 tensor)
            rq = torch.from_numpy(r).to(torch.bfloat16).double().numpy() - r
            print(f"  {li:2d} {spec.name:14s} max/max {np.abs(e).max() / np.abs(r).max():.3e}  rms/rms "
                  f"{np.sqrt((e * e).mean() / (r * r).mean()):.3e}   [store-only floor: max/max "
                  f"{np.abs(rq).max() / np.abs(r).max():.3e} rms/rms {np.sqrt((rq * rq).mean() / (r * r).mean()):.3e}]")
