"""Dev tool (GPU box): layer-wise error report of the HIP path vs the oracle for one training step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import unet_numpy as on
from tests.test_gpu_parity import make, data

def report(case, macro=True):
    B, H, W, C, sn, P, L, ic = case
    cfg, eng, p64, s64 = make(B, H, W, C, sn, P, L, ic, training=True)
    images, labels = data(B, H, W, C, ic)
    x = torch.from_numpy(images).cuda(); lab = torch.from_numpy(labels[..., 0].copy()).cuda()
    eng.set_dropout_step(3)
    mask = eng.dropout_mask(B).cpu().numpy().astype(np.float64)
    probs, _ = eng.forward(x, training=True, labels=lab)
    loss4 = eng.loss_dice().cpu().numpy()
    eng.backward(lab, macro=macro, loss_scale=0.5)
    torch.cuda.synchronize()
    ref, cache = on.forward(cfg, p64, s64, on.preprocess_u8(images, np.float64), training=True, dropout_mask=mask)
    loss, grads = on.backward(cfg, p64, cache, labels, macro=macro, loss_scale=0.5)
    plan = on.build_plan(cfg)
    print("case", case, "loss", loss4, "ref", loss)
    for li in range(len(plan) - 2, -1, -1):
        dz = eng.debug_dz(li)[:B].cpu().numpy(); r = cache[li]["dz"]
        e = np.abs(dz - r); sc = np.abs(r).max()
        idx = np.unravel_index(e.argmax(), e.shape)
        y = cache[li]["y"]
        yb = p64[li]["gamma"] * cache[li]["xhat"] + p64[li]["beta"]
        if e.max() / sc > 1e-3:
            bad = np.argwhere(e > 1e-3 * sc)
            print("     pre-activation (gamma*xhat+beta) at outliers:", [f"{yb[tuple(i)]:.2e}" for i in bad[:6]],
                  " rel L2 err", f"{np.linalg.norm(dz - r) / np.linalg.norm(r):.2e}", " frac>1e-3:", f"{(e > 1e-3 * sc).mean():.2e}",
                  " min|yb|", f"{np.abs(yb).min():.1e}")
        print(f"L{li:2d} {plan[li].name:12s} dz max|r|={sc:.3e} maxerr/sc={e.max()/sc:.2e} med={np.median(e)/sc:.1e} "
              f"n(>1e-3)={int((e > 1e-3*sc).sum())} at {idx} y_ref={y[idx]:.3e} z_ref={cache[li]['z'][idx]:.4f} dz={dz[idx]:.4e} ref={r[idx]:.4e}")
    g = eng.grads.cpu().numpy()
    for L_, gr in zip(eng.layers, grads):
        n = L_["kh"] * L_["kw"] * L_["cin"] * L_["cout"]; c = L_["cout"]
        pieces = [("kernel", L_["kernel_off"], n), ("bias", L_["bias_off"], c)]
        if L_["has_bn"]: pieces += [("gamma", L_["gamma_off"], c), ("beta", L_["beta_off"], c)]
        out = []
        for key, off, cnt in pieces:
            refv = gr[key].ravel(); sc = max(np.abs(refv).max(), 1e-30)
            out.append(f"{key} {np.abs(g[off:off+cnt]-refv).max()/sc:.1e}(|r|={sc:.1e})")
        print(f"  grad {L_['name']:12s} " + " ".join(out))

if __name__ == "__main__":
    import sys
    if len(sys.argv) > 1 and sys.argv[1] == "full":
        report((2, 256, 512, 3, 8, 4, 2, 1))
        sys.exit(0)
    report((2, 32, 64, 3, 8, 2, 2, 1))
