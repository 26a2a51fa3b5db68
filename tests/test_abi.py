"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol include/oct_unet.h
declares, and its host-side plan agrees with the oracle's restatement of models/unet.py.  No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import __graft_entry__ as ge
from oracle import unet_numpy as on

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hip():
    ge.build()
    from oct_image_segmentation_models_amd import _hip
    return _hip


def test_library_exports_every_declared_symbol(hip):
    header = open(os.path.join(ROOT, "include", "oct_unet.h")).read()
    declared = set(re.findall(r"\b(oct_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 20
    lib = C.CDLL(hip.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in oct_unet.h but not exported"
    assert declared == {n for n, _, _ in hip.SYMBOLS}, "ctypes binding and header disagree"
    assert b"gfx950" in hip.lib().oct_version()


@pytest.mark.parametrize("kw", [dict(), dict(num_classes=4), dict(pool_layers=5, image_height=512, image_width=1024),
                                dict(start_neurons=4, pool_layers=2, conv_layers=3, input_channels=3)])
def test_plan_matches_oracle(hip, kw):
    from oct_image_segmentation_models_amd.engine import make_cfg, layer_table
    base = dict(input_channels=1, num_classes=3, image_height=256, image_width=512)
    base.update(kw)
    cfg = make_cfg(**base)
    ocfg = on.UNetConfig(input_channels=base["input_channels"], num_classes=base["num_classes"],
                         start_neurons=base.get("start_neurons", 8), pool_layers=base.get("pool_layers", 4),
                         conv_layers=base.get("conv_layers", 2))
    t, s = on.param_count(ocfg)
    lib = hip.lib()
    assert lib.oct_unet_param_count(C.byref(cfg)) == t and lib.oct_unet_state_count(C.byref(cfg)) == s
    plan = on.build_plan(ocfg)
    layers = layer_table(cfg)
    assert [l["name"] for l in layers] == [p.name for p in plan]
    off = soff = 0
    for l, p in zip(layers, plan):
        assert (l["kh"], l["kw"], l["cin"], l["cout"], l["has_bn"]) == (p.kh, p.kw, p.cin, p.cout, int(p.has_bn))
        assert (l["out_h"], l["out_w"]) == (base["image_height"] >> p.level, base["image_width"] >> p.level)
        assert l["kernel_off"] == off; off += p.kh * p.kw * p.cin * p.cout
        assert l["bias_off"] == off; off += p.cout
        if p.has_bn:
            assert l["gamma_off"] == off and l["beta_off"] == off + p.cout; off += 2 * p.cout
            assert l["moving_mean_off"] == soff and l["moving_var_off"] == soff + p.cout; soff += 2 * p.cout


def test_default_counts_and_workspace(hip):
    from oct_image_segmentation_models_amd.engine import make_cfg
    lib = hip.lib()
    cfg = make_cfg(input_channels=1, num_classes=3, image_height=256, image_width=512, max_batch=32, training=True)
    assert lib.oct_unet_param_count(C.byref(cfg)) == 487403 and lib.oct_unet_state_count(C.byref(cfg)) == 1712
    ws_train = lib.oct_unet_workspace_bytes(C.byref(cfg))
    cfg.training = 0
    ws_inf = lib.oct_unet_workspace_bytes(C.byref(cfg))
    # conv outputs of the 22 BN blocks (39.9 MB/scan f32, from the layer table of SURVEY A.1) + pooled
    # tensors (2.0 MB) + scratch; training adds gradient buffers of the same size + dlogits + dW partials
    assert 32 * 41e6 < ws_inf < 32 * 44e6 and 1.9 * ws_inf < ws_train < 2.3 * ws_inf


def test_bad_configs_rejected(hip):
    from oct_image_segmentation_models_amd.engine import make_cfg
    from oct_image_segmentation_models_amd._hip import OctError
    for kw in (dict(image_height=250), dict(num_classes=9), dict(start_neurons=6), dict(enc_kernel=(5, 5)),
               dict(pool_layers=0)):
        base = dict(input_channels=1, num_classes=3, image_height=256, image_width=512)
        base.update(kw)
        with pytest.raises(OctError):
            make_cfg(**base)


def test_engine_fails_loudly_without_gpu(hip):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from oct_image_segmentation_models_amd.engine import UNetEngine
    from oct_image_segmentation_models_amd._hip import OctError
    with pytest.raises(OctError, match="no CPU fallback"):
        UNetEngine(input_channels=1, num_classes=3, image_height=32, image_width=64)


def test_glorot_init_statistics(hip):
    from oct_image_segmentation_models_amd.engine import make_cfg, glorot_init, layer_table
    cfg = make_cfg(input_channels=1, num_classes=3, image_height=256, image_width=512)
    p, s = glorot_init(cfg, 0)
    L = layer_table(cfg)[9]  # mid.conv1 128->128
    n = 9 * 128 * 128
    k = p[L["kernel_off"]:L["kernel_off"] + n]
    lim = np.sqrt(6.0 / (9 * 256))
    assert abs(k.max() - lim) < 1e-3 * lim + 1e-4 and abs(k.std() - lim / np.sqrt(3)) < 0.01 * lim
    assert np.all(p[L["gamma_off"]:L["gamma_off"] + 128] == 1) and np.all(p[L["bias_off"]:L["bias_off"] + 128] == 0)
    assert np.all(s[L["moving_var_off"]:L["moving_var_off"] + 128] == 1)
