"""Keras-H5 checkpoint import/export -- CPU tests.

h5py is not importable in the build container, so the file layer is exercised through an in-memory stand-in that
offers the h5py subset the module touches (File / groups / attrs / datasets); when h5py IS importable the same round
trip is also run against real files.  PARITY UNPINNED against a file written by Keras itself (TensorFlow absent):
the layout asserted here is the published one (layer_names / weight_names attributes, <layer>/<layer>/<weight>:0).
"""
import json

import numpy as np
import pytest

from oracle import unet_numpy as on
from oct_image_segmentation_models_amd.common import keras_h5


class FakeNode:
    def __init__(self):
        self.attrs, self.children, self.data = {}, {}, None

    def create_group(self, name):
        self.children[name] = FakeNode(); return self.children[name]

    def create_dataset(self, name, data):
        n = FakeNode(); n.data = np.array(data); self.children[name] = n; return n

    def __contains__(self, k): return k in self.children
    def __getitem__(self, k):
        if k == (): return self.data
        node = self
        for part in k.split("/"):
            node = node.children[part]
        return node
    def keys(self): return self.children.keys()


class FakeH5:
    """Module-like backend: File(path, mode) -> context manager over a dict of in-memory trees."""
    def __init__(self): self.files = {}

    def File(self, path, mode):
        store = self
        class Ctx:
            def __enter__(self_inner):
                if mode == "w":
                    store.files[path] = FakeNode()
                return store.files[path]
            def __exit__(self_inner, *a): return False
        return Ctx()


CFG = dict(input_channels=1, num_classes=3, image_height=32, image_width=64, start_neurons=8, pool_layers=2, conv_layers=2,
           enc_kernel=(3, 3), dec_kernel=(2, 2))


def weights_for(cfg, seed=0):
    ocfg = on.UNetConfig(input_channels=cfg["input_channels"], num_classes=cfg["num_classes"], start_neurons=cfg["start_neurons"],
                         pool_layers=cfg["pool_layers"], conv_layers=cfg["conv_layers"])
    params, state = on.init_params(ocfg, seed=seed, dtype=np.float32, randomize_bn=True)
    return on.keras_weight_list(params, state)


def test_conv_plan_matches_the_oracle_plan():
    for cfg in (CFG, dict(CFG, pool_layers=4, start_neurons=4, conv_layers=1, input_channels=3, num_classes=5)):
        ocfg = on.UNetConfig(input_channels=cfg["input_channels"], num_classes=cfg["num_classes"], start_neurons=cfg["start_neurons"],
                             pool_layers=cfg["pool_layers"], conv_layers=cfg["conv_layers"])
        assert [(s.kh, s.kw, s.cin, s.cout, s.has_bn) for s in on.build_plan(ocfg)] == keras_h5.conv_plan(cfg)


def test_layout_names_and_round_trip_through_the_stand_in():
    w = weights_for(CFG)
    be = FakeH5()
    keras_h5.export_keras_h5("m.hdf5", w, CFG, h5=be)
    root = be.files["m.hdf5"]["model_weights"]
    names = [n.decode() for n in root.attrs["layer_names"]]
    # creation order of UNet.build_model: conv2d, batch_normalization, conv2d_1, batch_normalization_1, ..., head conv last
    assert names[:4] == ["conv2d", "batch_normalization", "conv2d_1", "batch_normalization_1"]
    n_conv = len(keras_h5.conv_plan(CFG))
    assert names[-1] == f"conv2d_{n_conv - 1}" and f"batch_normalization_{n_conv - 1}" not in names
    assert [x.decode() for x in root["conv2d_1"].attrs["weight_names"]] == ["conv2d_1/kernel:0", "conv2d_1/bias:0"]
    assert [x.decode() for x in root["batch_normalization"].attrs["weight_names"]] == [
        "batch_normalization/gamma:0", "batch_normalization/beta:0", "batch_normalization/moving_mean:0",
        "batch_normalization/moving_variance:0"]
    assert root["conv2d"]["conv2d"]["kernel:0"].data.shape == (3, 3, 1, 8)        # HWIO
    back = keras_h5.import_keras_h5("m.hdf5", CFG, h5=be)
    assert len(back) == len(w) and all(np.array_equal(a, b) for a, b in zip(back, w))
    assert keras_h5.read_embedded_config("m.hdf5", h5=be)["pool_layers"] == 2
    # weights-only nesting (model.save_weights) reads the same
    keras_h5.export_keras_h5("w.h5", w, CFG, h5=be, full_model=False)
    assert "model_weights" not in be.files["w.h5"]
    assert all(np.array_equal(a, b) for a, b in zip(keras_h5.import_keras_h5("w.h5", CFG, h5=be), w))


def test_import_orders_layers_by_suffix_and_skips_weightless_layers():
    """A Keras process that built other models first numbers layers from an offset; weightless layers (input,
    activation, pooling ...) appear in layer_names with empty weight_names."""
    w = weights_for(CFG, seed=3)
    layers = keras_h5.weights_to_layers(w, CFG)
    be = FakeH5()
    with be.File("k.hdf5", "w") as f:
        f.attrs["model_config"] = b"{}"
        root = f.create_group("model_weights")
        renamed = []
        for name, ws in layers:
            base, idx = ("conv2d", keras_h5._suffix(name, "conv2d")) if name.startswith("conv2d") else \
                        ("batch_normalization", keras_h5._suffix(name, "batch_normalization"))
            renamed.append((f"{base}_{idx + 23}", ws))
        order = list(reversed(renamed))                      # file order must not matter
        extra = ["input_3", "activation_23", "max_pooling2d_4", "dropout_1", "up_sampling2d_4", "concatenate_4"]
        root.attrs["layer_names"] = np.array([n.encode() for n in extra] + [n.encode() for n, _ in order], dtype="S")
        for n in extra:
            root.create_group(n).attrs["weight_names"] = np.array([], dtype="S")
        for n, ws in order:
            g = root.create_group(n); g.attrs["weight_names"] = np.array([f"{n}/{k}".encode() for k in ws], dtype="S")
            inner = g.create_group(n)
            for k, a in ws.items():
                inner.create_dataset(k, data=a)
    back = keras_h5.import_keras_h5("k.hdf5", CFG, h5=be)
    assert all(np.array_equal(a, b) for a, b in zip(back, w))
    assert keras_h5.read_embedded_config("k.hdf5", h5=be) is None      # a Keras-written file: config comes from model_config.json


def test_mismatched_architecture_and_missing_backend_fail_loudly():
    w = weights_for(CFG)
    be = FakeH5()
    keras_h5.export_keras_h5("m.hdf5", w, CFG, h5=be)
    with pytest.raises(keras_h5.KerasH5Error, match="needs"):
        keras_h5.import_keras_h5("m.hdf5", dict(CFG, pool_layers=3), h5=be)
    with pytest.raises(keras_h5.KerasH5Error, match="do not match"):
        keras_h5.import_keras_h5("m.hdf5", dict(CFG, start_neurons=4), h5=be)
    with pytest.raises(keras_h5.KerasH5Error, match="expected"):
        keras_h5.export_keras_h5("x.hdf5", w[:-1], CFG, h5=be)
    if not keras_h5.have_h5py():
        with pytest.raises(keras_h5.KerasH5Error, match="h5py"):
            keras_h5.import_keras_h5("m.hdf5", CFG)


@pytest.mark.skipif(not keras_h5.have_h5py(), reason="h5py not importable here")
def test_round_trip_through_real_hdf5_and_model_loader(tmp_path):
    from oct_image_segmentation_models_amd.models.engine_model import Model, load_model
    w = weights_for(CFG)
    p = keras_h5.export_keras_h5(tmp_path / "model_epoch01.hdf5", w, CFG)
    assert all(np.array_equal(a, b) for a, b in zip(keras_h5.import_keras_h5(p, CFG), w))
    (tmp_path / "model_config.json").write_text(json.dumps(CFG))
    m = load_model(p)
    assert all(np.array_equal(a, b) for a, b in zip(m.get_weights(), w))
