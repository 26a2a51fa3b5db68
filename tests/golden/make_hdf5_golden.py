#!/opt/conda/bin/python3.9
"""Generates the REAL-HDF5 fixtures of tests/test_hdf5_real.py with the real ``h5py`` (3.3.0 under
/opt/conda/bin/python3.9 in the build image; the system interpreter has no h5py):

  tests/golden/dataset_small.hdf5        a dataset file with the reference's key contract
                                         (reference common/dataset_loader.py:9-33)
  tests/golden/keras_weights_small.hdf5  a Keras-2.x-layout full-model weight file for UNet(start_neurons=4,
                                         pool_layers=1, conv_layers=1), written here with plain h5py calls that follow the
                                         published layout (layer_names / weight_names attributes, <layer>/<layer>/<w>:0),
                                         INCLUDING what a real Keras process adds: weightless layers with empty
                                         weight_names, auto-numbered layer names that do not start at 0, a model_config
                                         JSON attribute stored as a variable-length string
  tests/golden/hdf5_expected.npz         the arrays both files hold (what the readers must return)

    /opt/conda/bin/python3.9 tests/golden/make_hdf5_golden.py
TensorFlow itself is not installable here, so the Keras file is still "layout as documented", not a Keras-written file."""
import json
import os

import h5py
import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(20240)
exp = {}

# ---- dataset ----
H, W = 16, 32
def scans(n):
    img = rng.integers(0, 256, (n, H, W, 1)).astype(np.uint8)
    lab = (np.arange(H)[None, :, None, None] // 6).astype(np.uint8) * np.ones((n, 1, W, 1), np.uint8)
    return img, lab
with h5py.File(os.path.join(here, "dataset_small.hdf5"), "w") as f:
    for split, n in (("train", 6), ("val", 2), ("test", 3)):
        img, lab = scans(n)
        f.create_dataset(f"{split}_images", data=img); f.create_dataset(f"{split}_labels", data=lab)
        exp[f"{split}_images"], exp[f"{split}_labels"] = img, lab
    src = np.array([f"volume_{i}.tiff".encode() for i in range(3)])
    f.create_dataset("test_images_source", data=src)
    exp["test_images_source"] = src
    f.attrs["description"] = "synthetic 16x32 scans, 3 classes"      # variable-length UTF-8, as h5py stores a str
    f.attrs["n_classes"] = 3

# ---- Keras-layout weights: UNet(input_channels=1, num_classes=3, start_neurons=4, pool_layers=1, conv_layers=1) ----
cfg = {"input_channels": 1, "num_classes": 3, "image_height": H, "image_width": W, "start_neurons": 4, "pool_layers": 1,
       "conv_layers": 1, "enc_kernel": [3, 3], "dec_kernel": [2, 2]}
convs = [(3, 3, 1, 4, True), (3, 3, 4, 8, True), (2, 2, 8, 4, True), (3, 3, 8, 4, True), (1, 1, 4, 3, False)]
OFF_C, OFF_B = 23, 22          # this "process" had built another model before: names start at conv2d_23 / batch_normalization_22
layer_names, groups, wi = ["input_2"], {}, 0
for ci, (kh, kw, cin, cout, bn) in enumerate(convs):
    cname = f"conv2d_{OFF_C + ci}"
    k = rng.standard_normal((kh, kw, cin, cout)).astype(np.float32); b = rng.standard_normal(cout).astype(np.float32)
    groups[cname] = {"kernel:0": k, "bias:0": b}; layer_names.append(cname)
    exp[f"w{wi:03d}"] = k; exp[f"w{wi + 1:03d}"] = b; wi += 2
    if bn:
        bname = f"batch_normalization_{OFF_B + ci}"
        arrs = {"gamma:0": rng.uniform(0.5, 1.5, cout).astype(np.float32), "beta:0": rng.standard_normal(cout).astype(np.float32),
                "moving_mean:0": rng.standard_normal(cout).astype(np.float32),
                "moving_variance:0": rng.uniform(0.5, 1.5, cout).astype(np.float32)}
        groups[bname] = arrs; layer_names.append(bname)
        for a in arrs.values():
            exp[f"w{wi:03d}"] = a; wi += 1
        layer_names.append(f"activation_{OFF_B + ci}"); groups[layer_names[-1]] = {}
    if ci == 0:
        layer_names.append("max_pooling2d_5"); groups["max_pooling2d_5"] = {}
    if ci == 1:
        layer_names.append("up_sampling2d_5"); groups["up_sampling2d_5"] = {}
    if ci == 2:
        layer_names.append("concatenate_5"); groups["concatenate_5"] = {}
groups["input_2"] = {}
with h5py.File(os.path.join(here, "keras_weights_small.hdf5"), "w") as f:
    f.attrs["keras_version"] = "2.9.0"; f.attrs["backend"] = "tensorflow"
    f.attrs["model_config"] = json.dumps({"class_name": "Functional", "config": {"name": "model_1"}})
    mw = f.create_group("model_weights")
    mw.attrs["layer_names"] = np.array([n.encode() for n in layer_names])
    mw.attrs["backend"] = "tensorflow"; mw.attrs["keras_version"] = "2.9.0"
    for name in layer_names:
        g = mw.create_group(name)
        ws = groups[name]
        g.attrs["weight_names"] = np.array([f"{name}/{w}".encode() for w in ws]) if ws else np.zeros((0,), "S1")
        if ws:
            inner = g.create_group(name)
            for w, a in ws.items():
                inner.create_dataset(w, data=a)
exp["config_json"] = np.array(json.dumps(cfg))
np.savez(os.path.join(here, "hdf5_expected.npz"), **exp)
print("wrote", sorted(os.listdir(here)))
