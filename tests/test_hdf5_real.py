"""REAL HDF5 files: the reference's on-disk contract -- dataset keys
(common/dataset_loader.py:9-33), ``training_params.hdf5`` attributes (training/training.py:39-132) and Keras
``model_epochNN.hdf5`` checkpoints (training/training.py:319-326, common/utils.py:63-69) -- exercised on actual HDF5
files, in both directions against the real ``h5py``:

* fixtures under tests/golden/ were WRITTEN by h5py 3.3.0 (tests/golden/make_hdf5_golden.py, run with the image's
  /opt/conda/bin/python3.9) and are READ here by the package's HDF5 backend (``h5lite`` = libhdf5 through ctypes, or
  h5py when the running interpreter has it);
* files WRITTEN here by that backend are handed to the real h5py in a child ``python3.9`` process and checked there.

Still PARITY UNPINNED against a file written by Keras itself: TensorFlow cannot be installed here; the Keras fixture
follows the published layout including what a live Keras process adds (weightless layers, name offsets, vlen strings)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oct_image_segmentation_models_amd.common import dataset_loader, h5io, keras_h5

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CONDA_PY = "/opt/conda/bin/python3.9"
HDF5_MAGIC = b"\x89HDF\r\n\x1a\n"

pytestmark = pytest.mark.skipif(not h5io.HAVE_H5PY, reason="neither h5py nor libhdf5 available")


def expected():
    with np.load(os.path.join(GOLD, "hdf5_expected.npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def test_backend_is_real_hdf5():
    assert h5io.BACKEND in ("h5py", "h5lite")
    assert open(os.path.join(GOLD, "dataset_small.hdf5"), "rb").read(8) == HDF5_MAGIC


def test_dataset_written_by_h5py_is_read_through_the_reference_key_contract():
    exp = expected()
    data = dataset_loader.open_dataset(os.path.join(GOLD, "dataset_small.hdf5"))
    tr_i, tr_l = dataset_loader.load_training_data(data)
    va_i, va_l = dataset_loader.load_validation_data(data)
    te_i, te_l, names = dataset_loader.load_testing_data(data)
    for got, key in ((tr_i, "train_images"), (tr_l, "train_labels"), (va_i, "val_images"), (va_l, "val_labels"),
                     (te_i, "test_images"), (te_l, "test_labels")):
        assert got.dtype == np.uint8 and np.array_equal(got, exp[key]), key
    assert [str(n) for n in names] == ["volume_0.tiff", "volume_1.tiff", "volume_2.tiff"]
    assert int(data["attr:n_classes"]) == 3
    assert bytes(data["attr:description"]).decode() == "synthetic 16x32 scans, 3 classes"     # variable-length string
    assert len(np.unique(tr_l)) == 3                                                           # training.py:176


def test_keras_layout_file_written_by_h5py_imports_into_the_engine_weight_list():
    exp = expected()
    cfg = json.loads(str(exp["config_json"]))
    w = keras_h5.import_keras_h5(os.path.join(GOLD, "keras_weights_small.hdf5"), cfg)
    want = [exp[k] for k in sorted(k for k in exp if k.startswith("w") and k[1:].isdigit())]
    assert len(w) == len(want) == 4 * 6 + 2
    for a, b in zip(w, want):
        assert a.dtype == np.float32 and np.array_equal(a, b)
    assert keras_h5.read_embedded_config(os.path.join(GOLD, "keras_weights_small.hdf5")) is None   # Keras' own files carry none
    with pytest.raises(keras_h5.KerasH5Error):
        keras_h5.import_keras_h5(os.path.join(GOLD, "keras_weights_small.hdf5"), dict(cfg, start_neurons=8))


def test_h5io_round_trip_is_a_real_hdf5_file(tmp_path):
    rng = np.random.default_rng(0)
    ds = {"train_loss": rng.random(7), "val_acc": rng.random(7).astype(np.float32), "labels": rng.integers(0, 4, (2, 5, 6, 1)).astype(np.uint8),
          "names": np.array([b"a.tiff", b"longer_name.tiff"])}
    attrs = {"epochs": 150, "batch_size": 4, "loss_name": np.array("dice_loss_macro", dtype="S1000"), "shuffle": True,
             "opt_param: learning_rate": 0.004, "timestamp": np.array("2026-10-04", dtype="S100")}
    p = h5io.save(tmp_path / "training_params.hdf5", ds, attrs)
    assert p == tmp_path / "training_params.hdf5" and open(p, "rb").read(8) == HDF5_MAGIC and not (tmp_path / "training_params.hdf5.npz").exists()
    back = h5io.load(p)
    for k, v in ds.items():
        assert np.array_equal(back[k], v) and back[k].dtype == np.asarray(v).dtype, k
    assert int(back["attr:epochs"]) == 150 and float(back["attr:opt_param: learning_rate"]) == 0.004
    assert bytes(back["attr:loss_name"]).rstrip(b"\x00") == b"dice_loss_macro" and int(back["attr:shuffle"]) == 1
    assert h5io.exists(p)
    h5io.remove(p)
    assert not h5io.exists(p)


CHILD = r"""
import json, sys
import h5py, numpy as np
path, npz = sys.argv[1], sys.argv[2]
exp = np.load(npz, allow_pickle=False)
with h5py.File(path, "r") as f:
    assert "model_weights" in f and f.attrs["keras_version"] == b"2.9.0", dict(f.attrs)
    mw = f["model_weights"]
    names = [n.decode() for n in mw.attrs["layer_names"]]
    assert names[0] == "conv2d" and names[1] == "batch_normalization" and names[-1].startswith("conv2d_"), names
    i = 0
    for n in names:
        for wn in mw[n].attrs["weight_names"]:
            wn = wn.decode()
            assert wn.startswith(n + "/") and wn.endswith(":0"), wn
            got = mw[n][wn][()]
            want = exp["w%03d" % i]; i += 1
            assert got.dtype == np.float32 and got.shape == want.shape and np.array_equal(got, want), wn
    assert i == len([k for k in exp.files if k.startswith("w")]), i
ds = sys.argv[3]
with h5py.File(ds, "r") as f:
    assert f["train_images"].dtype == np.uint8 and f["train_images"].shape == (6, 16, 32, 1)
    assert f["test_images_source"][0] == b"volume_0.tiff" and int(f.attrs["n"]) == 7 and f.attrs["tag"] == b"x"
print("child-ok")
"""


@pytest.mark.skipif(not os.path.exists(CONDA_PY), reason="no interpreter with the real h5py on this machine")
def test_files_written_here_are_read_by_the_real_h5py(tmp_path):
    """Writer direction: export with this package's backend, verify layout and values with h5py in a child process."""
    if h5io.BACKEND == "h5py":
        pytest.skip("running under h5py itself")
    exp = expected()
    cfg = json.loads(str(exp["config_json"]))
    w = [exp[k] for k in sorted(k for k in exp if k.startswith("w") and k[1:].isdigit())]
    kp = keras_h5.export_keras_h5(tmp_path / "model_epoch01.hdf5", w, cfg)
    dp = h5io.save(tmp_path / "data.hdf5", {k: exp[k] for k in exp if k.split("_")[0] in ("train", "val", "test")},
                   {"n": 7, "tag": b"x"})
    r = subprocess.run([CONDA_PY, "-c", CHILD, str(kp), os.path.join(GOLD, "hdf5_expected.npz"), str(dp)],
                       capture_output=True, text=True, timeout=120, env={"PATH": os.environ.get("PATH", "")})
    assert r.returncode == 0 and "child-ok" in r.stdout, r.stderr[-2000:]
    # and the same file comes back through this package's reader + Model loader (model_config embedded by export)
    from oct_image_segmentation_models_amd.models.engine_model import load_model
    m = load_model(kp)
    assert all(np.array_equal(a, b) for a, b in zip(m.get_weights(), w))


def test_h5lite_reads_shape_dtype_and_leading_axis_ranges_without_reading_the_dataset():
    """``Dataset.shape`` / ``.dtype`` / ``len`` come from the file's metadata and a contiguous range of the leading axis is
    read as an HDF5 hyperslab (what the batch generators do on multi-GB ``train_images``): checked on the file the REAL
    h5py wrote, against its expected arrays."""
    from oct_image_segmentation_models_amd.common import h5lite
    if not h5lite.available():
        pytest.skip("libhdf5 not loadable")
    exp = expected()
    with h5lite.File(os.path.join(GOLD, "dataset_small.hdf5"), "r") as f:
        d = f["train_images"]; e = exp["train_images"]
        reads = []
        orig = h5lite._read

        def spy(*a, **k):
            out = orig(*a, **k); reads.append(np.shape(out)); return out
        h5lite._read = spy
        try:
            assert d.shape == e.shape and d.dtype == e.dtype and len(d) == e.shape[0]
            assert reads == []                                            # metadata only
            assert np.array_equal(d[1:3], e[1:3]) and reads[-1] == e[1:3].shape      # only the two scans left the file
            assert np.array_equal(d[-1], e[-1]) and reads[-1] == e[-1:].shape
            assert np.array_equal(d[0:2, ...], e[0:2]) and d[2:2].shape == (0,) + e.shape[1:]
            assert np.array_equal(d[::2], e[::2]) and np.array_equal(d[1:, 3], e[1:, 3]) and np.array_equal(d[()], e)
            with pytest.raises(IndexError):
                d[e.shape[0]]
        finally:
            h5lite._read = orig
        names = f["test_images_source"]
        assert names.dtype == np.dtype("O") or names.dtype.kind == "S"
