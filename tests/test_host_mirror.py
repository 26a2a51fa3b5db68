"""CPU tests of the host-side mirror of the reference API (no GPU, no compute through the engine)."""
import json
import os

import numpy as np
import pytest

import __graft_entry__ as ge
from oracle import unet_numpy as on


@pytest.fixture(scope="module", autouse=True)
def built():
    ge.build()


def test_registry_and_unet_container():
    from oct_image_segmentation_models_amd.models import get_model_class, model_name_map
    assert set(model_name_map) == {"unet"}
    with pytest.raises(ValueError, match="could not be found"):
        get_model_class("deeplabv3plus")
    with pytest.raises(TypeError):
        get_model_class(3)
    UNet = get_model_class("unet")
    with pytest.raises(TypeError):
        UNet(1, 3, 256, 512)   # keyword-only, as the reference
    m = UNet(input_channels=1, num_classes=4, image_height=256, image_width=512, pool_layers=3, enc_kernel=[3, 3])
    cfg = m.get_config()
    assert cfg == {"input_channels": 1, "num_classes": 4, "image_height": 256, "image_width": 512, "start_neurons": 8,
                   "pool_layers": 3, "conv_layers": 2, "enc_kernel": (3, 3), "dec_kernel": (2, 2)}
    m2 = UNet(**json.loads(json.dumps(cfg)))       # round trip through model_config.json (tuples become lists)
    assert m2.get_config() == cfg
    f = m.get_preprocess_input_fn()
    assert np.allclose(f(np.array([0, 51, 255], np.uint8)), [0, 0.2, 1.0])
    model = m.build_model()
    assert model.name == "unet" and model.output.shape[-1] == 4
    assert model.count_params() == sum(on.param_count(on.UNetConfig(num_classes=4, pool_layers=3)))
    lines = []
    model.summary(print_fn=lines.append)
    assert any("enc0.conv0" in l for l in lines) and any("head" in l for l in lines)


def test_losses_and_metrics_match_oracle_and_registry_names():
    from oct_image_segmentation_models_amd.common import custom_losses as cl, custom_metrics as cm
    assert set(cl.custom_loss_objects) == {"bce_dice_loss", "dice_loss_micro", "dice_loss_macro", "focal_loss",
                                           "bce_focal_loss", "focal_dice_loss"}
    assert set(cm.training_monitor_metric_objects) == {"dice_coef_macro", "dice_coef_micro"}
    rng = np.random.default_rng(0)
    lab = rng.integers(0, 3, (2, 8, 16, 1))
    y = on.one_hot(lab, 3, np.float64)
    p = on.softmax(rng.normal(size=(2, 8, 16, 3)))
    for name, ref in (("dice_loss_macro", on.dice_loss_macro), ("dice_loss_micro", on.dice_loss_micro)):
        f = cl.custom_loss_objects[name]["function"](num_classes=3, is_y_true_sparse=False)
        assert f.oct_loss == name and abs(f(y, p) - ref(y, p)) < 1e-12
        fs = cl.custom_loss_objects[name]["function"](num_classes=3, is_y_true_sparse=True)
        assert abs(fs(lab, p) - ref(y, p)) < 1e-12
    fm = cm.dice_coef_macro(False, 3); fi = cm.dice_coef_micro(False, 3)
    assert fm.__name__ == "dice_coef_macro" and fi.__name__ == "dice_coef_micro"
    assert abs(fm(y, p) - on.dice_coef_macro(y, p)) < 1e-6 and abs(fi(y, p) - on.dice_coef_micro(y, p)) < 1e-6
    with pytest.raises(NotImplementedError):
        cl.custom_loss_objects["focal_loss"]["function"](num_classes=3, is_y_true_sparse=True)
    yc = np.transpose(y, (0, 3, 1, 2)); pc = np.transpose(p, (0, 3, 1, 2))
    assert np.allclose(cm.soft_dice_class(yc, pc), on.soft_dice_class(yc, pc))


def test_data_generator_contract():
    from oct_image_segmentation_models_amd.common.data_generator import DataGenerator
    rng = np.random.default_rng(1)
    images = rng.integers(0, 256, (10, 4, 6, 1)).astype(np.uint8)
    labels = rng.integers(0, 3, (10, 4, 6, 1)).astype(np.uint8)
    onehot = np.eye(3, dtype=np.float32)[labels[..., 0]]
    g = DataGenerator(images, onehot, 4, [], "none", (), False, lambda x: x / 255.0, seed=5)
    assert len(g) == 2 and g.get_total_samples() == 10        # floor(10/4): tail dropped
    order0 = g.batch_gen.sample_shuffle.copy()
    assert sorted(order0) == list(range(10)) and not np.array_equal(order0, np.arange(10))
    X, y = g[123]                                             # index is ignored: sequential consumption
    assert X.dtype == np.float32 and X.shape == (4, 4, 6, 1) and y.dtype == np.float64 and y.shape == (4, 4, 6, 3)
    assert np.array_equal(X, images[order0[:4]].astype(np.float32) / np.float32(255))
    assert np.array_equal(y, onehot[order0[:4]])
    xu, lu = g.next_batch_u8()
    assert xu.dtype == np.uint8 and np.array_equal(xu, images[order0[4:8]]) and np.array_equal(lu, labels[order0[4:8], ..., 0])
    g.on_epoch_end()
    order1 = g.batch_gen.sample_shuffle
    assert sorted(order1) == list(range(10)) and not np.array_equal(order0, order1)
    assert g.batch_gen.full_counter == 0
    # a DP rank's slice of the global batch: only those samples are gathered, the shuffled order advances by a whole batch
    xs, ls = g.next_batch_u8((2, 4))
    assert g.batch_size == 4 and xs.shape[0] == 2 and np.array_equal(xs, images[order1[2:4]]) and np.array_equal(ls, labels[order1[2:4], ..., 0])
    xn, _ = g.next_batch_u8((0, 2))
    assert np.array_equal(xn, images[order1[4:6]])
    # the sparse-label cache of a (N,H,W,1) uint8 array is a view, not a copy
    assert np.shares_memory(g.batch_gen.sparse_labels(), labels) is False      # (one-hot labels here: argmax makes a new array)
    gv = DataGenerator(images, labels, 4, [], "none", (), False, None, seed=5)
    assert np.shares_memory(gv.batch_gen.sparse_labels(), labels)
    # wrap-around of full_counter when batches overrun the sample count
    g2 = DataGenerator(images[:5], labels[:5], 4, [], "none", (), False, None, seed=1)
    idx = [g2.batch_gen._next_indices() for _ in range(3)]
    assert np.array_equal(np.concatenate(idx)[:10], np.tile(g2.batch_gen.sample_shuffle, 3)[:10])
    with pytest.raises(ValueError):
        DataGenerator(images, labels, 4, [], "one", (), False, None)   # an augmentation mode needs augmentations


def test_data_generator_non_uint8_dataset_is_normalised_not_passed_raw():
    """the reference divides by 255 whatever the dataset dtype; only uint8 may take the device /255 path."""
    from oct_image_segmentation_models_amd.common.data_generator import DataGenerator
    rng = np.random.default_rng(3)
    images = rng.integers(0, 256, (6, 4, 6, 1)).astype(np.float32)      # float32 in [0, 255]
    labels = rng.integers(0, 3, (6, 4, 6, 1)).astype(np.uint8)
    g = DataGenerator(images, labels, 2, [], "none", (), False, None, seed=1)
    assert g.oct_fast_path is False
    X, _ = g[0]
    assert X.dtype == np.float32 and X.max() <= 1.0
    assert np.array_equal(X, images[g.batch_gen.sample_shuffle[:2]] / np.float32(255))
    with pytest.raises(TypeError):
        g.next_batch_u8()
    assert DataGenerator(images.astype(np.uint8), labels, 2, [], "none", (), False, None, seed=1).oct_fast_path is True


def test_data_generator_augmentation_modes():
    """'all' / 'one' modes, fly / pre-computed, flip exact, noise within its documented range."""
    from oct_image_segmentation_models_amd.common import augmentation as aug
    from oct_image_segmentation_models_amd.common.data_generator import DataGenerator
    rng = np.random.default_rng(2)
    images = rng.integers(0, 256, (6, 4, 8, 1)).astype(np.uint8)
    labels = rng.integers(0, 3, (6, 4, 8, 1)).astype(np.uint8)
    fns = [(aug.augmentation_map["no_augmentation"], {}), (aug.augmentation_map["flip"], {"flip_type": "left-right"}),
           (aug.augmentation_map["flip"], {"flip_type": "up-down"})]
    for fly in (True, False):
        g = DataGenerator(images, labels, 3, fns, "all", (), fly, lambda x: x / 255.0, seed=1)
        assert g.get_total_samples() == 18 and len(g) == 6 and not g.oct_fast_path
        order = g.batch_gen.sample_shuffle.copy()
        X, y = g[0]          # first image with each of the three augmentations, in registry order
        f = images[order[0]].astype(np.float32) / np.float32(255)
        assert X.dtype == np.float32 and np.array_equal(X[0], f) and np.array_equal(X[1], f[:, ::-1]) and np.array_equal(X[2], f[::-1])
        assert np.array_equal(y[1], labels[order[0]][:, ::-1]) and np.array_equal(y[2], labels[order[0]][::-1])
        X2, _ = g[1]
        assert np.array_equal(X2[0], images[order[1]].astype(np.float32) / np.float32(255))
    g = DataGenerator(images, labels, 6, fns, "one", (0.0, 1.0, 0.0), True, None, seed=3)
    assert g.get_total_samples() == 6 and len(g) == 1
    order = g.batch_gen.sample_shuffle.copy()
    X, y = g[0]
    assert all(np.array_equal(X[k], (images[order[k]].astype(np.float32) / np.float32(255))[:, ::-1]) for k in range(6))
    with pytest.raises(ValueError):
        DataGenerator(images, labels, 3, fns, "one", (0.5, 0.5), True, None)
    aug.seed(0)
    img = images[0].astype(np.float64) / 255.0
    for mode in ("gaussian", "speckle", "s&p"):
        out, lab = aug.add_noise_aug(img, labels[0], {"mode": mode, "mean": 0.0, "variance": 0.01})
        assert out.shape == img.shape and out.min() >= 0 and out.max() <= 1 and np.shares_memory(lab, labels) and not np.array_equal(out, img)
    assert aug.flip_aug(None, None, {"flip_type": "up-down"}, True) == "flip aug: up-down"


def test_postprocess_helpers_match_oracle():
    from oct_image_segmentation_models_amd.common import utils as cu
    rng = np.random.default_rng(3)
    _, labels = on.synth_scans(2, 32, 48, 4, seed=4)
    probs = on.softmax(rng.normal(size=(2, 32, 48, 4)) + 4 * np.eye(4)[labels[..., 0]])
    am, cat = cu.perform_argmax(probs)
    am_o, cat_o = on.perform_argmax(probs)
    assert np.array_equal(am, am_o) and np.array_equal(cat, cat_o) and cat.dtype == np.float32
    assert np.array_equal(cu.labels_to_categorical(am, 4), cat)
    for kw in (dict(), dict(bg_ilm=False), dict(bg_csi=True)):
        assert np.array_equal(cu.convert_predictions_to_maps_semantic(cat.copy(), **kw),
                              on.convert_predictions_to_maps_semantic(cat, **kw))
    assert np.array_equal(cu.to_categorical(labels, 4), on.one_hot(labels, 4, np.float32))
    assert cu.to_categorical(labels, 4).shape == (2, 32, 48, 4)


def test_create_area_mask_roundtrip_and_replacement():
    from oct_image_segmentation_models_amd.common import utils as cu
    from oct_image_segmentation_models_amd.min_path_processing import utils as mu
    _, labels = on.synth_scans(1, 40, 24, 4, seed=8)
    lab = labels[0, :, :, 0]                                   # (H,W)
    segs = mu.generate_boundary(lab, axis=0)                   # (C-1, W) first row of each region
    mask = cu.create_area_mask((24, 40, 1), segs.astype(np.uint16))   # transposed frame (W,H,1)
    assert mask.shape == (24, 40, 1) and np.array_equal(mask[..., 0].T, lab)
    bad = segs.astype(np.float64).copy(); bad[0, 3] = 0; bad[2, 5] = np.nan
    m2 = cu.create_area_mask((24, 40), bad.copy().astype(np.float64).astype(np.int64, copy=True) if False else
                             np.where(np.isnan(bad), 0, bad).astype(np.int64))
    assert (m2[3] == 0).sum() == segs[1, 3]     # boundary 0 replaced by the next valid one
    assert (m2[5] == 3).sum() == 0              # last boundary missing -> replaced by image height


def test_h5io_roundtrip(tmp_path):
    from oct_image_segmentation_models_amd.common import h5io
    p = h5io.save(tmp_path / "x.hdf5", {"a": np.arange(6).reshape(2, 3), "b": np.array([1.5])},
                  {"name": np.array("abc", dtype="S10"), "k": 3})
    assert h5io.exists(tmp_path / "x.hdf5")
    d = h5io.load(tmp_path / "x.hdf5")
    assert np.array_equal(d["a"], np.arange(6).reshape(2, 3)) and d["attr:k"] == 3 and bytes(d["attr:name"]).rstrip(b"\x00") == b"abc"
    h5io.remove(tmp_path / "x.hdf5")
    assert not h5io.exists(tmp_path / "x.hdf5") and p.name.startswith("x.hdf5")


def test_params_objects_validate_like_reference(tmp_path):
    from oct_image_segmentation_models_amd.training.training_parameters import TrainingParams
    from oct_image_segmentation_models_amd import optimizers
    kw = dict(training_dataset_path=tmp_path / "d.hdf5", results_location=tmp_path, opt_con=optimizers.Adam,
              loss="dice_loss_macro", metric="dice_coef_macro", epochs=1, batch_size=2)
    with pytest.raises(SystemExit):
        TrainingParams(model_architecture=None, initial_model=None, **kw)
    with pytest.raises(SystemExit):
        TrainingParams(model_architecture="unet", initial_model=None, aug_mode="bogus", **kw)
    tp = TrainingParams(model_architecture="unet", initial_model=None, **kw)
    assert tp.model_save_monitor == ["val_dice_coef_macro", "max"] and tp.patience == 50 and tp.early_stopping
    assert optimizers.Adam(learning_rate=3e-4).get_config()["learning_rate"] == 3e-4
    assert optimizers.Adam(lr=1e-2).learning_rate == 1e-2


def test_callbacks_checkpoint_and_early_stopping(tmp_path):
    from oct_image_segmentation_models_amd.models.engine_model import EarlyStopping, ModelCheckpoint

    class Fake:
        stop_training = False
        w = [np.zeros(1)]
        def get_weights(self): return [a.copy() for a in self.w]
        def set_weights(self, w): self.w = w
        def save(self, path): self.saved = getattr(self, "saved", []) + [str(path)]; return path

    m = Fake()
    ck = ModelCheckpoint(tmp_path / "model_epoch{epoch:02d}.hdf5", save_best_only=True, monitor="val_dice_coef_macro", mode="max")
    es = EarlyStopping(monitor="val_dice_coef_macro", mode="max", patience=2, restore_best_weights=True)
    ck.set_model(m); es.set_model(m); es.on_train_begin()
    vals = [0.5, 0.7, 0.6, 0.65, 0.64]
    for ep, v in enumerate(vals):
        m.w = [np.full(1, float(ep))]
        ck.on_epoch_end(ep, {"val_dice_coef_macro": v}); es.on_epoch_end(ep, {"val_dice_coef_macro": v})
        if m.stop_training:
            break
    assert [os.path.basename(s) for s in m.saved] == ["model_epoch01.hdf5", "model_epoch02.hdf5"]
    assert m.stop_training and ep == 3 and m.w[0][0] == 1.0   # best weights (epoch index 1) restored


def test_overall_aggregation_matches_numpy_definition(tmp_path):
    from oct_image_segmentation_models_amd.common import h5io
    from oct_image_segmentation_models_amd.evaluation import evaluation as ev
    rng = np.random.default_rng(0)
    n, C, W = 3, 4, 10

    class P: pass
    p = P(); p.save_foldername = tmp_path; p.graph_search = True
    p.metrics = ["dice_coef_classes", "dice_coef_macro", "dice_coef_micro"]
    dc = rng.uniform(size=(n, C)); dm = rng.uniform(size=(n, 1)); errs = rng.normal(size=(n, C - 1, W))
    errs[0, 0, 2] = np.nan; dm[1, 0] = np.inf
    for i in range(n):
        d = tmp_path / f"image_{i}"; d.mkdir()
        h5io.save(d / ev.EVALUATION_RESULTS_FILENAME, {"dice_coef_classes": dc[i], "dice_coef_macro": dm[i], "dice_coef_micro": dm[i] * 0.5})
        h5io.save(d / ev.GS_EVALUATION_RESULTS_FILENAME, {"dice_coef_classes": dc[i], "dice_coef_macro": dm[i], "dice_coef_micro": dm[i], "errors": errs[i]})
    out = ev._calc_overall_dataset_errors(p, [f"img{i}" for i in range(n)])
    assert np.allclose(out["mean_dice_coef_classes"], dc.mean(0)) and np.allclose(out["sd_dice_coef_classes"], dc.std(0))
    assert np.allclose(out["mean_dice_coef_macro"], np.nanmean(np.where(np.isinf(dm), np.nan, dm), 0))
    assert np.allclose(out["mean_abs_errors"], np.nanmean(np.nanmean(np.abs(errs), axis=2), axis=0))
    assert np.allclose(out["median_errors"], np.nanmedian(np.nanmean(errs, axis=2), axis=0))
    txt = open(tmp_path / ev.OVERALL_EVALUATION_RESULTS_FILENAME_CSV).read().splitlines()
    assert txt[0].startswith("Mean dice_coef_classes,") and any(l.startswith("SD errors,") for l in txt)


def test_focal_dice_loss_factory_matches_oracle_and_compiles():
    """Registry entry ``focal_dice_loss`` (reference custom_losses.py:163-178, sparse labels) vs the oracle."""
    from oracle import unet_numpy as on
    from oct_image_segmentation_models_amd.common import custom_losses
    from oct_image_segmentation_models_amd.models.engine_model import Model
    entry = custom_losses.custom_loss_objects["focal_dice_loss"]
    assert entry["takes_sparse"] is True
    rng = np.random.default_rng(0)
    z = rng.normal(size=(2, 6, 7, 4)); p = np.exp(z) / np.exp(z).sum(-1, keepdims=True)
    lab = rng.integers(0, 4, (2, 6, 7, 1)).astype(np.uint8)
    for kw in (dict(), dict(gamma=1.5, class_weight=[1, 2, 3, 4], focal_loss_weight=0.2, dice_macro=False)):
        fn = entry["function"](num_classes=4, is_y_true_sparse=True, **kw)
        ref = on.focal_dice_loss(lab, p, 4, kw.get("gamma", 2), kw.get("class_weight"), kw.get("focal_loss_weight", 0.5),
                                 kw.get("dice_macro", True))
        assert abs(fn(lab, p) - ref) < 1e-12
        m = Model("unet", dict(input_channels=1, num_classes=4, image_height=16, image_width=32))
        m.compile(optimizer=None, loss=fn, metrics=[])
        assert m._loss_name == "focal_dice_loss" and m._focal["dice_macro"] == kw.get("dice_macro", True)
    with pytest.raises(ValueError):
        entry["function"](num_classes=4, class_weight=[1, 2])
