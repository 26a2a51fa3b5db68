"""GPU end-to-end test of the drop-in workflow (BASELINE config[0] analogue: small synthetic dataset,
3 boundaries = 4 classes, a few epochs): train_model -> evaluate_model (+graph search) -> predict, checked
against the oracle / numpy definitions."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import unet_numpy as on

pytestmark = pytest.mark.gpu

H, W, C = 64, 128, 4
EPOCHS = 150   # 360 steps: BN moving statistics (momentum 0.99) need a few hundred steps to converge


@pytest.fixture(scope="module")
def trained(tmp_path_factory):
    from oct_image_segmentation_models_amd import optimizers
    from oct_image_segmentation_models_amd.common import h5io
    from oct_image_segmentation_models_amd.training.training import train_model
    from oct_image_segmentation_models_amd.training.training_parameters import TrainingParams
    root = tmp_path_factory.mktemp("wf")
    tr_i, tr_l = on.synth_scans(12, H, W, C, seed=1)
    va_i, va_l = on.synth_scans(4, H, W, C, seed=2)
    te_i, te_l = on.synth_scans(5, H, W, C, seed=3)
    h5io.save(root / "data.hdf5", {"train_images": tr_i, "train_labels": tr_l, "val_images": va_i, "val_labels": va_l,
                                   "test_images": te_i, "test_labels": te_l,
                                   "test_images_source": np.array([f"scan_{i}.tiff".encode() for i in range(5)])})
    tp = TrainingParams(model_architecture="unet", training_dataset_path=root / "data.hdf5", initial_model=None,
                        results_location=root / "results", opt_con=optimizers.Adam, opt_params={"learning_rate": 4e-3},
                        loss="dice_loss_macro", metric="dice_coef_macro", epochs=EPOCHS, batch_size=4,
                        model_hyperparameters={"pool_layers": 3}, patience=EPOCHS + 1, seed=7)   # no early stop: the file/epoch bookkeeping below needs all EPOCHS
    res = train_model(tp, None)
    return root, res, (te_i, te_l)


def test_train_model_outputs_and_learning(trained):
    from oct_image_segmentation_models_amd.common import h5io
    root, res, _ = trained
    d = Path(res.save_foldername)
    assert d.parent == root / "results" and d.name.endswith("_unet")
    cfg = __import__("json").load(open(d / "model_config.json"))
    assert cfg["num_classes"] == C and cfg["image_height"] == H and cfg["pool_layers"] == 3
    tp = h5io.load(d / "training_params.hdf5")
    assert bytes(tp["attr:loss_name"]).rstrip(b"\x00") == b"dice_loss_macro" and tp["attr:batch_size"] == 4
    assert bytes(tp["attr:optimizer"]).rstrip(b"\x00") == b"Adam"
    hist = res.history
    assert set(hist) == {"loss", "dice_coef_macro", "val_loss", "val_dice_coef_macro"} and len(hist["loss"]) == EPOCHS
    assert hist["loss"][-1] < 0.6 * hist["loss"][0] and max(hist["val_dice_coef_macro"]) > 0.5
    stats = h5io.load(d / f"stats_epoch{EPOCHS:02d}.hdf5")
    assert len(stats["train_loss"]) == EPOCHS and not h5io.exists(d / f"stats_epoch{EPOCHS - 1:02d}.hdf5")   # rolling file
    assert np.allclose(stats["val_acc"], hist["val_dice_coef_macro"])
    assert len(res.checkpoints) >= 1 and all(Path(p).exists() for p in res.checkpoints)   # save_best_only
    best = int(np.argmax(hist["val_dice_coef_macro"])) + 1
    assert Path(res.checkpoints[-1]).name.startswith(f"model_epoch{best:02d}")


def test_evaluate_and_predict_match_engine_and_numpy_definitions(trained):
    from oct_image_segmentation_models_amd.common import custom_metrics, h5io, utils as cu
    from oct_image_segmentation_models_amd.common.dataset import Dataset
    from oct_image_segmentation_models_amd.evaluation import eval_model
    from oct_image_segmentation_models_amd.evaluation.evaluation_parameters import EvaluationParameters, EvaluationSaveParams
    from oct_image_segmentation_models_amd.prediction import predict
    from oct_image_segmentation_models_amd.prediction.prediction_parameters import PredictionParams, PredictionSaveParams
    root, res, (te_i, te_l) = trained
    ckpt = Path(res.checkpoints[-1])
    ep = EvaluationParameters(model_path=ckpt, mlflow_tracking_uri=None, mlflow_run_uuid=None,
                              test_dataset_path=root / "data.hdf5", save_foldername=root / "eval",
                              save_params=EvaluationSaveParams(categorical_pred=True), graph_search=True,
                              metrics=["dice_coef_classes", "dice_coef_macro", "dice_coef_micro"], batch_size=2)
    assert ep.num_classes == C and ep.loaded_model.name == "unet"
    outs = eval_model(ep)
    assert len(outs) == 5
    # the loaded checkpoint reproduces Model.predict (float path, preprocessed as the reference does)
    probs = ep.loaded_model.predict(te_i / 255.0, batch_size=3)
    assert probs.shape == (5, H, W, C) and probs.dtype == np.float32
    am = probs.argmax(-1)
    for i, o in enumerate(outs):
        assert str(o.image_name) == f"scan_{i}.tiff"
        assert np.array_equal(o.predicted_labels, am[i])
        cat = cu.labels_to_categorical(am[i:i + 1], C)
        assert np.array_equal(o.boundary_maps, on.convert_predictions_to_maps_semantic(cat)[0])
        y = on.one_hot(te_l[i:i + 1], C, np.float64)
        f = h5io.load(o.image_output_dir / "evaluation_results.hdf5")
        assert np.allclose(f["dice_coef_classes"], on.soft_dice_class(np.transpose(y, (0, 3, 1, 2)), cat)[0])
        assert abs(f["dice_coef_macro"][0] - on.dice_coef_macro(y, np.transpose(cat, (0, 2, 3, 1)))) < 1e-6
        assert f["predicted_segmentation_map"].dtype == np.uint8 and f["raw_segs"].shape == (C - 1, W)
        g = h5io.load(o.image_output_dir / "gs_evaluation_results.hdf5")
        assert g["gs_pred_segs"].shape == (C - 1, W) and g["errors"].shape == (C - 1, W)
        assert np.array_equal(g["gs_pred_segs"], o.gs_pred_segs)
        assert np.nanmean(np.abs(o.errors)) < 3.0          # trained net delineates within a few pixels
        assert (root / "eval" / f"image_{i}" / "gs_boundaries.csv").exists()
    overall = h5io.load(root / "eval" / "overall_evaluation_results.hdf5")
    assert overall["dice_coef_classes"].shape == (5, C) and overall["mean_dice_coef_macro"] > 0.45
    assert overall["errors"].shape == (5, C - 1, W)

    ds = Dataset(te_i, [Path(f"scan_{i}.tiff") for i in range(5)], [root / "pred" / f"image_{i}" for i in range(5)])
    pp = PredictionParams(model_path=ckpt, mlflow_tracking_uri=None, mlflow_run_uuid=None, dataset=ds,
                          config_output_dir=root / "pred", save_params=PredictionSaveParams(), graph_search=True, batch_size=4)
    pouts = predict(pp)
    for o, p in zip(outs, pouts):
        assert np.array_equal(o.predicted_labels, p.predicted_labels) and np.array_equal(o.gs_pred_segs, p.gs_pred_segs)
        info = h5io.load(p.image_output_dir / "prediction_info.hdf5")
        assert np.array_equal(info["predicted_labels"], p.predicted_labels) and "attr:convert_time" in info


def test_resume_from_initial_model(trained, tmp_path):
    from oct_image_segmentation_models_amd import optimizers
    from oct_image_segmentation_models_amd.training.training import train
    from oct_image_segmentation_models_amd.training.training_parameters import TrainingParams
    root, res, _ = trained
    tp = TrainingParams(model_architecture=None, training_dataset_path=root / "data.hdf5",
                        initial_model=Path(res.checkpoints[-1]), results_location=tmp_path / "r2",
                        opt_con=optimizers.SGD, opt_params={"learning_rate": 1e-3, "momentum": 0.9},
                        loss="dice_loss_micro", metric="dice_coef_micro", epochs=2, batch_size=4, early_stopping=False)
    r2 = train(tp, None)
    assert r2.history["val_dice_coef_micro"][0] > 0.45      # starts from the trained weights, not from scratch


def test_train_model_with_focal_dice_loss(tmp_path):
    """Registry name ``focal_dice_loss`` (sparse labels, loss_fn_kwargs as in the reference) trains through train_model."""
    from oct_image_segmentation_models_amd import optimizers
    from oct_image_segmentation_models_amd.common import h5io
    from oct_image_segmentation_models_amd.training.training import train_model
    from oct_image_segmentation_models_amd.training.training_parameters import TrainingParams
    tr_i, tr_l = on.synth_scans(8, 32, 64, 3, seed=11)
    va_i, va_l = on.synth_scans(4, 32, 64, 3, seed=12)
    h5io.save(tmp_path / "data.hdf5", {"train_images": tr_i, "train_labels": tr_l, "val_images": va_i, "val_labels": va_l})
    tp = TrainingParams(model_architecture="unet", training_dataset_path=tmp_path / "data.hdf5", initial_model=None,
                        results_location=tmp_path / "results", opt_con=optimizers.Adam, opt_params={"learning_rate": 4e-3},
                        loss="focal_dice_loss", loss_fn_kwargs={"gamma": 2.0, "focal_loss_weight": 0.4, "class_weight": [1.0, 2.0, 1.0]},
                        metric="dice_coef_macro", epochs=25, batch_size=4, model_hyperparameters={"pool_layers": 2},
                        patience=26, seed=3)
    res = train_model(tp, None)
    h = res.history
    assert len(h["loss"]) == 25 and np.isfinite(h["loss"]).all() and np.isfinite(h["val_loss"]).all()
    assert h["loss"][-1] < 0.7 * h["loss"][0]


def test_batched_pipeline_equals_per_image_path_at_batch_128():
    """BASELINE configs[4]: 256x512, device batch 128 (+ a ragged last batch) through evaluation/pipeline.py --
    hipGraph replay, pinned double-buffered u8 upload, u8 arg-max + boundary-map download -- equals the per-image
    forward bit for bit, and the pooled min-path post-process equals segment_maps run inline."""
    from oct_image_segmentation_models_amd.common.synthetic import make_scans
    from oct_image_segmentation_models_amd.engine import UNetEngine
    from oct_image_segmentation_models_amd.evaluation.pipeline import BatchedPredictor
    from oct_image_segmentation_models_amd.min_path_processing import graph_search
    from oct_image_segmentation_models_amd.min_path_processing.pool import SegmentPool
    Hh, Ww, Cc, B, N = 256, 512, 3, 128, 160
    with SegmentPool((Hh, Ww), 1, workers=4) as pool:                 # workers before this test's engine exists
        eng = UNetEngine(device="cuda:0", input_channels=1, num_classes=Cc, image_height=Hh, image_width=Ww, max_batch=B,
                         training=False, seed=2, init_seed=4)
        img8, _ = make_scans(8, Hh, Ww, Cc, seed=9)
        imgs = np.concatenate([np.roll(img8, 5 * k, axis=2) for k in range(N // 8)], axis=0)
        pred = BatchedPredictor(eng, B, want_maps=True)
        got_lab = np.empty((N, Hh, Ww), np.uint8); got_map = np.empty((N, Cc - 1, Hh, Ww), np.uint8)
        spans = []
        for lo, hi, lab, maps in pred.run(imgs):
            spans.append((lo, hi)); got_lab[lo:hi] = lab; got_map[lo:hi] = maps
        assert spans == [(0, 128), (128, 160)]
        for i in (0, 1, 77, 127, 128, 159):                            # per-image reference path
            x = torch.from_numpy(imgs[i:i + 1]).cuda()
            _, am = eng.forward(x, training=False, want_probs=False, want_argmax=True)
            assert np.array_equal(am.cpu().numpy()[0], got_lab[i]), i
            assert np.array_equal(eng.boundary_maps(am).cpu().numpy()[0], got_map[i]), i
        again = list(pred.run(imgs[:B]))                               # the predictor is reusable
        assert np.array_equal(again[0][2], got_lab[:B])
        res = pool.segment(got_map[:6])
        grid = graph_search.create_graph_structure((Ww, Hh), 1)
        for i in range(6):
            p_inline, e_inline, _ = graph_search.segment_maps(np.transpose(got_map[i], (0, 2, 1)), None, grid)
            assert np.array_equal(res[i][0], p_inline) and np.array_equal(res[i][1], e_inline)


def test_config0_at_its_stated_size(tmp_path):
    """BASELINE configs[0] as written: 4 train (+ 4 validation) 256x512x1 scans in a real HDF5 file, 3 boundaries
    (= 4 classes), 1 epoch through ``train_model`` (reference training/training.py:135-408) -- the plumbing case,
    checked against the output-file contract of SURVEY 8(a')."""
    import json
    from oct_image_segmentation_models_amd import optimizers
    from oct_image_segmentation_models_amd.common import h5io
    from oct_image_segmentation_models_amd.training.training import train_model
    from oct_image_segmentation_models_amd.training.training_parameters import TrainingParams
    Hc, Wc, Cc = 256, 512, 4
    tr_i, tr_l = on.synth_scans(4, Hc, Wc, Cc, seed=11)
    va_i, va_l = on.synth_scans(4, Hc, Wc, Cc, seed=12)
    assert tr_i.shape == (4, Hc, Wc, 1) and tr_i.dtype == np.uint8 and sorted(np.unique(tr_l)) == [0, 1, 2, 3]
    data = tmp_path / "config0.hdf5"
    h5io.save(data, {"train_images": tr_i, "train_labels": tr_l, "val_images": va_i, "val_labels": va_l})
    assert data.exists() and open(data, "rb").read(8) == b"\x89HDF\r\n\x1a\n"          # a real HDF5 container
    tp = TrainingParams(model_architecture="unet", training_dataset_path=data, initial_model=None,
                        results_location=tmp_path / "results", opt_con=optimizers.Adam, opt_params={"learning_rate": 1e-3},
                        loss="dice_loss_macro", metric="dice_coef_macro", epochs=1, batch_size=4, seed=3)
    res = train_model(tp, None)
    d = Path(res.save_foldername)
    assert d.parent == tmp_path / "results" and d.name.endswith("_unet")
    cfg = json.load(open(d / "model_config.json"))
    assert (cfg["num_classes"], cfg["image_height"], cfg["image_width"], cfg["input_channels"]) == (Cc, Hc, Wc, 1)
    assert cfg.get("start_neurons", 8) == 8 and cfg.get("pool_layers", 4) == 4           # the reference's defaults
    tpf = h5io.load(d / "training_params.hdf5")
    assert bytes(tpf["attr:loss_name"]).rstrip(b"\x00") == b"dice_loss_macro" and tpf["attr:batch_size"] == 4
    assert tpf["attr:epochs"] == 1 and bytes(tpf["attr:optimizer"]).rstrip(b"\x00") == b"Adam"
    hist = res.history
    assert set(hist) == {"loss", "dice_coef_macro", "val_loss", "val_dice_coef_macro"} and len(hist["loss"]) == 1
    assert all(np.isfinite(v[0]) for v in hist.values()) and 0.0 < hist["loss"][0] < 1.0
    stats = h5io.load(d / "stats_epoch01.hdf5")
    assert set(stats) >= {"train_acc", "val_acc", "train_loss", "val_loss", "epoch_time"} and len(stats["train_loss"]) == 1
    assert len(res.checkpoints) == 1 and Path(res.checkpoints[0]).name.startswith("model_epoch01")
    # the checkpoint loads back into a model that predicts (n, H, W, 4) probabilities
    from oct_image_segmentation_models_amd.models.engine_model import load_model
    m = load_model(res.checkpoints[0])
    p = m.predict(va_i[:2].astype(np.float32) / 255.0, batch_size=2)
    assert p.shape == (2, Hc, Wc, Cc) and np.abs(p.sum(-1) - 1).max() < 1e-5
