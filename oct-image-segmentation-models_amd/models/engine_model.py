"""``Model``: the duck-typed subset of ``tf.keras.Model`` that the reference's callers use
(``compile/fit/predict/summary/save/get_weights/set_weights/stop_training``, ``.name``,
``.output.shape[-1]``), backed by the HIP engine instead of TensorFlow.

Reference call sites this object serves: training/training.py:262-266 (compile), :345,:400 (summary),
:401-407 (fit with Sequence generators + callbacks), evaluation/evaluation.py:129-135 and
prediction/prediction.py:75-81 (predict), evaluation_parameters.py:85 (``output.shape[-1]``).

Data parallelism: when ``torch.distributed`` is initialised (one process per GPU), ``fit`` takes this rank's
contiguous slice of every GLOBAL batch the generator yields, scales the loss by 1/world and sum-all-reduces
the flat gradient buffer -- the semantics of ``MirroredStrategy`` (training.py:185-188,243).
"""
from __future__ import annotations

import json
import time
from pathlib import Path
from types import SimpleNamespace
from typing import List, Optional

import numpy as np
import torch

from .. import parallel
from .._hip import OctError

WEIGHTS_FORMAT = "oct_unet_weights_v1"


class Callback:
    """Minimal ``keras.callbacks.Callback`` protocol."""
    model = None

    def set_model(self, model): self.model = model
    def on_train_begin(self, logs=None): pass
    def on_train_end(self, logs=None): pass
    def on_epoch_begin(self, epoch, logs=None): pass
    def on_epoch_end(self, epoch, logs=None): pass


class ModelCheckpoint(Callback):
    """``ModelCheckpoint(filepath, save_best_only, monitor, mode)`` as used at training.py:319-326;
    ``filepath`` may contain ``{epoch:02d}`` (1-based)."""

    def __init__(self, filepath, save_best_only=False, monitor="val_loss", mode="auto", verbose=0):
        self.filepath, self.save_best_only, self.monitor = str(filepath), save_best_only, monitor
        if mode == "auto":
            mode = "max" if ("acc" in monitor or "dice" in monitor) else "min"
        self.sign = 1.0 if mode == "max" else -1.0
        self.best = -np.inf
        self.saved: List[str] = []

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        path = self.filepath.format(epoch=epoch + 1, **logs)
        if self.save_best_only:
            cur = logs.get(self.monitor)
            if cur is None or not np.isfinite(cur) or self.sign * cur <= self.best:
                return
            self.best = self.sign * cur
        self.saved.append(str(self.model.save(path)))


class EarlyStopping(Callback):
    """``EarlyStopping(monitor, mode, patience, restore_best_weights)`` as used at training.py:335-342."""

    def __init__(self, monitor="val_loss", mode="auto", patience=0, restore_best_weights=False, min_delta=0.0):
        if mode == "auto":
            mode = "max" if ("acc" in monitor or "dice" in monitor) else "min"
        self.monitor, self.patience, self.restore_best_weights, self.min_delta = monitor, patience, restore_best_weights, min_delta
        self.sign = 1.0 if mode == "max" else -1.0
        self.best, self.wait, self.best_weights, self.stopped_epoch = -np.inf, 0, None, 0

    def on_train_begin(self, logs=None):
        self.best, self.wait, self.best_weights, self.stopped_epoch = -np.inf, 0, None, 0

    def on_epoch_end(self, epoch, logs=None):
        cur = (logs or {}).get(self.monitor)
        if cur is None:
            return
        if self.restore_best_weights and self.best_weights is None:
            self.best_weights = self.model.get_weights()
        if np.isfinite(cur) and self.sign * cur - self.min_delta > self.best:
            self.best, self.wait = self.sign * cur, 0
            if self.restore_best_weights:
                self.best_weights = self.model.get_weights()
            return
        self.wait += 1
        if self.wait >= self.patience and epoch > 0:
            self.stopped_epoch = epoch
            self.model.stop_training = True
            if self.restore_best_weights and self.best_weights is not None:
                self.model.set_weights(self.best_weights)


class History(Callback):
    def on_train_begin(self, logs=None):
        self.history, self.epoch = {}, []

    def on_epoch_end(self, epoch, logs=None):
        self.epoch.append(epoch)
        for k, v in (logs or {}).items():
            self.history.setdefault(k, []).append(v)


class Model:
    def __init__(self, name: str, config: dict, device: Optional[str] = None):
        self.name = name
        self.config = dict(config)
        self.output = SimpleNamespace(shape=(None, None, None, int(config["num_classes"])))
        self.stop_training = False
        self.optimizer = None
        self._loss_name = self._metric_name = None
        self._engine = None
        self._pending_weights = None
        self._device = device
        self.history = None

    # ---- engine lifetime ----------------------------------------------------------------------------
    def _dev(self):
        if self._device is not None:
            return torch.device(self._device)
        _, local_rank, _ = parallel.env_rank()
        return torch.device("cuda", local_rank if torch.cuda.is_available() and local_rank < max(torch.cuda.device_count(), 1) else 0)

    def _ensure_engine(self, batch: int, training: bool):
        from ..engine import UNetEngine
        e = self._engine
        if e is not None and e.cfg.max_batch >= batch and (e.cfg.training or not training):
            return e
        weights = e.get_weights() if e is not None else self._pending_weights
        opt_state = None if e is None else (e._opt, e.opt_step)
        rank, _, _ = parallel.env_rank()
        c = self.config
        self._engine = UNetEngine(
            device=self._dev(), input_channels=c["input_channels"], num_classes=c["num_classes"],
            image_height=c["image_height"], image_width=c["image_width"],
            start_neurons=c.get("start_neurons", 8), pool_layers=c.get("pool_layers", 4),
            conv_layers=c.get("conv_layers", 2), enc_kernel=tuple(c.get("enc_kernel", (3, 3))),
            dec_kernel=tuple(c.get("dec_kernel", (2, 2))),
            max_batch=max(batch, e.cfg.max_batch if e is not None else 1),
            training=training or (e is not None and bool(e.cfg.training)),
            seed=int(c.get("seed", 0)) * 1000003 + rank, init_seed=int(c.get("seed", 0)),
            dtype=c.get("dtype", "float32"))   # extension key: 'bfloat16' = bf16 activation storage (BASELINE configs[2])
        if weights is not None:
            self._engine.set_weights(weights)
        if opt_state is not None:
            self._engine._opt, self._engine.opt_step = opt_state
        self._pending_weights = None
        return self._engine

    @property
    def engine(self):
        return self._ensure_engine(1, False)

    # ---- keras.Model surface -------------------------------------------------------------------------
    def compile(self, optimizer=None, loss=None, metrics=None, **kwargs):
        loss_name = getattr(loss, "oct_loss", None) if loss is not None else None
        if loss is not None and loss_name not in ("dice_loss_macro", "dice_loss_micro", "focal_dice_loss"):
            raise OctError("compile(loss=...): only the Dice losses and focal_dice_loss from common.custom_losses are "
                           "implemented by the HIP engine (the loss arithmetic is fused into the head kernels)")
        self._focal = dict(getattr(loss, "oct_focal", None) or {}) if loss_name == "focal_dice_loss" else None
        metric_name = None
        for m in metrics or []:
            metric_name = getattr(m, "oct_metric", None)
            if metric_name not in ("dice_coef_macro", "dice_coef_micro"):
                raise OctError("compile(metrics=...): only dice_coef_macro / dice_coef_micro are implemented")
        if optimizer is not None and not hasattr(optimizer, "apply"):
            raise OctError("compile(optimizer=...): pass an optimizers.Adam / optimizers.SGD instance")
        self.optimizer, self._loss_name, self._metric_name = optimizer, loss_name, metric_name

    def count_params(self) -> int:
        from ..engine import make_cfg
        from .. import _hip
        import ctypes as C
        c = self.config
        cfg = make_cfg(input_channels=c["input_channels"], num_classes=c["num_classes"], image_height=c["image_height"],
                       image_width=c["image_width"], start_neurons=c.get("start_neurons", 8),
                       pool_layers=c.get("pool_layers", 4), conv_layers=c.get("conv_layers", 2))
        l = _hip.lib()
        return int(l.oct_unet_param_count(C.byref(cfg)) + l.oct_unet_state_count(C.byref(cfg)))

    def summary(self, print_fn=print):
        from ..engine import make_cfg, layer_table
        c = self.config
        cfg = make_cfg(input_channels=c["input_channels"], num_classes=c["num_classes"], image_height=c["image_height"],
                       image_width=c["image_width"], start_neurons=c.get("start_neurons", 8),
                       pool_layers=c.get("pool_layers", 4), conv_layers=c.get("conv_layers", 2))
        print_fn(f'Model: "{self.name}"  (MI355X HIP engine)')
        print_fn(f"{'layer':14s}{'kernel':>8s}{'in':>6s}{'out':>6s}{'HxW':>12s}{'params':>10s}")
        tot = 0
        for L in layer_table(cfg):
            n = L["kh"] * L["kw"] * L["cin"] * L["cout"] + L["cout"] + (4 * L["cout"] if L["has_bn"] else 0)
            tot += n
            print_fn(f"{L['name']:14s}{str(L['kh']) + 'x' + str(L['kw']):>8s}{L['cin']:>6d}{L['cout']:>6d}"
                     f"{str(L['out_h']) + 'x' + str(L['out_w']):>12s}{n:>10d}")
        print_fn(f"Total params: {tot}")

    def get_weights(self):
        if self._engine is None:
            if self._pending_weights is not None:
                return [np.array(w) for w in self._pending_weights]
            return self.engine.get_weights()
        return self._engine.get_weights()

    def set_weights(self, weights):
        if self._engine is None:
            self._pending_weights = [np.asarray(w, np.float32) for w in weights]
        else:
            self._engine.set_weights(weights)

    def save(self, filepath, **kwargs) -> Path:
        """Write architecture config + weights (Keras ``get_weights()`` order).  A ``.h5`` / ``.hdf5`` path is
        written in the Keras HDF5 weight layout when ``h5py`` is importable (``common/keras_h5.py``: loadable by the
        reference's ``UNet(**cfg).build_model().load_weights``); otherwise -- and for any other suffix -- an ``.npz``
        container is written (``foo.hdf5`` becomes ``foo.hdf5.npz``)."""
        from ..common import keras_h5
        path = Path(filepath)
        as_h5 = path.suffix in (".h5", ".hdf5") and keras_h5.have_h5py()
        if not as_h5 and path.suffix != ".npz":
            path = Path(str(path) + ".npz")
        w = self.get_weights()
        if parallel.world_size() > 1 and self._engine is not None:
            # BN moving statistics are per replica; they are mean-reduced when read
            avg = parallel.average_moving_stats(self._engine.state)
            keep = self._engine.state.clone(); self._engine.state.copy_(avg)
            w = self._engine.get_weights(); self._engine.state.copy_(keep)
        payload = {f"w{i:03d}": a for i, a in enumerate(w)}
        payload["format"] = np.array(WEIGHTS_FORMAT)
        payload["name"] = np.array(self.name)
        payload["config_json"] = np.array(json.dumps(self.config))
        if parallel.env_rank()[0] == 0:
            path.parent.mkdir(parents=True, exist_ok=True)
            if as_h5:
                keras_h5.export_keras_h5(path, w, self.config)
            else:
                np.savez(path, **payload)
        return path

    # ---- training -------------------------------------------------------------------------------------
    def _host_batch(self, seq, index, rank, world):
        """One GLOBAL batch from the Sequence -> this rank's (x, sparse uint8 labels) host arrays."""
        if getattr(seq, "oct_fast_path", False):
            if world > 1:      # gather this rank's slice only (8 ranks: 1/8 of the host work per step and rank)
                return seq.next_batch_u8(parallel.shard_batch(seq.batch_size, rank, world))
            X, lab = seq.next_batch_u8()
        else:
            X, y = seq[index]
            X = np.ascontiguousarray(X, dtype=np.float32)
            y = np.asarray(y)
            lab = (np.argmax(y, axis=-1) if (y.ndim == 4 and y.shape[-1] > 1) else y.reshape(y.shape[:3])).astype(np.uint8)
        lo, hi = parallel.shard_batch(X.shape[0], rank, world)
        return X[lo:hi], lab[lo:hi]

    def _upload(self, X: np.ndarray, lab: np.ndarray, slot: int):
        """Queue the upload of one batch into device slot ``slot`` on the COPY stream (never on the compute stream: 8 MB of
        uint8 per 32-scan batch would otherwise sit in front of every step).  Pinned staging buffers and device buffers come
        in THREE slots: slot s is refilled only after the step that last read its device tensors has finished -- a HOST
        wait on that step's event, two steps back, which also keeps this thread at most two steps ahead of the GPU.  (With two
        slots the copy had to wait for the previous step on the copy stream; the H2D call then held the host until that
        step was over and every step started with the launch latency exposed: 0.91 of the resident-input rate.)
        Returns (x, labels, ready event)."""
        dev = self._engine.device if self._engine is not None else self._dev()
        st = self.__dict__.setdefault("_up", {"copy": None, "ev": {}, "done": {}, "dev": {}})
        if dev.type != "cuda":
            return torch.from_numpy(np.ascontiguousarray(X)), torch.from_numpy(np.ascontiguousarray(lab)), None
        if st["copy"] is None:
            st["copy"] = torch.cuda.Stream(device=dev)
        if slot in st["done"]:
            st["done"][slot].synchronize()        # host: the step that read this slot three batches ago is over
        if slot in st["ev"]:
            st["ev"][slot].synchronize()          # host: the H2D copies that last read this slot's pinned buffers have executed
        xp, lp = self._staged(("x", slot), X), self._staged(("l", slot), lab)
        bufs = st["dev"].get(slot)
        if bufs is None or bufs[0].shape != xp.shape or bufs[0].dtype != xp.dtype or bufs[1].shape != lp.shape:
            bufs = st["dev"][slot] = (torch.empty(xp.shape, dtype=xp.dtype, device=dev), torch.empty(lp.shape, dtype=lp.dtype, device=dev))
        with torch.cuda.stream(st["copy"]):
            bufs[0].copy_(xp, non_blocking=True); bufs[1].copy_(lp, non_blocking=True)
            ev = torch.cuda.Event(); ev.record(st["copy"])
        st["ev"][slot] = ev
        return bufs[0], bufs[1], ev

    def _release(self, slot: int):
        """The compute stream is done reading device slot ``slot`` (recorded behind the step that used it)."""
        st = self.__dict__.get("_up")
        if st and st["copy"] is not None:
            dev = self._engine.device
            e = torch.cuda.Event(); e.record(torch.cuda.current_stream(dev)); st["done"][slot] = e

    def _staged(self, key, arr: np.ndarray) -> torch.Tensor:
        arr = np.ascontiguousarray(arr)
        pool = self.__dict__.setdefault("_pinned", {})
        buf = pool.get(key)
        if buf is None or buf.shape != arr.shape or buf.numpy().dtype != arr.dtype:
            buf = torch.from_numpy(np.empty_like(arr))
            if torch.cuda.is_available():
                buf = buf.pin_memory()
            pool[key] = buf
        buf.numpy()[...] = arr
        return buf

    def _run_epoch(self, seq, training: bool, rank: int, world: int):
        focal = getattr(self, "_focal", None)
        macro = focal["dice_macro"] if focal else self._loss_name != "dice_loss_micro"
        acc = None
        n = len(seq)
        # the upload of batch i+1 (host gather -> pinned buffers -> H2D on the copy stream) is queued before step i is
        # launched, so it runs under that step
        nxt = self._upload(*self._host_batch(seq, 0, rank, world), 0) if n else None
        for i in range(n):
            x, lab, ready = nxt
            if ready is not None:
                torch.cuda.current_stream(x.device).wait_event(ready)
            if i + 1 < n:
                nxt = self._upload(*self._host_batch(seq, i + 1, rank, world), (i + 1) % 3)
            eng = self._ensure_engine(x.shape[0], training)
            if focal:      # (re)selected per batch: _ensure_engine may have built a new engine
                eng.set_focal_dice(focal["focal_loss_weight"], focal["gamma"], focal["class_weight"])
            elif getattr(eng, "_focal_active", False):
                eng.set_focal_dice(0.0)     # an engine last used by a focal model goes back to the plain Dice losses
            eng.forward(x, training=training, labels=lab, want_probs=False)
            loss4 = eng.loss_focal_dice() if focal else eng.loss_dice()
            if training:
                red = getattr(self, "_reducer", None)
                if red is None or red.engine is not eng:       # _ensure_engine may have built a new engine
                    red = self._reducer = parallel.GradReducer(eng, overlap=True)
                # backward + all-reduce (the decoder half on a side stream under the encoder backward)
                red.backward_and_reduce(lab, macro=macro, loss_scale=1.0 / world)
                self.optimizer.apply(eng)
            acc = loss4.clone() if acc is None else acc + loss4
            self._release(i % 3)
        if acc is None:
            return {}
        acc = acc / n
        if world > 1:
            torch.distributed.all_reduce(acc); acc /= world
        v = acc.cpu().numpy()
        out = {"loss": float((v[5] if macro else v[6]) if focal else (v[0] if macro else v[1]))}
        if self._metric_name:
            out[self._metric_name] = float(v[2] if self._metric_name == "dice_coef_macro" else v[3])
        return out

    def fit(self, x=None, validation_data=None, epochs: int = 1, callbacks=None, verbose: int = 1,
            initial_epoch: int = 0, **kwargs):
        if self.optimizer is None or self._loss_name is None:
            raise OctError("fit() before compile(optimizer=..., loss=...)")
        rank, _, world = parallel.env_rank()
        world = parallel.world_size() if world > 1 else 1
        hist = History()
        cbs = list(callbacks or []) + [hist]
        for cb in cbs:
            cb.set_model(self)
        self.stop_training = False
        for cb in cbs:
            cb.on_train_begin({})
        for epoch in range(initial_epoch, epochs):
            for cb in cbs:
                cb.on_epoch_begin(epoch, {})
            t0 = time.time()
            logs = self._run_epoch(x, True, rank, world)
            if hasattr(x, "on_epoch_end"):
                x.on_epoch_end()
            if validation_data is not None:
                vlogs = self._run_epoch(validation_data, False, rank, world)
                logs.update({"val_" + k: v for k, v in vlogs.items()})
                if hasattr(validation_data, "on_epoch_end"):
                    validation_data.on_epoch_end()
            if verbose and rank == 0:
                msg = " - ".join(f"{k}: {v:.4f}" for k, v in logs.items())
                print(f"Epoch {epoch + 1}/{epochs} - {time.time() - t0:.1f}s - {msg}")
            for cb in cbs:
                cb.on_epoch_end(epoch, logs)
            if self.stop_training:
                break
        for cb in cbs:
            cb.on_train_end({})
        self.history = hist
        return hist

    def evaluate(self, x, verbose: int = 0, **kwargs):
        rank, _, world = parallel.env_rank()
        world = parallel.world_size() if world > 1 else 1
        logs = self._run_epoch(x, False, rank, world)
        return [logs.get("loss")] + ([logs[self._metric_name]] if self._metric_name else [])

    # ---- inference --------------------------------------------------------------------------------------
    def predict(self, x, verbose=0, batch_size: Optional[int] = None, **kwargs) -> np.ndarray:
        """(n,H,W,C) float input ALREADY preprocessed to [0,1] (or raw uint8) -> (n,H,W,num_classes) float32."""
        x = np.asarray(x)
        if x.dtype != np.uint8:
            x = np.ascontiguousarray(x, dtype=np.float32)   # Keras casts the float64 the reference passes to float32
        n = x.shape[0]
        bs = int(batch_size or min(n, 32))
        eng = self._ensure_engine(min(bs, n), False)
        out = np.empty((n,) + tuple(x.shape[1:3]) + (self.config["num_classes"],), np.float32)
        for lo in range(0, n, bs):
            xb = torch.from_numpy(np.ascontiguousarray(x[lo:lo + bs])).to(eng.device)
            probs, _ = eng.forward(xb, training=False)
            out[lo:lo + bs] = probs.cpu().numpy()
        return out

    def predict_labels(self, x_u8: np.ndarray, batch_size: int = 32, want_maps: bool = False, bg_ilm: bool = True,
                       bg_csi: bool = False):
        """Raw uint8 images -> uint8 arg-max class maps (n,H,W), computed on the device (1 B/px back instead of
        4*C B/px).  With ``want_maps`` also the (n, C-1, H, W) uint8 boundary maps of
        ``convert_predictions_to_maps_semantic``, computed on the device from the class maps."""
        x_u8 = np.ascontiguousarray(x_u8)
        if x_u8.dtype != np.uint8:
            # the reference normalises x / 255 whatever the dtype (models/unet.py:87-91); the engine's float32 input
            # path means "already normalised", so do the division here rather than silently skipping it
            x_u8 = np.ascontiguousarray(x_u8.astype(np.float32) / np.float32(255.0))
        n = x_u8.shape[0]
        eng = self._ensure_engine(min(batch_size, n), False)
        out = np.empty(x_u8.shape[:3], np.uint8)
        maps = np.empty((n, self.config["num_classes"] - 1) + tuple(x_u8.shape[1:3]), np.uint8) if want_maps else None
        for lo in range(0, n, batch_size):
            xb = torch.from_numpy(x_u8[lo:lo + batch_size]).to(eng.device)
            _, am = eng.forward(xb, training=False, want_probs=False, want_argmax=True)
            out[lo:lo + batch_size] = am.cpu().numpy()
            if want_maps:
                maps[lo:lo + batch_size] = eng.boundary_maps(am, bg_ilm=bg_ilm, bg_csi=bg_csi).cpu().numpy()
        return (out, maps) if want_maps else out


def load_model(path) -> Model:
    """Counterpart of ``tf.keras.models.load_model(compile=False)`` for files written by ``Model.save``.
    Only arrays and JSON are read (``allow_pickle=False``)."""
    from ..common import keras_h5
    path = Path(path)
    if keras_h5.is_keras_h5_path(path):
        # a Keras HDF5 checkpoint (written by the reference's ModelCheckpoint, or by Model.save with h5py present)
        config = keras_h5.read_embedded_config(path)
        if config is None:
            with open(path.parent / "model_config.json", "r") as fh:     # reference: common/utils.py:68-69
                config = json.load(fh)
        m = Model(name=config.get("name", "unet") if isinstance(config.get("name"), str) else "unet", config=config)
        m.set_weights(keras_h5.import_keras_h5(path, config))
        return m
    if not path.exists() and Path(str(path) + ".npz").exists():
        path = Path(str(path) + ".npz")
    with np.load(path, allow_pickle=False) as z:
        if str(z["format"]) != WEIGHTS_FORMAT:
            raise OctError(f"{path}: unknown weights format {z['format']}")
        config = json.loads(str(z["config_json"]))
        name = str(z["name"])
        weights = [z[k] for k in sorted(k for k in z.files if k.startswith("w") and k[1:].isdigit())]
    m = Model(name=name, config=config)
    m.set_weights(weights)
    return m
