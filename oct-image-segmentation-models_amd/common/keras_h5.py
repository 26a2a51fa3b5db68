"""Keras-H5 checkpoint import / export.

The reference checkpoints with ``tf.keras.callbacks.ModelCheckpoint(filepath=".../model_epoch{epoch:02d}.hdf5")``
(training/training.py:319-326) and reloads with ``tf.keras.models.load_model`` + the sibling ``model_config.json``
(common/utils.py:63-69).  This module maps between that file layout and the engine's weight list so that a model
trained by the reference loads into the engine and vice versa.

Layout (the published Keras 2.x HDF5 format; TensorFlow itself is not in this repository's environments, so the
layout is restated from its documentation -- PARITY UNPINNED against a real Keras-written file)::

    /                      attrs: model_config (JSON, full-model files only), keras_version, backend
    /model_weights         (full-model files; weights-only files put the next level at the root)
        attrs: layer_names = [b"input_1", b"conv2d", b"batch_normalization", b"activation", ...]
        /<layer>           attrs: weight_names = [b"<layer>/kernel:0", b"<layer>/bias:0"]   (empty for weightless layers)
            /<layer>/kernel:0   float32 (kh, kw, cin, cout)        Conv2D:  kernel:0, bias:0
            /<layer>/gamma:0    float32 (c,)                        BatchNormalization: gamma:0, beta:0,
                                                                    moving_mean:0, moving_variance:0

Layer names are Keras' automatic ones, ``conv2d[_n]`` / ``batch_normalization[_n]`` numbered in creation order --
which for ``UNet.build_model`` (models/unet.py:106-153) is exactly the engine's layer order: every conv block is
Conv2D -> BatchNormalization, the 1x1 softmax head is the last Conv2D and has no BatchNormalization.  A process that
built other models first numbers its layers from an offset, so import orders the layers by their numeric suffix
instead of trusting absolute names.

Real files are touched through ``h5py`` when importable, else through ``common/h5lite.py`` (libhdf5 via ctypes);
every function also takes an optional ``h5`` backend (anything with ``File(path, mode)`` returning an h5py-like
object) -- some tests inject an in-memory stand-in -- and without any backend a clear error is raised (no silent
fallback).
"""
from __future__ import annotations

import json
import re
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

CONV_WEIGHTS = ("kernel:0", "bias:0")
BN_WEIGHTS = ("gamma:0", "beta:0", "moving_mean:0", "moving_variance:0")
KERAS_VERSION = b"2.9.0"


class KerasH5Error(RuntimeError):
    pass


def _backend(h5=None):
    if h5 is not None:
        return h5
    try:
        import h5py  # type: ignore
        return h5py
    except Exception:  # noqa: BLE001
        pass
    from . import h5lite
    if h5lite.available():       # the HDF5 C library through ctypes: real files without the h5py wheel
        return h5lite
    raise KerasH5Error("reading or writing Keras .hdf5 checkpoints needs h5py or the HDF5 C library (libhdf5.so), "
                       "neither is available here; use the engine's own .npz checkpoints (Model.save)")


def conv_plan(config: dict) -> List[Tuple[int, int, int, int, bool]]:
    """(kh, kw, cin, cout, has_bn) of every Conv2D of ``UNet.build_model`` in creation order (models/unet.py:106-153)."""
    sn, P, L = int(config.get("start_neurons", 8)), int(config.get("pool_layers", 4)), int(config.get("conv_layers", 2))
    ek = tuple(config.get("enc_kernel", (3, 3))); dk = tuple(config.get("dec_kernel", (2, 2)))
    plan, cin = [], int(config["input_channels"])
    for i in range(P + 1):                      # encoder levels + bottleneck
        for _ in range(L):
            plan.append((ek[0], ek[1], cin, sn * 2 ** i, True)); cin = sn * 2 ** i
    for i in range(P):                          # decoder: up-conv, then L convs on concat([up, skip])
        size = sn * 2 ** (P - 1 - i)
        plan.append((dk[0], dk[1], cin, size, True)); cin = 2 * size
        for _ in range(L):
            plan.append((ek[0], ek[1], cin, size, True)); cin = size
    plan.append((1, 1, cin, int(config["num_classes"]), False))
    return plan


def _auto_name(base: str, index: int) -> str:
    return base if index == 0 else f"{base}_{index}"


def weights_to_layers(weights: Sequence[np.ndarray], config: dict) -> List[Tuple[str, Dict[str, np.ndarray]]]:
    """Engine weight list (Keras ``get_weights()`` order: per conv block kernel, bias[, gamma, beta, moving_mean,
    moving_variance]) -> [(keras layer name, {weight name: array})] in creation order."""
    plan = conv_plan(config)
    need = sum(6 if bn else 2 for *_, bn in plan)
    if len(weights) != need:
        raise KerasH5Error(f"expected {need} weight arrays for this architecture, got {len(weights)}")
    out, wi, nb = [], 0, 0
    for ci, (kh, kw, cin, cout, has_bn) in enumerate(plan):
        k, b = np.asarray(weights[wi], np.float32), np.asarray(weights[wi + 1], np.float32); wi += 2
        if k.shape != (kh, kw, cin, cout) or b.shape != (cout,):
            raise KerasH5Error(f"conv {ci}: kernel {k.shape} / bias {b.shape} do not match {(kh, kw, cin, cout)}")
        out.append((_auto_name("conv2d", ci), {"kernel:0": k, "bias:0": b}))
        if has_bn:
            arrs = [np.asarray(w, np.float32) for w in weights[wi:wi + 4]]; wi += 4
            if any(a.shape != (cout,) for a in arrs):
                raise KerasH5Error(f"batch norm {nb}: expected four vectors of length {cout}")
            out.append((_auto_name("batch_normalization", nb), dict(zip(BN_WEIGHTS, arrs)))); nb += 1
    return out


def _suffix(name: str, base: str) -> Optional[int]:
    m = re.fullmatch(re.escape(base) + r"(?:_(\d+))?", name)
    return None if m is None else int(m.group(1) or 0)


def layers_to_weights(layers: Dict[str, Dict[str, np.ndarray]], config: dict) -> List[np.ndarray]:
    """{keras layer name: {weight name: array}} (any numbering offset) -> engine weight list, validated against the
    architecture ``config`` describes."""
    convs = sorted((s, n) for n in layers if (s := _suffix(n, "conv2d")) is not None and layers[n])
    bns = sorted((s, n) for n in layers if (s := _suffix(n, "batch_normalization")) is not None and layers[n])
    plan = conv_plan(config)
    n_bn = sum(1 for *_, bn in plan if bn)
    if len(convs) != len(plan) or len(bns) != n_bn:
        raise KerasH5Error(f"checkpoint has {len(convs)} Conv2D / {len(bns)} BatchNormalization layers with weights; "
                           f"this architecture needs {len(plan)} / {n_bn}")
    out, bi = [], 0
    for (kh, kw, cin, cout, has_bn), (_, cname) in zip(plan, convs):
        g = layers[cname]
        try:
            k, b = np.asarray(g["kernel:0"], np.float32), np.asarray(g["bias:0"], np.float32)
        except KeyError as e:
            raise KerasH5Error(f"layer {cname}: missing {e}") from e
        if k.shape != (kh, kw, cin, cout) or b.shape != (cout,):
            raise KerasH5Error(f"layer {cname}: kernel {k.shape} / bias {b.shape} do not match {(kh, kw, cin, cout)} "
                               "(different hyper-parameters in model_config.json?)")
        out += [k, b]
        if has_bn:
            bname = bns[bi][1]; bi += 1
            g = layers[bname]
            for w in BN_WEIGHTS:
                if w not in g or np.asarray(g[w]).shape != (cout,):
                    raise KerasH5Error(f"layer {bname}: {w} missing or not of length {cout}")
                out.append(np.asarray(g[w], np.float32))
    return out


def _as_str(x) -> str:
    return x.decode() if isinstance(x, (bytes, np.bytes_)) else str(x)


def export_keras_h5(path, weights: Sequence[np.ndarray], config: dict, h5=None, full_model: bool = True) -> Path:
    """Write the weights in the Keras HDF5 layout.  ``full_model`` nests them under ``/model_weights`` as
    ``model.save`` does (the architecture JSON Keras also stores there is NOT written: rebuild the graph with the
    reference's ``UNet(**model_config).build_model()`` and call ``load_weights(path)``, which reads either nesting)."""
    be = _backend(h5)
    layers = weights_to_layers(weights, config)
    path = Path(path)
    with be.File(str(path), "w") as f:
        f.attrs["keras_version"] = KERAS_VERSION
        f.attrs["backend"] = b"tensorflow"
        f.attrs["oct_model_config"] = json.dumps(config).encode()      # not a Keras attribute; ignored by Keras
        root = f.create_group("model_weights") if full_model else f
        root.attrs["layer_names"] = np.array([n.encode() for n, _ in layers], dtype="S")
        root.attrs["keras_version"] = KERAS_VERSION
        root.attrs["backend"] = b"tensorflow"
        for name, ws in layers:
            g = root.create_group(name)
            g.attrs["weight_names"] = np.array([f"{name}/{w}".encode() for w in ws], dtype="S")
            inner = g.create_group(name)
            for w, arr in ws.items():
                inner.create_dataset(w, data=np.ascontiguousarray(arr, np.float32))
    return path


def import_keras_h5(path, config: dict, h5=None) -> List[np.ndarray]:
    """Read a Keras ``.hdf5`` / ``.h5`` file (full model or weights only) into the engine's weight list."""
    be = _backend(h5)
    with be.File(str(path), "r") as f:
        root = f["model_weights"] if "model_weights" in f else f
        if "layer_names" not in root.attrs:
            raise KerasH5Error(f"{path}: no 'layer_names' attribute -- not a Keras weights file")
        layers: Dict[str, Dict[str, np.ndarray]] = {}
        for raw in root.attrs["layer_names"]:
            name = _as_str(raw)
            g = root[name]
            ws = {}
            for wraw in (g.attrs["weight_names"] if "weight_names" in g.attrs else []):
                wname = _as_str(wraw)                       # "<layer>/kernel:0"
                node = g
                for part in wname.split("/"):
                    node = node[part]
                ws[wname.split("/")[-1]] = np.asarray(node[()])
            layers[name] = ws
    return layers_to_weights(layers, config)


def is_keras_h5_path(path) -> bool:
    p = Path(path)
    return p.suffix in (".h5", ".hdf5") and p.exists() and not Path(str(p) + ".npz").exists()


def read_embedded_config(path, h5=None) -> Optional[dict]:
    """The architecture config this package embeds in files it exports (None for files written by Keras itself, whose
    architecture comes from the sibling ``model_config.json`` exactly as in the reference, common/utils.py:68-69)."""
    be = _backend(h5)
    with be.File(str(path), "r") as f:
        if "oct_model_config" in f.attrs:
            return json.loads(_as_str(f.attrs["oct_model_config"]))
    return None


def have_h5py() -> bool:
    """True when real HDF5 files can be read and written here (h5py, or libhdf5 through ``h5lite``)."""
    try:
        import h5py  # type: ignore  # noqa: F401
        return True
    except Exception:  # noqa: BLE001
        from . import h5lite
        return h5lite.available()
