"""Result/dataset container I/O.  The reference reads and writes HDF5 through ``h5py``; here the same key/attribute
contract is served, in this order, by ``h5py`` when importable, by ``h5lite`` (this package's ctypes binding of the
HDF5 C library -- real HDF5 files, present on the build / GPU image) and, only when neither exists, by ``.npz`` files
(a ``foo.hdf5`` request becomes ``foo.hdf5.npz``; attributes are stored under ``attr:<name>`` keys)."""
from __future__ import annotations

import os
from pathlib import Path
from typing import Dict, Optional

import numpy as np

try:  # pragma: no cover - depends on the environment
    import h5py  # type: ignore
    BACKEND = "h5py"
except Exception:  # noqa: BLE001
    from . import h5lite
    if h5lite.available():
        h5py = h5lite          # same File / create_dataset / attrs surface, served by libhdf5 through ctypes
        BACKEND = "h5lite"
    else:
        h5py = None
        BACKEND = "npz"
HAVE_H5PY = h5py is not None   # "an HDF5 backend exists"


def _npz_path(path) -> Path:
    p = Path(path)
    return p if p.suffix == ".npz" else Path(str(p) + ".npz")


def save(path, datasets: Dict[str, np.ndarray], attrs: Optional[Dict[str, object]] = None) -> Path:
    """Write ``datasets`` (+ file attributes) to ``path``; returns the path actually written."""
    attrs = attrs or {}
    path = Path(path)
    if HAVE_H5PY and path.suffix != ".npz":
        with h5py.File(path, "w") as f:
            for k, v in datasets.items():
                f.create_dataset(k, data=v)
            for k, v in attrs.items():
                f.attrs[k] = v
        return path
    out = _npz_path(path)
    payload = {k: np.asarray(v) for k, v in datasets.items()}
    payload.update({f"attr:{k}": np.asarray(v) for k, v in attrs.items()})
    np.savez(out, **payload)
    return out


def load(path) -> Dict[str, np.ndarray]:
    """Read every dataset (and ``attr:<name>`` attributes) of a container written by ``save`` or of a real
    HDF5 dataset file.  Nothing in the file is executed (``allow_pickle=False``)."""
    path = Path(path)
    if path.suffix != ".npz" and path.exists() and HAVE_H5PY:
        out = {}
        with h5py.File(path, "r") as f:
            for k in f.keys():
                out[k] = f[k][()]
            for k, v in f.attrs.items():
                out[f"attr:{k}"] = np.asarray(v)
        return out
    npz = path if path.suffix == ".npz" else _npz_path(path)
    if not npz.exists():
        if path.exists():
            raise RuntimeError(f"{path} is an HDF5 file but neither h5py nor the HDF5 C library is available here; "
                               "convert it to .npz with the same keys")
        raise FileNotFoundError(str(path))
    with np.load(npz, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def exists(path) -> bool:
    path = Path(path)
    return path.exists() or _npz_path(path).exists()


def remove(path) -> None:
    for p in (Path(path), _npz_path(path)):
        try:
            os.remove(p)
        except OSError:
            pass
