"""``h5lite``: the subset of the ``h5py`` API this package touches, served straight by the HDF5 C library
(``libhdf5.so``, C ABI, through ``ctypes``) -- for environments that have the library but not the ``h5py``
wheel (the build / GPU image has libhdf5 1.10.6 under /opt/conda/lib, h5py only for another
interpreter).  With it the reference's on-disk contract is met with REAL HDF5 files:

* datasets ``train_images / train_labels / val_* / test_* / test_images_source``
  (reference common/dataset_loader.py:9-33), ``training_params.hdf5`` attributes (training/training.py:39-132),
  ``stats_epochNN.hdf5``, evaluation result files;
* Keras ``model_epochNN.hdf5`` checkpoints (``common/keras_h5.py``; training/training.py:319-326, common/utils.py:63-69).

Supported: ``File(path, "r" | "w" | "a")`` as a context manager; groups (``create_group``, ``[]`` with ``/`` paths,
``in``, ``keys()``); datasets (``create_dataset(name, data=...)``, ``[()]`` / ``[:]``, ``.shape``, ``.dtype``);
``.attrs`` on files, groups and datasets (numeric scalars / arrays, ``bytes`` and fixed-length ``S`` arrays, ``str``;
variable-length strings are read).  Files written here are read by h5py and vice versa (tests/test_hdf5_real.py
checks both directions against the real h5py of /opt/conda/bin/python3.9)."""
from __future__ import annotations

import ctypes as C
import ctypes.util
import os
import weakref
from typing import Iterator, List, Optional, Tuple

import numpy as np

hid_t = C.c_int64
hsize_t = C.c_uint64

_CANDIDATES = [os.environ.get("OCT_LIBHDF5"), ctypes.util.find_library("hdf5"), "/opt/conda/lib/libhdf5.so",
               "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so", "/usr/lib/x86_64-linux-gnu/libhdf5.so"]
_lib = None
_types = {}

# constants of hdf5 1.10 (H5Fpublic.h, H5Ipublic.h, H5Tpublic.h, H5Spublic.h)
H5F_ACC_RDONLY, H5F_ACC_RDWR, H5F_ACC_TRUNC = 0, 1, 2
H5P_DEFAULT = 0
H5S_ALL = 0
H5S_SELECT_SET = 0          # H5S_seloper_t
H5S_SCALAR = 0
H5I_GROUP, H5I_DATASET = 2, 5
H5T_INTEGER, H5T_FLOAT, H5T_STRING = 0, 1, 3
H5T_SGN_NONE = 0
H5T_STR_NULLPAD = 1
H5T_CSET_UTF8 = 1
H5T_VARIABLE = C.c_size_t(-1).value


class H5Error(OSError):
    pass


def available() -> bool:
    try:
        lib()
        return True
    except H5Error:
        return False


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    last = None
    for cand in _CANDIDATES:
        if not cand:
            continue
        try:
            l = C.CDLL(cand)
        except OSError as e:
            last = e
            continue
        sig = {
            "H5open": (C.c_int, []), "H5Eset_auto2": (C.c_int, [hid_t, C.c_void_p, C.c_void_p]),
            "H5Fcreate": (hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]), "H5Fopen": (hid_t, [C.c_char_p, C.c_uint, hid_t]),
            "H5Fclose": (C.c_int, [hid_t]),
            "H5Gcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t]), "H5Gclose": (C.c_int, [hid_t]),
            "H5Oopen": (hid_t, [hid_t, C.c_char_p, hid_t]), "H5Oclose": (C.c_int, [hid_t]),
            "H5Iget_type": (C.c_int, [hid_t]),
            "H5Lexists": (C.c_int, [hid_t, C.c_char_p, hid_t]),
            "H5Gget_info": (C.c_int, [hid_t, C.c_void_p]),
            "H5Lget_name_by_idx": (C.c_ssize_t, [hid_t, C.c_char_p, C.c_int, C.c_int, hsize_t, C.c_char_p, C.c_size_t, hid_t]),
            "H5Dcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
            "H5Dwrite": (C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
            "H5Dread": (C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
            "H5Dget_space": (hid_t, [hid_t]), "H5Dget_type": (hid_t, [hid_t]), "H5Dclose": (C.c_int, [hid_t]),
            "H5Dvlen_reclaim": (C.c_int, [hid_t, hid_t, hid_t, C.c_void_p]),
            "H5Screate_simple": (hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
            "H5Screate": (hid_t, [C.c_int]), "H5Sclose": (C.c_int, [hid_t]),
            "H5Sget_simple_extent_ndims": (C.c_int, [hid_t]),
            "H5Sselect_hyperslab": (C.c_int, [hid_t, C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t), C.POINTER(hsize_t), C.POINTER(hsize_t)]),
            "H5Sget_simple_extent_dims": (C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
            "H5Tcopy": (hid_t, [hid_t]), "H5Tset_size": (C.c_int, [hid_t, C.c_size_t]), "H5Tclose": (C.c_int, [hid_t]),
            "H5Tset_strpad": (C.c_int, [hid_t, C.c_int]), "H5Tset_cset": (C.c_int, [hid_t, C.c_int]),
            "H5Tget_class": (C.c_int, [hid_t]), "H5Tget_size": (C.c_size_t, [hid_t]), "H5Tget_sign": (C.c_int, [hid_t]),
            "H5Tis_variable_str": (C.c_int, [hid_t]),
            "H5Acreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t]),
            "H5Awrite": (C.c_int, [hid_t, hid_t, C.c_void_p]), "H5Aread": (C.c_int, [hid_t, hid_t, C.c_void_p]),
            "H5Aopen": (hid_t, [hid_t, C.c_char_p, hid_t]), "H5Aclose": (C.c_int, [hid_t]),
            "H5Aexists": (C.c_int, [hid_t, C.c_char_p]), "H5Adelete": (C.c_int, [hid_t, C.c_char_p]),
            "H5Aget_space": (hid_t, [hid_t]), "H5Aget_type": (hid_t, [hid_t]),
            "H5Aiterate2": (C.c_int, [hid_t, C.c_int, C.c_int, C.POINTER(hsize_t), C.c_void_p, C.c_void_p]),
        }
        try:
            for name, (res, args) in sig.items():
                fn = getattr(l, name)
                fn.restype, fn.argtypes = res, args
        except AttributeError as e:
            last = e
            continue
        if l.H5open() < 0:
            last = H5Error("H5open failed")
            continue
        l.H5Eset_auto2(0, None, None)        # errors are reported through return codes / exceptions, not stderr dumps
        for np_name, sym in (("uint8", "H5T_NATIVE_UINT8_g"), ("int8", "H5T_NATIVE_INT8_g"), ("uint16", "H5T_NATIVE_UINT16_g"),
                             ("int16", "H5T_NATIVE_INT16_g"), ("uint32", "H5T_NATIVE_UINT32_g"), ("int32", "H5T_NATIVE_INT32_g"),
                             ("uint64", "H5T_NATIVE_UINT64_g"), ("int64", "H5T_NATIVE_INT64_g"),
                             ("float32", "H5T_NATIVE_FLOAT_g"), ("float64", "H5T_NATIVE_DOUBLE_g"), ("c_s1", "H5T_C_S1_g")):
            _types[np_name] = hid_t.in_dll(l, sym).value
        _lib = l
        return l
    raise H5Error(f"no usable HDF5 C library found (tried {[c for c in _CANDIDATES if c]}): {last}")


def _chk(rc, what):
    if rc < 0:
        raise H5Error(f"HDF5: {what} failed")
    return rc


def _mem_type(arr: np.ndarray) -> Tuple[int, bool]:
    """(HDF5 memory/file type for ``arr``, must_close)."""
    l = lib()
    if arr.dtype.kind == "S":
        t = _chk(l.H5Tcopy(_types["c_s1"]), "H5Tcopy")
        l.H5Tset_size(t, max(arr.dtype.itemsize, 1)); l.H5Tset_strpad(t, H5T_STR_NULLPAD)
        return t, True
    if arr.dtype.kind == "b":
        raise H5Error("bool arrays are not supported; cast to uint8")
    name = arr.dtype.name
    if name not in _types:
        raise H5Error(f"unsupported dtype {arr.dtype}")
    return _types[name], False


def _space_of(arr: np.ndarray) -> int:
    l = lib()
    if arr.ndim == 0:
        return _chk(l.H5Screate(H5S_SCALAR), "H5Screate")
    dims = (hsize_t * arr.ndim)(*arr.shape)
    return _chk(l.H5Screate_simple(arr.ndim, dims, None), "H5Screate_simple")


def _as_array(value) -> np.ndarray:
    if isinstance(value, str):
        value = value.encode("utf-8")
    if isinstance(value, (bytes, np.bytes_)):
        return np.array(value, dtype=f"S{max(len(value), 1)}")
    a = np.asarray(value)
    if a.dtype.kind == "U":
        a = np.char.encode(a, "utf-8")
    if a.dtype.kind == "O":
        raise H5Error("object arrays are not supported")
    if a.dtype.kind == "b":
        a = a.astype(np.uint8)
    return np.require(a, requirements="C")        # (np.ascontiguousarray would turn a scalar into a 1-vector)


def _shape_of(sid: int) -> Tuple[int, ...]:
    l = lib()
    nd = l.H5Sget_simple_extent_ndims(sid)
    if nd <= 0:
        return ()
    dims = (hsize_t * nd)()
    l.H5Sget_simple_extent_dims(sid, dims, None)
    return tuple(int(d) for d in dims)


def _dtype_of(tid: int) -> np.dtype:
    """numpy dtype of a file type (metadata only; variable-length strings report object like h5py)."""
    l = lib()
    cls, size = l.H5Tget_class(tid), int(l.H5Tget_size(tid))
    if cls == H5T_STRING:
        return np.dtype("O") if l.H5Tis_variable_str(tid) > 0 else np.dtype(f"S{size}")
    if cls == H5T_INTEGER:
        return np.dtype(f"{'u' if l.H5Tget_sign(tid) == H5T_SGN_NONE else 'i'}{size}")
    if cls == H5T_FLOAT:
        return np.dtype(f"f{size}")
    raise H5Error(f"unsupported HDF5 type class {cls}")


def _read(obj: int, tid: int, sid: int, reader, is_attr: bool, shape: Optional[Tuple[int, ...]] = None):
    """Read a dataset / attribute whose file type is ``tid`` and dataspace ``sid`` into numpy (``shape``: the shape of
    the selection being read when it is not the whole dataspace)."""
    l = lib()
    if shape is None:
        shape = _shape_of(sid)
    cls, size = l.H5Tget_class(tid), int(l.H5Tget_size(tid))
    n = int(np.prod(shape)) if shape else 1
    if cls == H5T_STRING and l.H5Tis_variable_str(tid) > 0:
        buf = (C.c_char_p * n)()
        mt = _chk(l.H5Tcopy(tid), "H5Tcopy")          # the file's own variable-length string type (ASCII or UTF-8)
        _chk(reader(obj, mt, buf), "read (variable-length strings)")
        vals = [(buf[i] or b"") for i in range(n)]
        l.H5Dvlen_reclaim(mt, sid, H5P_DEFAULT, buf); l.H5Tclose(mt)
        out = np.array(vals, dtype="S").reshape(shape) if shape else np.bytes_(vals[0])
        return out
    if cls == H5T_STRING:
        dt = np.dtype(f"S{size}")
        mt = _chk(l.H5Tcopy(tid), "H5Tcopy"); close = True
    elif cls == H5T_INTEGER:
        dt = np.dtype(f"{'u' if l.H5Tget_sign(tid) == H5T_SGN_NONE else 'i'}{size}"); mt, close = _types[dt.name], False
    elif cls == H5T_FLOAT:
        dt = np.dtype(f"f{size}")
        if dt.name not in _types:
            raise H5Error(f"unsupported float size {size}")
        mt, close = _types[dt.name], False
    else:
        raise H5Error(f"unsupported HDF5 type class {cls}")
    out = np.empty(shape, dtype=dt)
    _chk(reader(obj, mt, out.ctypes.data_as(C.c_void_p)), "read")
    if close:
        l.H5Tclose(mt)
    return out if shape else out[()]


class AttributeManager:
    def __init__(self, owner):
        self._owner = owner          # keeps a temporary node (f["grp"].attrs[...]) alive while its attributes are used

    @property
    def _id(self) -> int:
        return self._owner._id

    def __contains__(self, name: str) -> bool:
        return lib().H5Aexists(self._id, name.encode()) > 0

    def __setitem__(self, name: str, value) -> None:
        l = lib()
        arr = _as_array(value)
        if name in self:
            l.H5Adelete(self._id, name.encode())
        tid, close = _mem_type(arr)
        sid = _space_of(arr)
        aid = _chk(l.H5Acreate2(self._id, name.encode(), tid, sid, H5P_DEFAULT, H5P_DEFAULT), f"H5Acreate2({name})")
        rc = l.H5Awrite(aid, tid, arr.ctypes.data_as(C.c_void_p))
        l.H5Aclose(aid); l.H5Sclose(sid)
        if close:
            l.H5Tclose(tid)
        _chk(rc, f"H5Awrite({name})")

    def __getitem__(self, name: str):
        l = lib()
        aid = l.H5Aopen(self._id, name.encode(), H5P_DEFAULT)
        if aid < 0:
            raise KeyError(name)
        tid, sid = l.H5Aget_type(aid), l.H5Aget_space(aid)
        try:
            return _read(aid, tid, sid, lambda o, mt, buf: l.H5Aread(o, mt, buf), True)
        finally:
            l.H5Tclose(tid); l.H5Sclose(sid); l.H5Aclose(aid)

    def keys(self) -> List[str]:
        names: List[str] = []
        cb_t = C.CFUNCTYPE(C.c_int, hid_t, C.c_char_p, C.c_void_p, C.c_void_p)

        def cb(_loc, nm, _info, _data):
            names.append(nm.decode()); return 0
        idx = hsize_t(0)
        fn = cb_t(cb)
        lib().H5Aiterate2(self._id, 0, 0, C.byref(idx), C.cast(fn, C.c_void_p), None)   # H5_INDEX_NAME, H5_ITER_INC
        return names

    def __iter__(self) -> Iterator[str]:
        return iter(self.keys())

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def get(self, name, default=None):
        return self[name] if name in self else default


class _Node:
    def __init__(self, oid: int, closer, root=None):
        self._id, self._closer = oid, closer
        self.attrs = AttributeManager(self)
        self._root = root if root is not None else self
        if root is not None:
            root._children.append(weakref.ref(self))     # closed with the file, whatever still references them

    def _close(self):
        if self._id >= 0:
            self._closer(self._id)
            self._id = -1

    def __del__(self):
        try:
            self._close()
        except Exception:      # noqa: BLE001  interpreter teardown
            pass


class Dataset(_Node):
    def _meta(self):
        l = lib()
        tid, sid = l.H5Dget_type(self._id), l.H5Dget_space(self._id)
        return tid, sid

    def __getitem__(self, key):
        l = lib()
        tid, sid = self._meta()
        try:
            shape = _shape_of(sid)
            # a contiguous range (or one index) of the LEADING axis is read as a hyperslab: only those rows leave the file
            # -- what the batch generators do on multi-GB train_images; anything fancier reads the dataset and indexes it
            lead = key[0] if isinstance(key, tuple) and len(key) >= 1 else key
            rest = key[1:] if isinstance(key, tuple) else ()
            plain_rest = all(isinstance(k, slice) and k == slice(None) for k in rest) or rest == (Ellipsis,)
            is_var = l.H5Tget_class(tid) == H5T_STRING and l.H5Tis_variable_str(tid) > 0
            if shape and plain_rest and not is_var and isinstance(lead, (int, np.integer, slice)) and not isinstance(lead, bool):
                if isinstance(lead, slice):
                    lo, hi, st = lead.indices(shape[0]); single = False
                else:
                    i = int(lead) + (shape[0] if int(lead) < 0 else 0)
                    if not 0 <= i < shape[0]:
                        raise IndexError(f"index {int(lead)} out of range for axis 0 with size {shape[0]}")
                    lo, hi, st, single = i, i + 1, 1, True
                if st == 1:
                    cnt = max(0, hi - lo)
                    sub = (cnt,) + shape[1:]
                    if cnt == 0 or 0 in sub:
                        out = np.empty(sub, dtype=_dtype_of(tid))
                    else:
                        nd = len(shape)
                        start, count = (hsize_t * nd)(lo, *([0] * (nd - 1))), (hsize_t * nd)(*sub)
                        _chk(l.H5Sselect_hyperslab(sid, H5S_SELECT_SET, start, None, count, None), "H5Sselect_hyperslab")
                        msid = _chk(l.H5Screate_simple(nd, count, None), "H5Screate_simple")
                        try:
                            out = _read(self._id, tid, sid, lambda o, mt, buf: l.H5Dread(o, mt, msid, sid, H5P_DEFAULT, buf), False, shape=sub)
                        finally:
                            l.H5Sclose(msid)
                    return out[0] if single else out
            full = _read(self._id, tid, sid, lambda o, mt, buf: l.H5Dread(o, mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf), False)
        finally:
            l.H5Tclose(tid); l.H5Sclose(sid)
        if key == () or key is Ellipsis:
            return full
        return np.asarray(full)[key]

    @property
    def shape(self):
        """From the dataspace: no data is read."""
        l = lib()
        sid = l.H5Dget_space(self._id)
        try:
            return _shape_of(sid)
        finally:
            l.H5Sclose(sid)

    @property
    def dtype(self):
        """From the file type: no data is read."""
        l = lib()
        tid = l.H5Dget_type(self._id)
        try:
            return _dtype_of(tid)
        finally:
            l.H5Tclose(tid)

    def __len__(self):
        sh = self.shape
        if not sh:
            raise TypeError("scalar dataset has no len()")
        return sh[0]


class Group(_Node):
    def create_group(self, name: str) -> "Group":
        l = lib()
        gid = _chk(l.H5Gcreate2(self._id, name.encode(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), f"H5Gcreate2({name})")
        return Group(gid, l.H5Gclose, self._root)

    def create_dataset(self, name: str, data=None, **_ignored) -> Dataset:
        l = lib()
        arr = _as_array(data)
        tid, close = _mem_type(arr)
        sid = _space_of(arr)
        did = l.H5Dcreate2(self._id, name.encode(), tid, sid, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)
        if did < 0:
            l.H5Sclose(sid)
            raise H5Error(f"HDF5: H5Dcreate2({name}) failed (name exists?)")
        rc = l.H5Dwrite(did, tid, H5S_ALL, H5S_ALL, H5P_DEFAULT, arr.ctypes.data_as(C.c_void_p)) if arr.size else 0
        l.H5Sclose(sid)
        if close:
            l.H5Tclose(tid)
        _chk(rc, f"H5Dwrite({name})")
        return Dataset(did, l.H5Dclose, self._root)

    def __contains__(self, name: str) -> bool:
        l = lib()
        cur = ""
        for part in [p for p in name.split("/") if p]:       # H5Lexists needs every intermediate link to exist
            cur = part if not cur else cur + "/" + part
            if l.H5Lexists(self._id, cur.encode(), H5P_DEFAULT) <= 0:
                return False
        return True

    def __getitem__(self, name: str):
        l = lib()
        if name not in self:
            raise KeyError(name)
        oid = _chk(l.H5Oopen(self._id, name.encode(), H5P_DEFAULT), f"H5Oopen({name})")
        kind = l.H5Iget_type(oid)
        if kind == H5I_DATASET:
            return Dataset(oid, l.H5Oclose, self._root)
        if kind == H5I_GROUP:
            return Group(oid, l.H5Oclose, self._root)
        l.H5Oclose(oid)
        raise H5Error(f"{name}: unsupported object type {kind}")

    def keys(self) -> List[str]:
        l = lib()

        class Info(C.Structure):
            _fields_ = [("storage_type", C.c_int), ("nlinks", hsize_t), ("max_corder", C.c_int64), ("mounted", C.c_int)]
        info = Info()
        _chk(l.H5Gget_info(self._id, C.byref(info)), "H5Gget_info")
        out = []
        for i in range(int(info.nlinks)):
            n = l.H5Lget_name_by_idx(self._id, b".", 0, 0, i, None, 0, H5P_DEFAULT)
            buf = C.create_string_buffer(int(n) + 1)
            l.H5Lget_name_by_idx(self._id, b".", 0, 0, i, buf, int(n) + 1, H5P_DEFAULT)
            out.append(buf.value.decode())
        return out

    def __iter__(self):
        return iter(self.keys())

    def get(self, name, default=None):
        return self[name] if name in self else default


class File(Group):
    def __init__(self, path, mode: str = "r"):
        l = lib()
        p = os.fspath(path).encode()
        if mode == "r":
            fid = l.H5Fopen(p, H5F_ACC_RDONLY, H5P_DEFAULT)
        elif mode == "w":
            fid = l.H5Fcreate(p, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
        elif mode in ("a", "r+"):
            fid = l.H5Fopen(p, H5F_ACC_RDWR, H5P_DEFAULT) if os.path.exists(path) else \
                l.H5Fcreate(p, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
        else:
            raise ValueError(f"mode {mode!r}")
        if fid < 0:
            raise H5Error(f"cannot open {path!r} (mode {mode}) as HDF5")
        self._children = []
        super().__init__(fid, l.H5Fclose)

    def close(self):
        for ref in self._children:
            node = ref()
            if node is not None:
                node._close()
        self._children = []
        self._close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
