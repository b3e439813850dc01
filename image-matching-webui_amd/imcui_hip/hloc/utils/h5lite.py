"""The subset of the h5py API the hloc feature / match files use, on the HDF5 C library through ctypes.

The reference stores features and matches in HDF5 (`imcui/hloc/extract_features.py:221-243`: one group per image, a
dataset per tensor, `keypoints.attrs["uncertainty"]`; `imcui/hloc/match_features.py:73-83`: group `name0/name1` with
`matches0` int16 / `matching_scores0` float16; readers in `utils/io.py`, `list_h5_names` in `utils/parsers.py`).  `h5py`
is the reference's dependency but it is not installed in every environment the HIP backend runs in; the C library often
is (`libhdf5.so`).  `open_h5()` returns `h5py.File` when h5py imports and this module's `File` otherwise, so the batch
drivers write REAL HDF5 files either way (checked against h5py in tests/test_h5_files_cpu.py).  No library at all ->
ImportError: there is no silent alternative format.

Covered: File(path, "r" | "a" | "w"), groups (`create_group` with intermediate groups, `in`, `[]`, `del`, `keys`,
`items`, `visititems`), datasets (`create_dataset(name, data=ndarray)`, `[()]`, `__array__`, `shape`, `dtype`) of
float16/32/64, int8/16/32/64, uint8, and float scalar attributes.
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
import glob
import os
import threading

import numpy as np

hid_t = C.c_int64
herr_t = C.c_int
hsize_t = C.c_uint64
_lock = threading.RLock()
_lib = None

H5F_ACC_RDONLY, H5F_ACC_RDWR, H5F_ACC_TRUNC, H5F_ACC_EXCL = 0, 1, 2, 4
H5P_DEFAULT, H5S_ALL = 0, 0
H5T_INTEGER, H5T_FLOAT = 0, 1
H5T_SGN_NONE = 0
H5_INDEX_NAME, H5_ITER_INC = 0, 0


def _find_library() -> str:
    cands = []
    if os.environ.get("IMCUI_HDF5_LIB"):
        cands.append(os.environ["IMCUI_HDF5_LIB"])
    found = ctypes.util.find_library("hdf5") or ctypes.util.find_library("hdf5_serial")
    if found:
        cands.append(found)
    for pat in ("/usr/lib/x86_64-linux-gnu/libhdf5_serial.so*", "/usr/lib/x86_64-linux-gnu/libhdf5.so*", "/usr/lib64/libhdf5.so*",
                "/usr/local/lib/libhdf5.so*", "/opt/conda/lib/libhdf5.so*"):  # fmt: skip
        cands.extend(sorted(glob.glob(pat)))
    for c in cands:
        try:
            C.CDLL(c)
            return c
        except OSError:
            continue
    raise ImportError("neither h5py nor the HDF5 C library (libhdf5.so; set IMCUI_HDF5_LIB) is available: cannot read or write hloc .h5 files")


def _load():
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        lib = C.CDLL(_find_library())
        sig = {
            "H5open": (herr_t, []),
            "H5Eset_auto2": (herr_t, [hid_t, C.c_void_p, C.c_void_p]),
            "H5Fcreate": (hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]),
            "H5Fopen": (hid_t, [C.c_char_p, C.c_uint, hid_t]),
            "H5Fclose": (herr_t, [hid_t]),
            "H5Fflush": (herr_t, [hid_t, C.c_int]),
            "H5Gcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t]),
            "H5Gopen2": (hid_t, [hid_t, C.c_char_p, hid_t]),
            "H5Gclose": (herr_t, [hid_t]),
            "H5Lexists": (C.c_int, [hid_t, C.c_char_p, hid_t]),
            "H5Ldelete": (herr_t, [hid_t, C.c_char_p, hid_t]),
            "H5Literate": (herr_t, [hid_t, C.c_int, C.c_int, C.POINTER(hsize_t), C.c_void_p, C.c_void_p]),
            "H5Pcreate": (hid_t, [hid_t]),
            "H5Pclose": (herr_t, [hid_t]),
            "H5Pset_create_intermediate_group": (herr_t, [hid_t, C.c_uint]),
            "H5Pset_fclose_degree": (herr_t, [hid_t, C.c_int]),
            "H5Screate_simple": (hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
            "H5Screate": (hid_t, [C.c_int]),
            "H5Sclose": (herr_t, [hid_t]),
            "H5Sget_simple_extent_ndims": (C.c_int, [hid_t]),
            "H5Sget_simple_extent_dims": (C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
            "H5Dcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
            "H5Dopen2": (hid_t, [hid_t, C.c_char_p, hid_t]),
            "H5Dclose": (herr_t, [hid_t]),
            "H5Dwrite": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
            "H5Dread": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
            "H5Dget_space": (hid_t, [hid_t]),
            "H5Dget_type": (hid_t, [hid_t]),
            "H5Tcopy": (hid_t, [hid_t]),
            "H5Tclose": (herr_t, [hid_t]),
            "H5Tget_class": (C.c_int, [hid_t]),
            "H5Tget_size": (C.c_size_t, [hid_t]),
            "H5Tget_sign": (C.c_int, [hid_t]),
            "H5Tset_fields": (herr_t, [hid_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t]),
            "H5Tset_size": (herr_t, [hid_t, C.c_size_t]),
            "H5Tset_precision": (herr_t, [hid_t, C.c_size_t]),
            "H5Tset_offset": (herr_t, [hid_t, C.c_size_t]),
            "H5Tset_ebias": (herr_t, [hid_t, C.c_size_t]),
            "H5Acreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t]),
            "H5Aopen": (hid_t, [hid_t, C.c_char_p, hid_t]),
            "H5Aexists": (C.c_int, [hid_t, C.c_char_p]),
            "H5Adelete": (herr_t, [hid_t, C.c_char_p]),
            "H5Awrite": (herr_t, [hid_t, hid_t, C.c_void_p]),
            "H5Aread": (herr_t, [hid_t, hid_t, C.c_void_p]),
            "H5Aclose": (herr_t, [hid_t]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.H5open() < 0:
            raise ImportError("H5open failed")
        lib.H5Eset_auto2(0, None, None)  # probing with H5Gopen2 / H5Lexists must not print error stacks
        _lib = lib
        return lib


def _global(name: str) -> int:
    return hid_t.in_dll(_load(), name).value


def _type_for(dtype: np.dtype):
    """(memory/file type id, owned) for a numpy dtype (little-endian hosts)."""
    lib = _load()
    dt = np.dtype(dtype)
    table = {
        "float32": "H5T_NATIVE_FLOAT_g", "float64": "H5T_NATIVE_DOUBLE_g", "int8": "H5T_NATIVE_INT8_g", "uint8": "H5T_NATIVE_UINT8_g",
        "int16": "H5T_NATIVE_INT16_g", "int32": "H5T_NATIVE_INT32_g", "int64": "H5T_NATIVE_INT64_g", "uint16": "H5T_NATIVE_UINT16_g",
        "uint32": "H5T_NATIVE_UINT32_g", "uint64": "H5T_NATIVE_UINT64_g",
    }  # fmt: skip
    if dt.name in table:
        return _global(table[dt.name]), False
    if dt.name == "float16":  # IEEE binary16, built the way h5py builds it
        t = lib.H5Tcopy(_global("H5T_IEEE_F32LE_g"))
        lib.H5Tset_fields(t, 15, 10, 5, 0, 10)
        lib.H5Tset_size(t, 2)
        lib.H5Tset_ebias(t, 15)
        return t, True
    raise TypeError(f"h5lite: unsupported dtype {dt}")


def _numpy_dtype(tid) -> np.dtype:
    lib = _load()
    cls, size = lib.H5Tget_class(tid), lib.H5Tget_size(tid)
    if cls == H5T_FLOAT:
        return np.dtype({2: "float16", 4: "float32", 8: "float64"}[size])
    if cls == H5T_INTEGER:
        unsigned = lib.H5Tget_sign(tid) == H5T_SGN_NONE
        return np.dtype(("uint" if unsigned else "int") + str(8 * size))
    raise TypeError(f"h5lite: unsupported HDF5 type class {cls}")


class AttributeManager:
    def __init__(self, obj):
        self._obj = obj

    def __setitem__(self, name, value):
        lib = _load()
        with _lock:
            val = np.asarray(value)
            if val.dtype.kind == "f" or val.dtype.kind in "iu":
                val = val.astype(np.float64 if val.dtype.kind == "f" else np.int64)
            if val.ndim != 0:
                raise TypeError("h5lite: only scalar attributes are supported")
            tid, owned = _type_for(val.dtype)
            bname = name.encode()
            if lib.H5Aexists(self._obj._id, bname) > 0:
                lib.H5Adelete(self._obj._id, bname)
            sid = lib.H5Screate(0)  # H5S_SCALAR
            aid = lib.H5Acreate2(self._obj._id, bname, tid, sid, H5P_DEFAULT, H5P_DEFAULT)
            if aid < 0:
                raise OSError(f"h5lite: cannot create attribute {name}")
            buf = np.ascontiguousarray(val)
            lib.H5Awrite(aid, tid, buf.ctypes.data)
            lib.H5Aclose(aid)
            lib.H5Sclose(sid)
            if owned:
                lib.H5Tclose(tid)

    def __getitem__(self, name):
        lib = _load()
        with _lock:
            aid = lib.H5Aopen(self._obj._id, name.encode(), H5P_DEFAULT)
            if aid < 0:
                raise KeyError(name)
            out = np.zeros((), dtype=np.float64)
            lib.H5Aread(aid, _global("H5T_NATIVE_DOUBLE_g"), out.ctypes.data)
            lib.H5Aclose(aid)
            return out[()]

    def __contains__(self, name):
        return _load().H5Aexists(self._obj._id, name.encode()) > 0


class Dataset:
    def __init__(self, did: int, name: str):
        self._id, self.name = did, name
        lib = _load()
        sid = lib.H5Dget_space(did)
        nd = lib.H5Sget_simple_extent_ndims(sid)
        dims = (hsize_t * max(nd, 1))()
        if nd > 0:
            lib.H5Sget_simple_extent_dims(sid, dims, None)
        lib.H5Sclose(sid)
        self.shape = tuple(int(d) for d in dims[:nd])
        tid = lib.H5Dget_type(did)
        self.dtype = _numpy_dtype(tid)
        lib.H5Tclose(tid)

    @property
    def attrs(self):
        return AttributeManager(self)

    def __array__(self, dtype=None, copy=None):
        lib = _load()
        with _lock:
            out = np.empty(self.shape, dtype=self.dtype)
            tid, owned = _type_for(self.dtype)
            if out.size and lib.H5Dread(self._id, tid, H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data) < 0:
                raise OSError(f"h5lite: cannot read {self.name}")
            if owned:
                lib.H5Tclose(tid)
        return out if dtype is None else out.astype(dtype)

    def __getitem__(self, key):
        return self.__array__()[key]

    def __len__(self):
        return self.shape[0]

    def __del__(self):
        try:
            _load().H5Dclose(self._id)
        except Exception:  # noqa: BLE001
            pass


class Group:
    def __init__(self, gid: int, name: str, owner=None):
        self._id, self.name, self._owner = gid, name, owner

    @property
    def attrs(self):
        return AttributeManager(self)

    def _child(self, name: str) -> str:
        return (self.name.rstrip("/") + "/" + name) if not name.startswith("/") else name

    def __contains__(self, name) -> bool:
        lib = _load()
        with _lock:
            # H5Lexists needs every intermediate link to exist: walk the path
            cur = ""
            for part in str(name).strip("/").split("/"):
                cur = part if not cur else cur + "/" + part
                if lib.H5Lexists(self._id, cur.encode(), H5P_DEFAULT) <= 0:
                    return False
            return True

    def __getitem__(self, name):
        lib = _load()
        with _lock:
            if name not in self:
                raise KeyError(name)
            gid = lib.H5Gopen2(self._id, name.encode(), H5P_DEFAULT)
            if gid >= 0:
                return Group(gid, self._child(name), self)
            did = lib.H5Dopen2(self._id, name.encode(), H5P_DEFAULT)
            if did < 0:
                raise KeyError(name)
            return Dataset(did, self._child(name))

    def __delitem__(self, name):
        if _load().H5Ldelete(self._id, name.encode(), H5P_DEFAULT) < 0:
            raise KeyError(name)

    def create_group(self, name: str) -> "Group":
        lib = _load()
        with _lock:
            lcpl = lib.H5Pcreate(_global("H5P_CLS_LINK_CREATE_ID_g"))
            lib.H5Pset_create_intermediate_group(lcpl, 1)
            gid = lib.H5Gcreate2(self._id, name.encode(), lcpl, H5P_DEFAULT, H5P_DEFAULT)
            lib.H5Pclose(lcpl)
            if gid < 0:
                raise ValueError(f"h5lite: cannot create group {name} (it may exist already)")
            return Group(gid, self._child(name), self)

    def create_dataset(self, name: str, data=None, **_unused) -> Dataset:
        lib = _load()
        arr = np.ascontiguousarray(np.asarray(data))
        with _lock:
            tid, owned = _type_for(arr.dtype)
            nd = arr.ndim
            if nd == 0:
                sid = lib.H5Screate(0)
            else:
                dims = (hsize_t * nd)(*arr.shape)
                sid = lib.H5Screate_simple(nd, dims, None)
            lcpl = lib.H5Pcreate(_global("H5P_CLS_LINK_CREATE_ID_g"))
            lib.H5Pset_create_intermediate_group(lcpl, 1)
            did = lib.H5Dcreate2(self._id, name.encode(), tid, sid, lcpl, H5P_DEFAULT, H5P_DEFAULT)
            lib.H5Pclose(lcpl)
            if did < 0:
                lib.H5Sclose(sid)
                raise OSError(f"h5lite: cannot create dataset {name}")
            if arr.size and lib.H5Dwrite(did, tid, H5S_ALL, H5S_ALL, H5P_DEFAULT, arr.ctypes.data) < 0:
                raise OSError(f"h5lite: cannot write dataset {name} (No space left on device?)")
            lib.H5Sclose(sid)
            if owned:
                lib.H5Tclose(tid)
            return Dataset(did, self._child(name))

    def keys(self):
        lib = _load()
        names = []
        cb_t = C.CFUNCTYPE(herr_t, hid_t, C.c_char_p, C.c_void_p, C.c_void_p)

        def cb(_g, name, _info, _data):
            names.append(name.decode())
            return 0

        with _lock:
            idx = hsize_t(0)
            fn = cb_t(cb)
            lib.H5Literate(self._id, H5_INDEX_NAME, H5_ITER_INC, C.byref(idx), C.cast(fn, C.c_void_p), None)
        return names

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self.keys())

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def visititems(self, func):
        """Depth-first over all members with paths relative to this group, like h5py (stops when func returns a value)."""

        def walk(grp, prefix):
            for k in grp.keys():
                obj = grp[k]
                path = prefix + k
                r = func(path, obj)
                if r is not None:
                    return r
                if isinstance(obj, Group):
                    r = walk(obj, path + "/")
                    if r is not None:
                        return r
            return None

        return walk(self, "")

    def __del__(self):
        if type(self) is Group:
            try:
                _load().H5Gclose(self._id)
            except Exception:  # noqa: BLE001
                pass


class File(Group):
    def __init__(self, path, mode: str = "r", libver=None):
        lib = _load()
        p = os.fspath(path).encode()
        with _lock:
            # H5F_CLOSE_STRONG: closing the file closes every group / dataset handle still open on it, so the file (and
            # its lock) is released at `close()` / the end of a `with` block, like h5py does
            fapl = lib.H5Pcreate(_global("H5P_CLS_FILE_ACCESS_ID_g"))
            lib.H5Pset_fclose_degree(fapl, 3)
            if mode == "r":
                fid = lib.H5Fopen(p, H5F_ACC_RDONLY, fapl)
            elif mode == "w":
                fid = lib.H5Fcreate(p, H5F_ACC_TRUNC, H5P_DEFAULT, fapl)
            elif mode == "a":
                fid = lib.H5Fopen(p, H5F_ACC_RDWR, fapl) if os.path.exists(path) else lib.H5Fcreate(p, H5F_ACC_EXCL, H5P_DEFAULT, fapl)
            else:
                lib.H5Pclose(fapl)
                raise ValueError(f"h5lite: unsupported mode {mode!r}")
            lib.H5Pclose(fapl)
        if fid < 0:
            raise OSError(f"h5lite: cannot open {path} (mode {mode})")
        super().__init__(fid, "/")
        self.filename = os.fspath(path)

    def close(self):
        if self._id >= 0:
            _load().H5Fclose(self._id)
            self._id = -1

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def open_h5(path, mode: str = "r"):
    """`h5py.File(path, mode, libver="latest")` when h5py is installed (the reference's exact call), this module's
    File on the HDF5 C library otherwise.  Raises ImportError when neither exists."""
    try:
        import h5py

        return h5py.File(str(path), mode, libver="latest")
    except ImportError:
        return File(path, mode)
