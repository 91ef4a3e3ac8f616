"""Minimal HDF5 access over ctypes + libhdf5 (h5py is not part of the image): exactly what
the XDMF checkpoint files of ``fedm.file_io.file_output`` (fedm/file_io.py:597-604) need --
nested groups, 2-D float64 / int64 datasets, listing a group, reading a dataset."""
import ctypes as C
import ctypes.util
import glob

import numpy as np

_H5F_ACC_RDONLY, _H5F_ACC_RDWR, _H5F_ACC_TRUNC = 0x0000, 0x0001, 0x0002
_H5P_DEFAULT, _H5S_ALL = 0, 0
_hid = C.c_int64


def _find_library():
    cands = [ctypes.util.find_library("hdf5")] + sorted(glob.glob("/opt/conda/lib/libhdf5.so*")) + \
        sorted(glob.glob("/usr/lib/x86_64-linux-gnu/libhdf5*.so*"))
    for c in cands:
        if not c:
            continue
        try:
            return C.CDLL(c)
        except OSError:
            continue
    raise RuntimeError("libhdf5 not found: XDMF/HDF5 checkpoints need the HDF5 C library")


class _Lib:
    _inst = None

    def __new__(cls):
        if cls._inst is None:
            self = super().__new__(cls)
            lib = self.lib = _find_library()
            lib.H5open()
            sig = {
                "H5Fopen": (_hid, [C.c_char_p, C.c_uint, _hid]),
                "H5Fcreate": (_hid, [C.c_char_p, C.c_uint, _hid, _hid]),
                "H5Fclose": (C.c_int, [_hid]),
                "H5Gcreate2": (_hid, [_hid, C.c_char_p, _hid, _hid, _hid]),
                "H5Gopen2": (_hid, [_hid, C.c_char_p, _hid]),
                "H5Gclose": (C.c_int, [_hid]),
                "H5Lexists": (C.c_int, [_hid, C.c_char_p, _hid]),
                "H5Gget_num_objs": (C.c_int, [_hid, C.POINTER(C.c_uint64)]),
                "H5Gget_objname_by_idx": (C.c_ssize_t, [_hid, C.c_uint64, C.c_char_p, C.c_size_t]),
                "H5Screate_simple": (_hid, [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
                "H5Sclose": (C.c_int, [_hid]),
                "H5Dcreate2": (_hid, [_hid, C.c_char_p, _hid, _hid, _hid, _hid, _hid]),
                "H5Dopen2": (_hid, [_hid, C.c_char_p, _hid]),
                "H5Dget_space": (_hid, [_hid]),
                "H5Sget_simple_extent_ndims": (C.c_int, [_hid]),
                "H5Sget_simple_extent_dims": (C.c_int, [_hid, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
                "H5Dwrite": (C.c_int, [_hid, _hid, _hid, _hid, _hid, C.c_void_p]),
                "H5Dread": (C.c_int, [_hid, _hid, _hid, _hid, _hid, C.c_void_p]),
                "H5Dclose": (C.c_int, [_hid]),
            }
            for name, (res, args) in sig.items():
                fn = getattr(lib, name)
                fn.restype, fn.argtypes = res, args
            self.types = {np.dtype(np.float64): _hid.in_dll(lib, "H5T_NATIVE_DOUBLE_g").value,
                          np.dtype(np.int64): _hid.in_dll(lib, "H5T_NATIVE_INT64_g").value}
            cls._inst = self
        return cls._inst


class File:
    """``File(path, 'w' | 'a' | 'r')`` with ``write(name, array)``, ``read(name)``, ``keys(group)``
    and ``__contains__``; names are absolute HDF5 paths, missing groups are created."""

    def __init__(self, path, mode="r"):
        self._l = _Lib()
        lib, p = self._l.lib, str(path).encode()
        if mode == "w":
            self.id = lib.H5Fcreate(p, _H5F_ACC_TRUNC, _H5P_DEFAULT, _H5P_DEFAULT)
        elif mode == "a":
            # append: create the file only when it does not exist -- a file that exists but cannot be
            # opened (locked, unreadable, not HDF5) is an error, never a reason to truncate a checkpoint
            import os
            if os.path.exists(path):
                self.id = lib.H5Fopen(p, _H5F_ACC_RDWR, _H5P_DEFAULT)
            else:
                self.id = lib.H5Fcreate(p, _H5F_ACC_TRUNC, _H5P_DEFAULT, _H5P_DEFAULT)
        elif mode == "r":
            self.id = lib.H5Fopen(p, _H5F_ACC_RDONLY, _H5P_DEFAULT)
        else:
            raise ValueError("mode must be 'r', 'w' or 'a'")
        if self.id < 0:
            raise IOError(f"cannot open {path}")

    def close(self):
        if self.id >= 0:
            self._l.lib.H5Fclose(self.id)
            self.id = -1

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __contains__(self, name):
        lib, cur = self._l.lib, ""
        for part in name.strip("/").split("/"):
            cur += "/" + part
            if lib.H5Lexists(self.id, cur.encode(), _H5P_DEFAULT) <= 0:
                return False
        return True

    def _ensure_groups(self, name):
        lib, cur = self._l.lib, ""
        for part in name.strip("/").split("/")[:-1]:
            cur += "/" + part
            if lib.H5Lexists(self.id, cur.encode(), _H5P_DEFAULT) <= 0:
                g = lib.H5Gcreate2(self.id, cur.encode(), _H5P_DEFAULT, _H5P_DEFAULT, _H5P_DEFAULT)
                if g < 0:
                    raise IOError(f"cannot create group {cur}")
                lib.H5Gclose(g)

    def write(self, name, array):
        lib = self._l.lib
        a = np.ascontiguousarray(array)
        if a.dtype not in self._l.types:
            a = a.astype(np.int64 if np.issubdtype(a.dtype, np.integer) else np.float64)
        self._ensure_groups(name)
        dims = (C.c_uint64 * a.ndim)(*a.shape)
        space = lib.H5Screate_simple(a.ndim, dims, None)
        t = self._l.types[a.dtype]
        d = lib.H5Dcreate2(self.id, name.encode(), t, space, _H5P_DEFAULT, _H5P_DEFAULT, _H5P_DEFAULT)
        if d < 0:
            lib.H5Sclose(space)
            raise IOError(f"cannot create dataset {name}")
        rc = lib.H5Dwrite(d, t, _H5S_ALL, _H5S_ALL, _H5P_DEFAULT, a.ctypes.data_as(C.c_void_p))
        lib.H5Dclose(d)
        lib.H5Sclose(space)
        if rc < 0:
            raise IOError(f"H5Dwrite failed for {name}")

    def read(self, name, dtype=np.float64):
        lib = self._l.lib
        d = lib.H5Dopen2(self.id, name.encode(), _H5P_DEFAULT)
        if d < 0:
            raise KeyError(name)
        s = lib.H5Dget_space(d)
        nd = lib.H5Sget_simple_extent_ndims(s)
        dims = (C.c_uint64 * max(nd, 1))()
        lib.H5Sget_simple_extent_dims(s, dims, None)
        out = np.empty(tuple(dims[:nd]), dtype=dtype)
        rc = lib.H5Dread(d, self._l.types[np.dtype(dtype)], _H5S_ALL, _H5S_ALL, _H5P_DEFAULT,
                         out.ctypes.data_as(C.c_void_p))
        lib.H5Sclose(s)
        lib.H5Dclose(d)
        if rc < 0:
            raise IOError(f"H5Dread failed for {name}")
        return out

    def keys(self, group="/"):
        lib = self._l.lib
        g = lib.H5Gopen2(self.id, group.encode(), _H5P_DEFAULT)
        if g < 0:
            raise KeyError(group)
        n = C.c_uint64()
        lib.H5Gget_num_objs(g, C.byref(n))
        names = []
        buf = C.create_string_buffer(1024)
        for i in range(n.value):
            lib.H5Gget_objname_by_idx(g, i, buf, 1024)
            names.append(buf.value.decode())
        lib.H5Gclose(g)
        return names
