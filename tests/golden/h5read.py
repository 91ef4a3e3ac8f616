"""Minimal HDF5 dataset reader over ctypes + libhdf5 (h5py is not installed).
Used only by make_fixtures.py in the development container."""
import ctypes as C

import numpy as np

H5F_ACC_RDONLY, H5P_DEFAULT, H5S_ALL = 0, 0, 0


class H5:
    def __init__(self, libpath="/opt/conda/lib/libhdf5.so.103"):
        self.lib = lib = C.CDLL(libpath)
        lib.H5open()
        hid = C.c_int64
        lib.H5Fopen.restype = hid
        lib.H5Fopen.argtypes = [C.c_char_p, C.c_uint, hid]
        lib.H5Dopen2.restype = hid
        lib.H5Dopen2.argtypes = [hid, C.c_char_p, hid]
        lib.H5Dget_space.restype = hid
        lib.H5Dget_space.argtypes = [hid]
        lib.H5Sget_simple_extent_ndims.argtypes = [hid]
        lib.H5Sget_simple_extent_dims.argtypes = [hid, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        lib.H5Dread.argtypes = [hid, hid, hid, hid, hid, C.c_void_p]
        lib.H5Dclose.argtypes = [hid]
        lib.H5Fclose.argtypes = [hid]
        self.f64 = hid.in_dll(lib, "H5T_NATIVE_DOUBLE_g").value
        self.i64 = hid.in_dll(lib, "H5T_NATIVE_INT64_g").value

    def read(self, path, dataset, dtype=np.float64):
        lib = self.lib
        f = lib.H5Fopen(str(path).encode(), H5F_ACC_RDONLY, H5P_DEFAULT)
        if f < 0:
            raise IOError(f"cannot open {path}")
        d = lib.H5Dopen2(f, dataset.encode(), H5P_DEFAULT)
        if d < 0:
            lib.H5Fclose(f)
            raise KeyError(dataset)
        s = lib.H5Dget_space(d)
        nd = lib.H5Sget_simple_extent_ndims(s)
        dims = (C.c_uint64 * nd)()
        lib.H5Sget_simple_extent_dims(s, dims, None)
        out = np.empty(tuple(dims), dtype=dtype)
        mem = self.f64 if dtype == np.float64 else self.i64
        rc = lib.H5Dread(d, mem, H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data_as(C.c_void_p))
        lib.H5Dclose(d)
        lib.H5Fclose(f)
        if rc < 0:
            raise IOError(f"H5Dread failed for {dataset}")
        return out
