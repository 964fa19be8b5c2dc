"""Minimal HDF5 access (ctypes over the HDF5 C library) for the opacity-data loader.

The reference reads its tables with h5fortran (src/radtran/clima_radtran_types_create.f90);
this image has no h5py, but it ships the HDF5 1.10 C library, which is all that is needed:
open a file, test for a dataset, get its shape/type class, read it as float64, and -- for the
tests that manufacture data directories -- write float64 datasets.

Shapes are reported in HDF5 (C) order.  The Fortran API presents the same dataset with its
dimensions reversed, so a dataset the reference declares as `log10k(ngauss, npress, ntemp, nwav)`
has C shape (nwav, ntemp, npress, ngauss) -- and the same bytes.
"""
import ctypes as C
import ctypes.util
import os

import numpy as np

_CANDIDATES = [os.environ.get("CLIMA_HDF5_LIB", ""), "/opt/conda/lib/libhdf5.so", "/opt/conda/lib/libhdf5.so.103",
               ctypes.util.find_library("hdf5") or "", "libhdf5.so", "libhdf5_serial.so"]

_H = None
hid_t = C.c_int64
H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5T_FLOAT = 1
H5P_DEFAULT = 0
H5S_ALL = 0


class H5Error(Exception):
    pass


def lib():
    global _H
    if _H is not None:
        return _H
    last = None
    for cand in _CANDIDATES:
        if not cand:
            continue
        try:
            h = C.CDLL(cand)
            break
        except OSError as e:
            last = e
    else:
        raise H5Error("the HDF5 C library was not found (set CLIMA_HDF5_LIB): %s" % last)
    h.H5open.restype = C.c_int
    if h.H5open() < 0:
        raise H5Error("H5open failed")
    sig = {
        "H5Fopen": (hid_t, [C.c_char_p, C.c_uint, hid_t]),
        "H5Fcreate": (hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]),
        "H5Fclose": (C.c_int, [hid_t]),
        "H5Fis_hdf5": (C.c_int, [C.c_char_p]),
        "H5Lexists": (C.c_int, [hid_t, C.c_char_p, hid_t]),
        "H5Dopen2": (hid_t, [hid_t, C.c_char_p, hid_t]),
        "H5Dclose": (C.c_int, [hid_t]),
        "H5Dget_space": (hid_t, [hid_t]),
        "H5Dget_type": (hid_t, [hid_t]),
        "H5Tget_class": (C.c_int, [hid_t]),
        "H5Tclose": (C.c_int, [hid_t]),
        "H5Sclose": (C.c_int, [hid_t]),
        "H5Sget_simple_extent_ndims": (C.c_int, [hid_t]),
        "H5Sget_simple_extent_dims": (C.c_int, [hid_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "H5Screate_simple": (hid_t, [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "H5Dread": (C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Dwrite": (C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Dcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
        "H5Eset_auto2": (C.c_int, [hid_t, C.c_void_p, C.c_void_p]),
    }
    for name, (res, args) in sig.items():
        f = getattr(h, name)
        f.restype, f.argtypes = res, args
    h.H5Eset_auto2(0, None, None)  # errors are reported through return codes here
    h.NATIVE_DOUBLE = hid_t.in_dll(h, "H5T_NATIVE_DOUBLE_g").value
    _H = h
    return h


def is_hdf5(path):
    return os.path.isfile(path) and lib().H5Fis_hdf5(path.encode()) > 0


class File:
    """Read-only HDF5 file: `exists`, `shape`, `is_float`, `read`."""

    def __init__(self, path):
        self._h = lib()
        self.path = path
        self._f = self._h.H5Fopen(path.encode(), H5F_ACC_RDONLY, H5P_DEFAULT)
        if self._f < 0:
            raise H5Error('Failed to read "%s".' % path)

    def close(self):
        if self._f is not None and self._f >= 0:
            self._h.H5Fclose(self._f)
        self._f = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        self.close()

    def exists(self, name):
        return self._h.H5Lexists(self._f, name.encode(), H5P_DEFAULT) > 0

    def _open(self, name):
        d = self._h.H5Dopen2(self._f, name.encode(), H5P_DEFAULT)
        if d < 0:
            raise H5Error('%s: dataset "%s" does not exist' % (self.path, name))
        return d

    def shape(self, name):
        d = self._open(name)
        try:
            sp = self._h.H5Dget_space(d)
            n = self._h.H5Sget_simple_extent_ndims(sp)
            dims = (C.c_uint64 * max(n, 1))()
            if n > 0:
                self._h.H5Sget_simple_extent_dims(sp, dims, None)
            self._h.H5Sclose(sp)
            return tuple(int(x) for x in dims[:n])
        finally:
            self._h.H5Dclose(d)

    def is_float(self, name):
        d = self._open(name)
        try:
            t = self._h.H5Dget_type(d)
            cls = self._h.H5Tget_class(t)
            self._h.H5Tclose(t)
            return cls == H5T_FLOAT
        finally:
            self._h.H5Dclose(d)

    def read(self, name):
        """Whole dataset as a C-ordered float64 array (HDF5 converts from the stored type)."""
        shp = self.shape(name)
        out = np.empty(shp if shp else (), dtype=np.float64)
        d = self._open(name)
        try:
            rc = self._h.H5Dread(d, self._h.NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data_as(C.c_void_p))
        finally:
            self._h.H5Dclose(d)
        if rc < 0:
            raise H5Error('%s: could not read "%s"' % (self.path, name))
        return out


def write(path, datasets):
    """Create `path` with one float64 dataset per item of `datasets` (name -> array, C order)."""
    h = lib()
    f = h.H5Fcreate(path.encode(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
    if f < 0:
        raise H5Error("could not create " + path)
    try:
        for name, arr in datasets.items():
            a = np.ascontiguousarray(arr, dtype=np.float64)
            dims = (C.c_uint64 * max(a.ndim, 1))(*a.shape)
            sp = h.H5Screate_simple(a.ndim, dims, None)
            d = h.H5Dcreate2(f, name.encode(), h.NATIVE_DOUBLE, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)
            if d < 0:
                raise H5Error("could not create dataset " + name)
            rc = h.H5Dwrite(d, h.NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, a.ctypes.data_as(C.c_void_p))
            h.H5Dclose(d)
            h.H5Sclose(sp)
            if rc < 0:
                raise H5Error("could not write dataset " + name)
    finally:
        h.H5Fclose(f)
