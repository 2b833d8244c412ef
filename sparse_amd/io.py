"""Matrix Market files in front of the COO boundary.  The reference has no
file format; this is the loader that lets a SuiteSparse matrix take the place of a synthetic one:

    n, m, (rs, cs, vs) = read_mtx("matrix.mtx")          # parsed by libsparse_amd.so (C++, threaded)
    A = mk_compressed("f64").sr.sparse(n, m, (rs, cs, vs))

Parsing is host work inside the shared library (sparse_amd/csrc/mtx.hpp); nothing here computes."""
import ctypes
import os

import numpy as np

from .compressed import load_library

_NP = {"f64": np.float64, "f32": np.float32, "i32": np.int32}


class MtxError(ValueError):
    pass


def _take_error(lib, err):
    msg = ctypes.string_at(err).decode() if err else "unknown error"
    if err:
        lib.sparse_amd_mtx_free(err)
    return msg


def read_mtx(path, dtype="f64", threads=None):
    """-> (n, m, (rows, cols, vals)): 0-based int64 indices, values as `dtype`, file order."""
    lib = load_library()
    lib.sparse_amd_mtx_free.argtypes = [ctypes.c_void_p]
    n, m, k = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    r, c, v, err = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    if threads is None:
        threads = min(16, os.cpu_count() or 1)
    rc = lib.sparse_amd_mtx_read(os.fsencode(path), ctypes.c_int(threads), ctypes.byref(n), ctypes.byref(m),
                                 ctypes.byref(k), ctypes.byref(r), ctypes.byref(c), ctypes.byref(v), ctypes.byref(err))
    if rc != 0:
        raise MtxError("%s: %s" % (path, _take_error(lib, err.value)))
    try:
        cnt = k.value
        rows = np.ctypeslib.as_array(ctypes.cast(r, ctypes.POINTER(ctypes.c_int64)), shape=(max(cnt, 1),))[:cnt].copy()
        cols = np.ctypeslib.as_array(ctypes.cast(c, ctypes.POINTER(ctypes.c_int64)), shape=(max(cnt, 1),))[:cnt].copy()
        vals = np.ctypeslib.as_array(ctypes.cast(v, ctypes.POINTER(ctypes.c_double)), shape=(max(cnt, 1),))[:cnt]
        vals = vals.astype(_NP[dtype])
    finally:
        for p in (r, c, v):
            lib.sparse_amd_mtx_free(p)
    return n.value, m.value, (rows, cols, vals)


def write_mtx(path, n, m, coo):
    """Writes `coordinate real general`; coo = (rows, cols, vals), 0-based."""
    lib = load_library()
    lib.sparse_amd_mtx_free.argtypes = [ctypes.c_void_p]
    rows = np.ascontiguousarray(coo[0], dtype=np.int64)
    cols = np.ascontiguousarray(coo[1], dtype=np.int64)
    vals = np.ascontiguousarray(coo[2], dtype=np.float64)
    assert len(rows) == len(cols) == len(vals)
    err = ctypes.c_void_p()
    rc = lib.sparse_amd_mtx_write(os.fsencode(path), ctypes.c_int64(n), ctypes.c_int64(m), ctypes.c_int64(len(rows)),
                                  rows.ctypes.data_as(ctypes.c_void_p), cols.ctypes.data_as(ctypes.c_void_p),
                                  vals.ctypes.data_as(ctypes.c_void_p), ctypes.byref(err))
    if rc != 0:
        raise MtxError("%s: %s" % (path, _take_error(lib, err.value)))
