"""TEST INFRASTRUCTURE ONLY.  ctypes door to oracle/overlap_ref.c (built by oracle/Makefile)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboverlap_ref.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            raise RuntimeError(f"{_LIB} missing: run `make -C oracle`")
        L = C.CDLL(_LIB)
        L.qko_gram_pairs.restype = C.c_int
        L.qko_gram_pairs.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        _lib = L
    return _lib


def _table(states):
    dims = np.ascontiguousarray([[1] + [t.shape[2] for t in s] for s in states], dtype=np.int32)
    keep = [[np.ascontiguousarray(t, dtype=np.complex128) for t in s] for s in states]
    flat = [t.ctypes.data for ts in keep for t in ts]
    return dims, keep, (C.c_void_p * len(flat))(*flat)


def gram_pairs(xs, ys, pairs, threads=0):
    """xs, ys: lists of site-tensor lists [chi_l, 2, chi_r]; ys=None means Y is X.
    Returns (values, z, threads_used) for pairs[t] = (i, j)."""
    pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
    xd, xk, xp = _table(xs)
    if ys is None:
        yd, yp, ny = None, None, len(xs)
    else:
        yd, yk, yp = _table(ys)
        ny = len(ys)
    n = pairs.shape[0]
    vals = np.empty(n)
    z = np.empty((n, 2))
    used = lib().qko_gram_pairs(
        xd.shape[1] - 1, len(xs), xd.ctypes.data, xp, ny, None if yd is None else yd.ctypes.data, yp, n,
        pairs.ctypes.data, vals.ctypes.data, z.ctypes.data, int(threads),
    )
    if used < 0:
        raise MemoryError("qko_gram_pairs")
    return vals, z[:, 0] + 1j * z[:, 1], used


# ------------------------------------------------------------------ the zgemm-based leg (oracle/overlap_blas.c)
_BLAS_LIB = os.path.join(_HERE, "_build", "liboverlap_blas.so")
_blas = None


def blas_lib():
    """liboverlap_blas.so with zgemm resolved from the OpenBLAS that scipy ships."""
    global _blas
    if _blas is None:
        import glob

        import scipy

        if not os.path.exists(_BLAS_LIB):
            raise RuntimeError(f"{_BLAS_LIB} missing: run `make -C oracle`")
        cands = sorted(glob.glob(os.path.join(os.path.dirname(scipy.__file__), "..", "scipy.libs", "libscipy_openblas*.so")))
        if not cands:
            raise RuntimeError("scipy's OpenBLAS not found")
        L = C.CDLL(_BLAS_LIB)
        L.qkob_init.restype = C.c_int
        L.qkob_init.argtypes = [C.c_char_p]
        L.qkob_gram_pairs.restype = C.c_int
        L.qkob_gram_pairs.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        rc = L.qkob_init(os.path.realpath(cands[0]).encode())
        if rc != 0:
            raise RuntimeError(f"qkob_init failed ({rc}) on {cands[0]}")
        _blas = L
    return _blas


def gram_pairs_blas(xs, ys, pairs, threads=0):
    """Same as gram_pairs, every contraction through BLAS zgemm (single-threaded BLAS, OpenMP over pairs)."""
    pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
    xd, xk, xp = _table(xs)
    if ys is None:
        yd, yp, ny = None, None, len(xs)
    else:
        yd, yk, yp = _table(ys)
        ny = len(ys)
    n = pairs.shape[0]
    vals = np.empty(n)
    z = np.empty((n, 2))
    used = blas_lib().qkob_gram_pairs(
        xd.shape[1] - 1, len(xs), xd.ctypes.data, xp, ny, None if yd is None else yd.ctypes.data, yp, n,
        pairs.ctypes.data, vals.ctypes.data, z.ctypes.data, int(threads),
    )
    if used < 0:
        raise RuntimeError(f"qkob_gram_pairs failed ({used})")
    return vals, z[:, 0] + 1j * z[:, 1], used
