"""ctypes front-end of oracle/knn_exact.c -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package never does.

The arithmetic it pins is described in knn_exact.c (restating
/root/reference/model.py:25-40 with a fixed fp32 FMA order and the
lower-index-wins tie rule).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libknn_exact.so")
_SRC = os.path.join(_HERE, "knn_exact.c")
_lib = None


def build(force=False):
    """Compile libknn_exact.so with the committed Makefile if missing/stale."""
    stale = (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(_SRC)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "libknn_exact.so"] + (["-B"] if force else []))
    return _SO


def _load():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(_SO)
        lib.knn_exact_f32.restype = ctypes.c_int
        lib.knn_exact_f32.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        lib.knn_exact_pd.restype = ctypes.c_float
        lib.knn_exact_pd.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        lib.knn_exact_threads.restype = ctypes.c_int
        _lib = lib
    return _lib


def knn_exact(x, k, return_pd=False):
    """x: array-like [B, C, N] fp32 (reference layout) -> int32 idx [B, N, k]
    (and the pd values of the selected neighbours if return_pd)."""
    x = np.ascontiguousarray(np.asarray(x, dtype=np.float32))
    if x.ndim != 3:
        raise ValueError("x must be [B, C, N]")
    B, C, N = x.shape
    if k > N:
        # the reference raises from topk: "selected index k out of range" (model.py:39)
        raise RuntimeError("selected index k out of range")
    idx = np.empty((B, N, k), dtype=np.int32)
    pd = np.empty((B, N, k), dtype=np.float32) if return_pd else None
    rc = _load().knn_exact_f32(x.ctypes.data, B, C, N, k, idx.ctypes.data,
                               pd.ctypes.data if return_pd else None)
    if rc != 0:
        raise RuntimeError("knn_exact_f32 failed (rc=%d)" % rc)
    return (idx, pd) if return_pd else idx


def pd_pair(xb, i, l):
    """pd(i, l) for one batch element xb [C, N] under the oracle arithmetic."""
    xb = np.ascontiguousarray(np.asarray(xb, dtype=np.float32))
    C, N = xb.shape
    return float(_load().knn_exact_pd(xb.ctypes.data, C, N, int(i), int(l)))


def threads():
    return int(_load().knn_exact_threads())
