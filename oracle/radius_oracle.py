"""TEST INFRASTRUCTURE - ctypes front of oracle/radius_oracle.c (the CPU oracle of the radius-graph construction,
SURVEY.md §8 row f2).  Imported by tests/ only."""
import ctypes
import os

import numpy as np

from . import build_oracle

_lib = None


def _load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build_oracle.build())
        _lib.gpde_oracle_radius_edges.restype = ctypes.c_long
        _lib.gpde_oracle_radius_edges.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long, ctypes.c_int,
                                                  ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                                  ctypes.c_void_p, ctypes.c_long]
    return _lib


def radius_edges(x, r, y=None, reference_ties=True):
    """int64 [2, E]: edges (i -> j) with |x_i - y_j| <= r in np.where (row-major) order, in the reference's
    sklearn arithmetic (reference_ties=True) or by the exact sum of squares.  y=None: one point set (self-loops in,
    diagonal forced to 0 as sklearn does)."""
    lib = _load()
    x = np.ascontiguousarray(x, dtype=np.float64).reshape(len(x), -1)
    same = y is None
    yy = x if same else np.ascontiguousarray(y, dtype=np.float64).reshape(len(y), -1)
    args = (x.ctypes.data, len(x), yy.ctypes.data, len(yy), x.shape[1], float(r), int(same), int(not reference_ties))
    n = lib.gpde_oracle_radius_edges(*args, None, None, 0)
    src = np.empty(n, dtype=np.int64)
    dst = np.empty(n, dtype=np.int64)
    lib.gpde_oracle_radius_edges(*args, src.ctypes.data, dst.ctypes.data, n)
    return np.stack([src, dst])
