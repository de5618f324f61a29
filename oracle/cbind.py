"""ctypes binding of oracle/liboracle.so (TEST INFRASTRUCTURE ONLY)."""
import ctypes
import os
import subprocess
import numpy as np
from ctypes import c_int, c_int64, c_double, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "minco_oracle.c")
    if force or not os.path.exists(_PATH) or os.path.getmtime(_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B"], check=True, capture_output=True)
    return _PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_PATH)
        L.oracle_minco_solve_batch.restype = c_int
        L.oracle_minco_solve_batch.argtypes = [c_int, c_int, c_int, c_int64] + [c_void_p] * 6 + [c_int]
        L.oracle_traj_cost.restype = c_double
        L.oracle_traj_cost.argtypes = [c_int, c_int, c_void_p, c_void_p, c_double]
        L.oracle_piece_eval.restype = None
        L.oracle_piece_eval.argtypes = [c_int, c_void_p, c_double, c_int, c_void_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(c_void_p) if a is not None else None


def minco_solve_batch(s, head, tail, wps, T, nthreads=1, want_coeffs=True):
    head = np.ascontiguousarray(head, dtype=np.float64)
    tail = np.ascontiguousarray(tail, dtype=np.float64)
    T = np.ascontiguousarray(T, dtype=np.float64)
    B, _, c = head.shape
    N = T.shape[1]
    wps = np.ascontiguousarray(wps if N > 1 else np.zeros((B, 0, 3)), dtype=np.float64)
    coeffs = np.empty((B, N, 3, 2 * s)) if want_coeffs else None
    energy = np.empty(B)
    rc = lib().oracle_minco_solve_batch(s, c, N, B, _p(head), _p(tail), _p(wps), _p(T), _p(coeffs),
                                        _p(energy), nthreads)
    if rc:
        raise RuntimeError(f"oracle_minco_solve_batch failed: {rc}")
    return coeffs, energy


def traj_cost(order, coeffs, T, m34=1400.0):
    coeffs = np.ascontiguousarray(coeffs, dtype=np.float64)
    T = np.ascontiguousarray(T, dtype=np.float64)
    return lib().oracle_traj_cost(order, len(T), _p(coeffs), _p(T), m34)


def piece_eval(cm, t, d):
    cm = np.ascontiguousarray(cm, dtype=np.float64)
    out = np.zeros(3)
    lib().oracle_piece_eval(cm.shape[1], _p(cm), float(t), int(d), _p(out))
    return out
