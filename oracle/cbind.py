"""ctypes binding of oracle/liboracle.so (TEST INFRASTRUCTURE ONLY)."""
import ctypes
import os
import subprocess
import numpy as np
from ctypes import c_int, c_int64, c_double, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "liboracle.so")
_CPU_PATH = os.path.join(_HERE, "libcpu_reduced.so")
_lib = None
_cpu_lib = None


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("minco_oracle.c", "lbfgs_oracle.c", "minco_costgrad.c", "qp_ipm_port.c",
                                             "minco_cpu_reduced.cpp")]
    srcs += [os.path.join(os.path.dirname(_HERE), "allocnet_amd", "csrc", f) for f in ("minco_core.h", "minco_tables.h")]
    outs = (_PATH, _CPU_PATH)
    stale = lambda: not all(os.path.exists(o) for o in outs) or any(os.path.getmtime(o) < os.path.getmtime(s) for o in outs for s in srcs)
    if force or stale():
        # one builder at a time: the ranks of a multi-process test all get here together when a source is newer than the libraries
        import fcntl
        with open(os.path.join(_HERE, ".build.lock"), "w") as lk:
            fcntl.flock(lk, fcntl.LOCK_EX)
            try:
                if force or stale():
                    subprocess.run(["make", "-C", _HERE, "-B", "all"], check=True, capture_output=True)
            finally:
                fcntl.flock(lk, fcntl.LOCK_UN)
    return _PATH


def cpu_reduced_solve_batch(s, head, tail, wps, T, nthreads=1, want_coeffs=True, out=None):
    """CPU BASELINE of bench.py (not an oracle): the kernels' own reduced algorithm compiled for the host
    (oracle/minco_cpu_reduced.cpp).  Same arguments and returns as minco_solve_batch."""
    global _cpu_lib
    if _cpu_lib is None:
        build()
        _cpu_lib = ctypes.CDLL(_CPU_PATH)
        _cpu_lib.cpu_reduced_minco_solve_batch.restype = c_int
        _cpu_lib.cpu_reduced_minco_solve_batch.argtypes = [c_int, c_int, c_int, c_int64] + [c_void_p] * 6 + [c_int]
    head = np.ascontiguousarray(head, dtype=np.float64)
    tail = np.ascontiguousarray(tail, dtype=np.float64)
    T = np.ascontiguousarray(T, dtype=np.float64)
    B, _, c = head.shape
    N = T.shape[1]
    wps = np.ascontiguousarray(wps if N > 1 else np.zeros((B, 0, 3)), dtype=np.float64)
    coeffs, energy = out if out is not None else ((np.empty((B, N, 3, 2 * s)) if want_coeffs else None), np.empty(B))
    rc = _cpu_lib.cpu_reduced_minco_solve_batch(s, c, N, B, _p(head), _p(tail), _p(wps), _p(T), _p(coeffs), _p(energy), nthreads)
    if rc:
        raise RuntimeError(f"cpu_reduced_minco_solve_batch failed: {rc}")
    return coeffs, energy


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
        L.oracle_lbfgs_default_param.argtypes = [ctypes.POINTER(LbfgsParam)]
        L.oracle_lbfgs_optimize.restype = c_int
        L.oracle_lbfgs_optimize.argtypes = [c_int, c_void_p, c_void_p, EVAL_T, c_void_p,
                                            ctypes.POINTER(LbfgsParam), c_void_p, c_void_p]
        L.oracle_lbfgs_mvie.restype = c_int
        L.oracle_lbfgs_mvie.argtypes = [c_int, c_void_p, c_double, c_double, c_void_p, c_void_p,
                                        ctypes.POINTER(LbfgsParam), c_void_p, c_void_p]
        L.oracle_cost_mvie.restype = c_double
        L.oracle_cost_mvie.argtypes = [c_void_p, c_void_p, c_void_p, c_int]
        L.oracle_minco_cost_grad_batch.restype = c_int
        L.oracle_minco_cost_grad_batch.argtypes = [c_int, c_int, c_int, c_int64] + [c_void_p] * 5 + \
            [ctypes.POINTER(Penalty)] + [c_void_p] * 3 + [c_int]
        L.oracle_lbfgs_minco_batch.restype = c_int
        L.oracle_lbfgs_minco_batch.argtypes = [c_int, c_int, c_int, c_int64] + [c_void_p] * 5 + \
            [ctypes.POINTER(Penalty), ctypes.POINTER(LbfgsParam)] + [c_void_p] * 4 + [c_int, c_double]
        L.oracle_qp_ipm_batch.restype = c_int
        L.oracle_qp_ipm_batch.argtypes = [c_int, c_int, c_int, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_double, c_double,
                                          c_double, c_double, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int]
        _lib = L
    return _lib


def qp_ipm_batch(s, state, T, hpolys, res=20, vmax=4.0, amax=6.0, m34=1400.0, tol=1e-8, max_iter=80, want_coeffs=True,
                 nthreads=1):
    """CPU port of the structured interior point (oracle/qp_ipm_port.c): the reference's inequality QP
    (qp_solver.hpp:119-358) in Hermite node coordinates, one problem per task.  state (B,2,3,3) [start/end][axis][p,v,a],
    T (B,N), hpolys (B,N,M,4) rows a.x <= b (zero rows = padding).  Returns dict(coeffs (B,N,3,2s) or None, obj, status
    (1 solved to tol, 2 solved to 1e-7 only -- stalled at its rounding floor --, -2 not converged / infeasible), iters)."""
    state = np.ascontiguousarray(state, dtype=np.float64)
    T = np.ascontiguousarray(T, dtype=np.float64)
    hpolys = np.ascontiguousarray(hpolys, dtype=np.float64)
    B, N = T.shape
    M = hpolys.shape[2]
    assert state.shape == (B, 2, 3, 3) and hpolys.shape == (B, N, M, 4)
    coeffs = np.zeros((B, N, 3, 2 * s)) if want_coeffs else None
    obj = np.zeros(B); status = np.zeros(B, dtype=np.int32); iters = np.zeros(B, dtype=np.int32)
    rc = lib().oracle_qp_ipm_batch(s, N, int(res), M, B, _p(state), _p(T), _p(hpolys), float(vmax), float(amax), float(m34),
                                   float(tol), int(max_iter), _p(coeffs), _p(obj), _p(status), _p(iters), int(nthreads))
    if rc:
        raise RuntimeError(f"oracle_qp_ipm_batch failed: {rc}")
    return dict(coeffs=coeffs, obj=obj, status=status, iters=iters)


class Penalty(ctypes.Structure):
    """oracle_penalty (oracle/minco_costgrad.c): weights of the penalty functional on the reference's inequality rows."""
    _fields_ = [("rho", c_double), ("wc", c_double), ("wv", c_double), ("wa", c_double), ("mu", c_double),
                ("vmax", c_double), ("amax", c_double), ("res", c_int), ("M", c_int)]


def _cg_args(head, tail, wps, T, hpolys, copy=False):
    head = np.ascontiguousarray(head, dtype=np.float64)
    tail = np.ascontiguousarray(tail, dtype=np.float64)
    B, _, c = head.shape
    T = np.array(T, dtype=np.float64, copy=True) if copy else np.ascontiguousarray(T, dtype=np.float64)
    N = T.shape[1]
    wps = np.zeros((B, 0, 3)) if N == 1 else wps
    wps = np.array(wps, dtype=np.float64, copy=True) if copy else np.ascontiguousarray(wps, dtype=np.float64)
    hpolys = None if hpolys is None else np.ascontiguousarray(hpolys, dtype=np.float64)
    return head, tail, wps, T, hpolys, B, c, N


def minco_cost_grad_batch(s, head, tail, wps, T, hpolys, rho, res, vmax, amax, wc, wv, wa, mu, nthreads=1):
    """Classic banded-LU MINCO cost + analytic gradient (oracle/minco_costgrad.c).  head/tail (B,3,c), wps (B,N-1,3),
    T (B,N), hpolys (B,N,M,4) or None.  Returns cost (B,), gradP (B,N-1,3), gradT (B,N)."""
    head, tail, wps, T, hpolys, B, c, N = _cg_args(head, tail, wps, T, hpolys)
    pen = Penalty(rho, wc, wv, wa, mu, vmax, amax, res, 0 if hpolys is None else hpolys.shape[2])
    cost = np.empty(B); gP = np.empty((B, N - 1, 3)); gT = np.empty((B, N))
    rc = lib().oracle_minco_cost_grad_batch(s, c, N, B, _p(head), _p(tail), _p(wps), _p(T), _p(hpolys), ctypes.byref(pen),
                                            _p(cost), _p(gP), _p(gT), nthreads)
    if rc:
        raise RuntimeError(f"oracle_minco_cost_grad_batch failed: {rc}")
    return cost, gP, gT


def lbfgs_minco_batch(s, head, tail, wps, T, hpolys, rho, res, vmax, amax, wc, wv, wa, mu, param=None, nthreads=1,
                      min_duration=0.0):
    """oracle_lbfgs_optimize (lbfgs.hpp:434-717 restated) on the cost above in the variables [waypoints, tau],
    T = forward_T(tau); one problem per task.  min_duration > 0: with the minimum-duration step bound as lbfgs_optimize's
    proc_stepbound (lbfgs.hpp:557-565).  Returns dict(wps, T, cost, status, iters, evals)."""
    head, tail, wps, T, hpolys, B, c, N = _cg_args(head, tail, wps, T, hpolys, copy=True)
    pen = Penalty(rho, wc, wv, wa, mu, vmax, amax, res, 0 if hpolys is None else hpolys.shape[2])
    param = param or lbfgs_default_param()
    cost = np.empty(B)
    status = np.empty(B, dtype=np.int32); iters = np.empty(B, dtype=np.int32); evals = np.empty(B, dtype=np.int32)
    rc = lib().oracle_lbfgs_minco_batch(s, c, N, B, _p(head), _p(tail), _p(wps), _p(T), _p(hpolys), ctypes.byref(pen),
                                        ctypes.byref(param), _p(cost), _p(status), _p(iters), _p(evals), nthreads,
                                        float(min_duration))
    if rc:
        raise RuntimeError(f"oracle_lbfgs_minco_batch failed: {rc}")
    return dict(wps=wps, T=T, cost=cost, status=status, iters=iters, evals=evals)


class LbfgsParam(ctypes.Structure):
    """lbfgs::lbfgs_parameter_t (lbfgs.hpp:15-129)."""
    _fields_ = [("mem_size", c_int), ("g_epsilon", c_double), ("past", c_int), ("delta", c_double),
                ("max_iterations", c_int), ("max_linesearch", c_int), ("min_step", c_double),
                ("max_step", c_double), ("f_dec_coeff", c_double), ("s_curv_coeff", c_double),
                ("cautious_factor", c_double), ("machine_prec", c_double)]


class MvieData(ctypes.Structure):
    _fields_ = [("M", c_int), ("eps", c_double), ("wt", c_double), ("A", c_void_p)]


EVAL_T = ctypes.CFUNCTYPE(c_double, c_void_p, ctypes.POINTER(c_double), ctypes.POINTER(c_double), c_int)


def lbfgs_default_param(**over):
    p = LbfgsParam()
    lib().oracle_lbfgs_default_param(ctypes.byref(p))
    for k, v in over.items():
        setattr(p, k, v)
    return p


BOUND_T = ctypes.CFUNCTYPE(c_double, c_void_p, ctypes.POINTER(c_double), ctypes.POINTER(c_double), c_int)
PROGRESS_T = ctypes.CFUNCTYPE(c_int, c_void_p, ctypes.POINTER(c_double), ctypes.POINTER(c_double), c_double, c_double, c_int, c_int,
                              c_int)


def lbfgs_optimize(x0, fun, param=None, stepbound=None, progress=None):
    """fun(x) -> (f, g); stepbound(xp, d) -> float (lbfgs.hpp:557-565); progress(x, g, fx, step, k, ls) -> int, non-zero
    cancels (lbfgs.hpp:580-587).  Returns ret, x, f, iters, evals."""
    param = param or lbfgs_default_param()
    x = np.array(x0, dtype=np.float64)
    n = x.size

    def cb(inst, xp, gp, nn):
        xv = np.ctypeslib.as_array(xp, shape=(nn,))
        f, g = fun(xv.copy())
        np.ctypeslib.as_array(gp, shape=(nn,))[:] = g
        return float(f)

    def sb(inst, xp, dp, nn):
        return float(stepbound(np.ctypeslib.as_array(xp, shape=(nn,)).copy(), np.ctypeslib.as_array(dp, shape=(nn,)).copy()))

    def pg(inst, xp, gp, fx, step, k, ls, nn):
        return int(bool(progress(np.ctypeslib.as_array(xp, shape=(nn,)).copy(), np.ctypeslib.as_array(gp, shape=(nn,)).copy(),
                                 fx, step, k, ls)))
    f = c_double(0.0); it = c_int(0); ev = c_int(0)
    L = lib()
    L.oracle_lbfgs_optimize_full.restype = c_int
    ret = L.oracle_lbfgs_optimize_full(n, _p(x), ctypes.byref(f), EVAL_T(cb), BOUND_T(sb) if stepbound else None,
                                       PROGRESS_T(pg) if progress else None, None, ctypes.byref(param), ctypes.byref(it),
                                       ctypes.byref(ev))
    return ret, x, f.value, it.value, ev.value


def cost_mvie(A, eps, wt, x):
    """A: (M,3) rows of the normalised polytope (firi.hpp:198-200)."""
    Ac = np.asfortranarray(np.asarray(A, dtype=np.float64))
    d = MvieData(Ac.shape[0], eps, wt, Ac.ctypes.data)
    x = np.ascontiguousarray(x, dtype=np.float64); g = np.zeros(9)
    f = lib().oracle_cost_mvie(ctypes.byref(d), _p(x), _p(g), 9)
    return f, g


def lbfgs_mvie(A, eps, wt, x0, param=None):
    param = param or lbfgs_default_param()
    Ac = np.asfortranarray(np.asarray(A, dtype=np.float64))
    x = np.array(x0, dtype=np.float64)
    f = c_double(0.0); it = c_int(0); ev = c_int(0)
    ret = lib().oracle_lbfgs_mvie(Ac.shape[0], Ac.ctypes.data, eps, wt, _p(x), ctypes.byref(f),
                                  ctypes.byref(param), ctypes.byref(it), ctypes.byref(ev))
    return ret, x, f.value, it.value, ev.value


def _p(a):
    return a.ctypes.data_as(c_void_p) if a is not None else None


def minco_solve_batch(s, head, tail, wps, T, nthreads=1, want_coeffs=True, out=None):
    head = np.ascontiguousarray(head, dtype=np.float64)
    tail = np.ascontiguousarray(tail, dtype=np.float64)
    T = np.ascontiguousarray(T, dtype=np.float64)
    B, _, c = head.shape
    N = T.shape[1]
    wps = np.ascontiguousarray(wps if N > 1 else np.zeros((B, 0, 3)), dtype=np.float64)
    coeffs, energy = out if out is not None else ((np.empty((B, N, 3, 2 * s)) if want_coeffs else None), np.empty(B))
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
