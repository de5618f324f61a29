"""Host mirror of the reference's L-BFGS surface (gcopter/lbfgs.hpp): the parameter struct, the
return codes and lbfgs_strerror keep the reference's names and values; lbfgs_optimize itself is a
host-callback API and has exactly one caller in the reference (firi::maxVolInsEllipsoid with
costMVIE, firi.hpp:207-227) -- the batched entry points below run that objective, and the MINCO
trajectory cost, entirely on the GPU."""
import ctypes
import numpy as np

from .context import default_context
from ._lib import LbfgsParams, load

# lbfgs.hpp:135-184
LBFGS_CONVERGENCE = 0
LBFGS_STOP = 1
LBFGS_CANCELED = 2
LBFGSERR_UNKNOWNERROR = -1024
LBFGSERR_INVALID_N = -1023
LBFGSERR_INVALID_MEMSIZE = -1022
LBFGSERR_INVALID_GEPSILON = -1021
LBFGSERR_INVALID_TESTPERIOD = -1020
LBFGSERR_INVALID_DELTA = -1019
LBFGSERR_INVALID_MINSTEP = -1018
LBFGSERR_INVALID_MAXSTEP = -1017
LBFGSERR_INVALID_FDECCOEFF = -1016
LBFGSERR_INVALID_SCURVCOEFF = -1015
LBFGSERR_INVALID_MACHINEPREC = -1014
LBFGSERR_INVALID_MAXLINESEARCH = -1013
LBFGSERR_INVALID_FUNCVAL = -1012
LBFGSERR_MINIMUMSTEP = -1011
LBFGSERR_MAXIMUMSTEP = -1010
LBFGSERR_MAXIMUMLINESEARCH = -1009
LBFGSERR_MAXIMUMITERATION = -1008
LBFGSERR_WIDTHTOOSMALL = -1007
LBFGSERR_INVALIDPARAMETERS = -1006
LBFGSERR_INCREASEGRADIENT = -1005
LBFGS_RUNNING = 2147483647

OPT_WAYPOINTS = 1
OPT_TIMES = 2
OPT_LOCKSTEP = 4      # force the launch-per-evaluation kernels (default: one launch, one wave per problem, when it fits)


def lbfgs_parameter_t(**over):
    """lbfgs::lbfgs_parameter_t with the reference's defaults (lbfgs.hpp:15-129)."""
    p = LbfgsParams()
    load().anet_lbfgs_default_params(ctypes.byref(p))
    for k, v in over.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def lbfgs_strerror(code):
    return load().anet_lbfgs_strerror(int(code)).decode()


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def lbfgs_mvie(A, x0, smooth_eps=1.0e-2, penalty_wt=1.0e3, param=None, max_evals=4000, ctx=None):
    """Batched firi::maxVolInsEllipsoid inner optimisation (firi.hpp:202-227).
    A: (B, M, 3) rows of A (a.x <= 1 form, firi.hpp:198-200); x0: (B, 9).
    Default parameters are the call site's (firi.hpp:212-217) when param is None.
    Returns x (B,9), f (B,), status (B,), iters (B,), evals (B,)."""
    ctx = ctx or default_context()
    if param is None:
        param = lbfgs_parameter_t(mem_size=18, g_epsilon=0.0, min_step=1.0e-32, past=3, delta=1.0e-7)
    A = np.asarray(A, dtype=np.float64)
    B, M, _ = A.shape
    Acm = np.ascontiguousarray(np.transpose(A, (0, 2, 1)))       # per problem column-major M x 3
    x = np.array(x0, dtype=np.float64).reshape(B, 9).copy()
    f = np.empty(B)
    status = np.empty(B, dtype=np.int32); iters = np.empty(B, dtype=np.int32); evals = np.empty(B, dtype=np.int32)
    ctx.check(ctx.lib.anet_lbfgs_mvie(ctx.handle, B, M, _ptr(Acm), float(smooth_eps), float(penalty_wt), _ptr(x),
                                      _ptr(f), ctypes.cast(ctypes.pointer(param), ctypes.c_void_p), int(max_evals),
                                      _ptr(status), _ptr(iters), _ptr(evals)))
    return x, f, status, iters, evals


def lbfgs_minco(head, tail, wps, T, s, hpolys=None, penalty=None, param=None, opt=OPT_WAYPOINTS | OPT_TIMES,
                max_evals=2000, want_coeffs=True, ctx=None, min_duration=0.0):
    """Batched spatial-temporal trajectory optimisation: L-BFGS on the MINCO cost
    (anet_lbfgs_minco).  Returns dict(wps, T, cost, coeffs, status, iters, evals, wide_spread); wide_spread[b] marks the
    problems whose optimised durations spread over more than 50 (coefficients re-solved by the pivoted collocation solve).
    min_duration > 0: lbfgs_optimize's step bound (lbfgs.hpp:557-565) with the built-in minimum-duration bound -- no line
    search leaves T_i >= min_duration (anet_lbfgs_minco_bounded)."""
    ctx = ctx or default_context()
    param = param or lbfgs_parameter_t()
    head = np.ascontiguousarray(head, dtype=np.float64)
    B, _, c = head.shape
    tail = np.ascontiguousarray(tail, dtype=np.float64)
    T = np.array(T, dtype=np.float64).copy()
    N = T.shape[1]
    wps = np.array(wps if wps is not None else np.zeros((B, 0, 3)), dtype=np.float64).reshape(B, N - 1, 3).copy()
    if hpolys is not None:
        hpolys = np.ascontiguousarray(hpolys, dtype=np.float64)
        if penalty is None or hpolys.shape != (B, N, penalty.poly_rows, 4):
            raise ValueError("hpolys must be (B, N, penalty.poly_rows, 4)")
    cost = np.empty(B)
    coeffs = np.empty((B, N, 3, 2 * s)) if want_coeffs else None
    status = np.empty(B, dtype=np.int32); iters = np.empty(B, dtype=np.int32); evals = np.empty(B, dtype=np.int32)
    ctx.check(ctx.lib.anet_lbfgs_minco_bounded(
        ctx.handle, s, c, N, B, _ptr(head), _ptr(tail), _ptr(wps), _ptr(T), _ptr(hpolys),
        ctypes.cast(ctypes.pointer(penalty), ctypes.c_void_p) if penalty is not None else None,
        ctypes.cast(ctypes.pointer(param), ctypes.c_void_p), int(opt), int(max_evals), float(min_duration), _ptr(cost),
        _ptr(coeffs), _ptr(status), _ptr(iters), _ptr(evals)))
    # problems whose optimised durations spread over more than 50: their returned coefficients come from the pivoted
    # collocation solve (the cost and gradients inside the loop keep the reduced system's accuracy envelope)
    wide = T.max(axis=1) > 50.0 * T.min(axis=1)
    return dict(wps=wps, T=T, cost=cost, coeffs=coeffs, status=status, iters=iters, evals=evals, wide_spread=wide)


def launch_order_from_counts(evals):
    """Longest first: the launch order for `lbfgs_minco_dev` from the evaluation counts of a previous solve of the same
    or a similar batch (int32 CUDA tensor in, int32 CUDA permutation out)."""
    import torch
    return torch.argsort(evals, descending=True, stable=True).to(torch.int32).contiguous()


def launch_order_from_counts_dev(evals, stream=None, ctx=None, fine=False):
    """The same through the library's own counting sort (anet_launch_order_from_counts_dev: what a C / C++ caller uses;
    buckets of 16 evaluations, order inside a bucket unspecified).  fine=True: buckets of one (anet_launch_order_from_steps_dev),
    for the Newton-step counts of `qp_solve_dev`."""
    import torch
    ctx = ctx or default_context(evals.device.index or 0)
    if not (evals.is_cuda and evals.dtype == torch.int32 and evals.is_contiguous() and evals.dim() == 1):
        raise ValueError("evals: contiguous int32 CUDA vector")
    order = torch.empty_like(evals)
    work = torch.empty(4096, device=evals.device, dtype=torch.int32)
    st = stream if stream is not None else torch.cuda.current_stream(evals.device).cuda_stream
    fn = ctx.lib.anet_launch_order_from_steps_dev if fine else ctx.lib.anet_launch_order_from_counts_dev
    ctx.check(fn(ctx.handle, evals.numel(), ctypes.c_void_p(evals.data_ptr()),
                                                        ctypes.c_void_p(order.data_ptr()), ctypes.c_void_p(work.data_ptr()),
                                                        ctypes.c_void_p(st)))
    return order


def lbfgs_minco_dev(head, tail, wps, T, s, c, N, B, hpolys=None, penalty=None, param=None,
                    opt=OPT_WAYPOINTS | OPT_TIMES, max_evals=2000, coeffs=None, stream=None, ctx=None, launch_order=None,
                    min_duration=0.0, return_work=False):
    """Device entry point -> anet_lbfgs_minco_[ordered_]dev.  torch CUDA float64 tensors, batch-minor, common row
    stride; wps and T are updated in place.  Returns dict(cost, status, iters, evals) of device tensors.
    launch_order: optional int32 CUDA tensor (B,), a permutation -- the problem each successive workgroup of the
    one-launch shape takes (`launch_order_from_counts(previous_evals)` when re-solving a similar batch); results
    do not depend on it, the run time does.  min_duration: as in `lbfgs_minco`."""
    import torch
    ctx = ctx or default_context(T.device.index or 0)
    param = param or lbfgs_parameter_t()
    ld = T.stride(0)
    dev = T.device
    nwork = ctx.lib.anet_lbfgs_minco_workspace(s, N, ld, ctypes.cast(ctypes.pointer(param), ctypes.c_void_p))
    work = torch.empty(nwork, device=dev, dtype=torch.float64)
    cost = torch.empty(ld, device=dev, dtype=torch.float64)
    status = torch.empty(ld, device=dev, dtype=torch.int32)
    iters = torch.empty(ld, device=dev, dtype=torch.int32)
    evals = torch.empty(ld, device=dev, dtype=torch.int32)
    if stream is None:
        stream = torch.cuda.current_stream(dev).cuda_stream
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    if launch_order is not None and not (launch_order.is_cuda and launch_order.dtype == torch.int32 and
                                         launch_order.is_contiguous() and launch_order.shape == (B,)):
        raise ValueError("launch_order: contiguous int32 CUDA tensor of shape (B,)")
    ctx.check(ctx.lib.anet_lbfgs_minco_bounded_dev(
        ctx.handle, s, c, N, B, ld, p(head), p(tail), p(wps) if N > 1 else None, p(T), p(hpolys),
        ctypes.cast(ctypes.pointer(penalty), ctypes.c_void_p) if penalty is not None else None,
        ctypes.cast(ctypes.pointer(param), ctypes.c_void_p), int(opt), int(max_evals), float(min_duration), p(launch_order),
        p(work), p(cost),
        p(coeffs), p(status), p(iters), p(evals), ctypes.c_void_p(stream)))
    wide = torch.empty(ld, device=dev, dtype=torch.int32)
    ctx.check(ctx.lib.anet_minco_spread_flags_dev(ctx.handle, N, B, ld, p(T), 0.0, p(wide), ctypes.c_void_p(stream)))
    out = dict(cost=cost[:B], status=status[:B], iters=iters[:B], evals=evals[:B], wide_spread=wide[:B])
    if return_work:  # (the workspace, for probes of what the two-launch form parks: tools/lbfgs_split_features.py)
        out["_work"] = work
    return out


_EVAL_T = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                           ctypes.c_int64, ctypes.c_int, ctypes.c_void_p)


def lbfgs_optimize_dev(x, evaluate, batch=None, param=None, max_evals=2000, bound_from=None, bound_min=0.0, ctx=None):
    """lbfgs::lbfgs_optimize (lbfgs.hpp:434-717) for an objective the caller evaluates on the device -> anet_lbfgs_optimize_dev.

    x: torch CUDA float64 tensor (n, ld), batch-minor (variable i of problem b at x[i, b]), start point in, result out.
    evaluate(x, f, g): called once per evaluation step of the batch; must fill f (ld,) and g (n, ld) -- torch tensors owned by
    this call -- with the objective and its gradient at x for every problem, using torch operations on the CURRENT stream
    (values of problems that have stopped are ignored).  bound_from / bound_min: lbfgs_optimize's step bound as the built-in
    lower bound on the variables i >= bound_from (None: no bound).  Returns dict(f, status, iters, evals) of device tensors."""
    import torch
    ctx = ctx or default_context(x.device.index or 0)
    param = param or lbfgs_parameter_t()
    if not (x.is_cuda and x.dtype == torch.float64 and x.dim() == 2 and x.stride(1) == 1):
        raise ValueError("x: CUDA float64 tensor (n, ld), batch-minor")
    n, ld = x.shape[0], x.stride(0)
    B = int(batch) if batch is not None else x.shape[1]
    dev = x.device
    f = torch.zeros(ld, device=dev, dtype=torch.float64)
    g = torch.zeros(n, ld, device=dev, dtype=torch.float64)
    work = torch.empty(ctx.lib.anet_lbfgs_workspace(n, ld, ctypes.cast(ctypes.pointer(param), ctypes.c_void_p)), device=dev,
                       dtype=torch.float64)
    status = torch.empty(ld, device=dev, dtype=torch.int32)
    iters = torch.empty(ld, device=dev, dtype=torch.int32)
    evals = torch.empty(ld, device=dev, dtype=torch.int32)
    err = []

    def _cb(_inst, _x, _f, _g, _b, _ld, _n, _stream):
        try:
            evaluate(x, f, g)
            return 0
        except Exception as exc:          # (an exception must not unwind through the C frames)
            err.append(exc)
            return 1
    cb = _EVAL_T(_cb)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    stream = torch.cuda.current_stream(dev).cuda_stream
    rc = ctx.lib.anet_lbfgs_optimize_dev(ctx.handle, n, B, ld, p(x), p(f), p(g), ctypes.cast(cb, ctypes.c_void_p), None,
                                         ctypes.cast(ctypes.pointer(param), ctypes.c_void_p), int(max_evals),
                                         n if bound_from is None else int(bound_from), float(bound_min), p(work), p(status),
                                         p(iters), p(evals), ctypes.c_void_p(stream))
    if err:
        raise err[0]
    ctx.check(rc)
    return dict(f=f[:B], status=status[:B], iters=iters[:B], evals=evals[:B])


_HOST_EVAL_T = ctypes.CFUNCTYPE(ctypes.c_double, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                ctypes.c_int)
_HOST_BOUND_T = ctypes.CFUNCTYPE(ctypes.c_double, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                 ctypes.c_int)
_HOST_PROGRESS_T = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                    ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int)


def lbfgs_optimize(x0, evaluate, stepbound=None, progress=None, param=None, ctx=None):
    """lbfgs::lbfgs_optimize (lbfgs.hpp:434-717) with HOST callbacks -> anet_lbfgs_optimize_host: one problem, the objective
    evaluated by `evaluate(x) -> (f, g)` on the host, `stepbound(xp, d) -> float` at the entry of every line search
    (lbfgs.hpp:557-565), `progress(x, g, fx, step, k, ls) -> int` after every successful one (lbfgs.hpp:580-587; non-zero
    cancels); the optimiser's vectors and arithmetic stay on the device.  Returns (ret, x, f, iters, evals)."""
    ctx = ctx or default_context()
    param = param or lbfgs_parameter_t()
    x = np.array(x0, dtype=np.float64).ravel()
    n = x.size
    err = []
    arr = lambda p: np.ctypeslib.as_array(p, shape=(n,))

    def _ev(_inst, xp, gp, _n):
        try:
            f, g = evaluate(arr(xp).copy())
            arr(gp)[:] = g
            return float(f)
        except Exception as exc:          # (an exception must not unwind through the C frames)
            err.append(exc)
            return float("nan")

    def _sb(_inst, xp, dp, _n):
        try:
            return float(stepbound(arr(xp).copy(), arr(dp).copy()))
        except Exception as exc:
            err.append(exc)
            return 0.0

    def _pg(_inst, xp, gp, fx, step, k, ls, _n):
        try:
            return int(bool(progress(arr(xp).copy(), arr(gp).copy(), fx, step, k, ls)))
        except Exception as exc:
            err.append(exc)
            return 1
    ev, sb, pg = _HOST_EVAL_T(_ev), _HOST_BOUND_T(_sb), _HOST_PROGRESS_T(_pg)
    f = ctypes.c_double(0.0)
    ret, it, nev = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0)
    vp = lambda o: ctypes.cast(o, ctypes.c_void_p)
    rc = ctx.lib.anet_lbfgs_optimize_host(ctx.handle, n, x.ctypes.data_as(ctypes.c_void_p), vp(ctypes.pointer(f)), vp(ev),
                                          vp(sb) if stepbound else None, vp(pg) if progress else None, None,
                                          vp(ctypes.pointer(param)), vp(ctypes.pointer(ret)), vp(ctypes.pointer(it)),
                                          vp(ctypes.pointer(nev)))
    if err:
        raise err[0]
    ctx.check(rc)
    return ret.value, x, f.value, it.value, nev.value
