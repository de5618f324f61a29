"""Host mirror of the reference's QP assembly: QPSolver::solve steps one to three
(src/planner/include/planner/qp_solver.hpp:119-296) and MinTrajOpt.update / fill_eq_obj / fill_ineq
(network/utils/min_traj_opt.py:68-178, 377-613), batched on the GPU (anet_qp_assemble)."""
import ctypes
import numpy as np

from .context import default_context
from ._lib import QpDims

ORDER_CPP = 0      # per sample: polytope rows then 12 box rows (qp_solver.hpp:258-294)
ORDER_PYTHON = 1   # all corridor rows (G1,h1), then all box rows (G2,h2) (min_traj_opt.py:535-613)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def qp_dims(order, rows, res, ctx=None):
    ctx = ctx or default_context()
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    d = QpDims()
    rc = ctx.lib.anet_qp_dims_of(int(order), len(rows), int(res), _p(rows), ctypes.byref(d))
    if rc:
        raise ValueError("bad QP dimensions")
    return d.n, d.m_e, d.m_g


def qp_assemble(order, iniPVA, finPVA, hPolys, times, res=20, max_vel=4.0, max_acc=6.0, m34=1400.0,
                float_time=False, row_order=ORDER_CPP, ctx=None):
    """Batched QP assembly.
    iniPVA, finPVA : (B,3,3) row = axis, cols p,v,a          (learning_planning.cpp:150-151)
    hPolys         : (B,N,M,4) rows (a,b), a.x <= b, zero rows = padding; or a list (one trajectory)
                     of N arrays (m_i,4) like the reference's std::vector<Eigen::MatrixX4d>
    times          : (B,N)
    Returns Q (B,n,n), A (B,me,n), b (B,me), G (B,mg,n), h (B,mg).
    Defaults: MaxVelBox/MaxAccBox/ConstRes of config/planner.yaml:17-21."""
    ctx = ctx or default_context()
    single = isinstance(hPolys, (list, tuple))
    if single:
        N = len(hPolys)
        M = max(1, max(p.shape[0] for p in hPolys))
        hp = np.zeros((1, N, M, 4))
        rows = np.zeros((1, N), dtype=np.int32)
        for i, p in enumerate(hPolys):
            hp[0, i, :p.shape[0]] = p
            rows[0, i] = p.shape[0]
        iniPVA = np.asarray(iniPVA, dtype=np.float64)[None]
        finPVA = np.asarray(finPVA, dtype=np.float64)[None]
        times = np.asarray(times, dtype=np.float64)[None]
    else:
        hp = np.ascontiguousarray(hPolys, dtype=np.float64)
        rows = (np.abs(hp).sum(axis=3) > 0).sum(axis=2).astype(np.int32)   # valid rows are leading
        N, M = hp.shape[1], hp.shape[2]
        # zero rows are kept as inert rows so that every trajectory has the same shape
        rows[:] = M
    B = hp.shape[0]
    state = np.ascontiguousarray(np.stack([np.asarray(iniPVA, dtype=np.float64),
                                           np.asarray(finPVA, dtype=np.float64)], axis=1))      # (B,2,3,3)
    T = np.ascontiguousarray(times, dtype=np.float64)
    if T.shape != (B, N) or state.shape != (B, 2, 3, 3):
        raise ValueError("shape mismatch")
    n, me, mg = qp_dims(order, rows[0], res, ctx)
    Q = np.empty((B, n, n)); A = np.empty((B, me, n)); b = np.empty((B, me)); G = np.empty((B, mg, n)); h = np.empty((B, mg))
    hp = np.ascontiguousarray(hp); rows = np.ascontiguousarray(rows)
    ctx.check(ctx.lib.anet_qp_assemble(ctx.handle, int(order), N, B, int(res), M, float(max_vel), float(max_acc),
                                       float(m34), int(bool(float_time)), int(row_order), _p(state), _p(T), _p(hp),
                                       _p(rows), _p(Q), _p(A), _p(b), _p(G), _p(h)))
    if single:
        return Q[0], A[0], b[0], G[0], h[0]
    return Q, A, b, G, h


# -------------------------------------------------------------------------------------------------
# the solve: QPSolver (planner/qp_solver.hpp:28-366)
# -------------------------------------------------------------------------------------------------
QP_METHOD_ADMM = 0             # OSQP's algorithm and settings (opt-in)
QP_METHOD_INTERIOR_POINT = 1   # primal-dual interior point in Hermite node coordinates (csrc/qp_ipm.h): the default


def qp_settings(**over):
    """anet_qp_default_settings: OSQP's default tolerances and iteration parameters as the reference uses them (it never
    overrides any: qp_solver.hpp:299-302, layers.py:79), method = QP_METHOD_INTERIOR_POINT; method=QP_METHOD_ADMM
    selects OSQP's own iteration."""
    from ._lib import QpSettings, load
    s = QpSettings()
    load().anet_qp_default_settings(ctypes.byref(s))
    for k, v in over.items():
        if not hasattr(s, k):
            raise AttributeError(k)
        setattr(s, k, v)
    return s


def qp_solve(order, iniPVA, finPVA, hPolys, times, res=20, max_vel=4.0, max_acc=6.0, m34=1400.0,
             settings=None, time_grad=False, ctx=None):
    """Batched QPSolver::solve.  iniPVA/finPVA (B,3,3); hPolys (B,N,M,4) rows a.x <= b (zero rows =
    padding); times (B,N).  Returns dict(coeffs (B,N,3,2s), obj (B,), status, iters, residuals (B,2));
    with time_grad=True also grad_T (B,N) = d(obj)/d(times), the derivative of the optimal cost through
    the inequality QP (anet_qp_solve_time_grad: envelope theorem on the solver's multipliers)."""
    ctx = ctx or default_context()
    hp = np.ascontiguousarray(hPolys, dtype=np.float64)
    B, N, M, _ = hp.shape
    state = np.ascontiguousarray(np.stack([np.asarray(iniPVA, dtype=np.float64),
                                           np.asarray(finPVA, dtype=np.float64)], axis=1))
    T = np.ascontiguousarray(times, dtype=np.float64)
    if state.shape != (B, 2, 3, 3) or T.shape != (B, N):
        raise ValueError("shape mismatch")
    D = 2 * order
    coeffs = np.empty((B, N, 3, D)); obj = np.empty(B)
    status = np.empty(B, dtype=np.int32); iters = np.empty(B, dtype=np.int32); resid = np.empty((B, 2))
    sp = ctypes.cast(ctypes.pointer(settings), ctypes.c_void_p) if settings is not None else None
    args = (ctx.handle, int(order), N, B, int(res), M, float(max_vel), float(max_acc), float(m34), _p(state), _p(T),
            _p(hp), sp, _p(coeffs), _p(obj), _p(status), _p(iters), _p(resid))
    out = dict(coeffs=coeffs, obj=obj, status=status, iters=iters, residuals=resid)
    if time_grad:
        gT = np.empty((B, N))
        ctx.check(ctx.lib.anet_qp_solve_time_grad(*args, _p(gT)))
        out["grad_T"] = gT
    else:
        ctx.check(ctx.lib.anet_qp_solve(*args))
    return out


def qp_solve_vjp(order, iniPVA, finPVA, hPolys, times, grad_z, res=20, max_vel=4.0, max_acc=6.0, m34=1400.0,
                 settings=None, ctx=None):
    """Backward pass through the QP (anet_qp_solve_vjp; the KKT hook of layers.py:129-141 carried through to the
    durations): grad_z (B,N,3,2s) = d loss / d optimal coefficients -> grad_T (B,N) = d loss / d times.  Returns the
    dict of qp_solve with grad_T added.  Interior-point method only."""
    ctx = ctx or default_context()
    hp = np.ascontiguousarray(hPolys, dtype=np.float64)
    B, N, M, _ = hp.shape
    state = np.ascontiguousarray(np.stack([np.asarray(iniPVA, dtype=np.float64),
                                           np.asarray(finPVA, dtype=np.float64)], axis=1))
    T = np.ascontiguousarray(times, dtype=np.float64)
    D = 2 * order
    gz = np.ascontiguousarray(grad_z, dtype=np.float64).reshape(B, N, 3, D)
    if state.shape != (B, 2, 3, 3) or T.shape != (B, N):
        raise ValueError("shape mismatch")
    coeffs = np.empty((B, N, 3, D)); obj = np.empty(B); gT = np.empty((B, N))
    status = np.empty(B, dtype=np.int32); iters = np.empty(B, dtype=np.int32); resid = np.empty((B, 2))
    sp = ctypes.cast(ctypes.pointer(settings), ctypes.c_void_p) if settings is not None else None
    ctx.check(ctx.lib.anet_qp_solve_vjp(ctx.handle, int(order), N, B, int(res), M, float(max_vel), float(max_acc), float(m34),
                                        _p(state), _p(T), _p(hp), sp, _p(gz), _p(coeffs), _p(obj), _p(status), _p(iters),
                                        _p(resid), _p(gT)))
    return dict(coeffs=coeffs, obj=obj, status=status, iters=iters, residuals=resid, grad_T=gT)


def _qp_dev_common(order, state, times, hPolys, res, ctx):
    import torch
    if not (state.is_cuda and times.is_cuda and hPolys.is_cuda):
        raise ValueError("CUDA tensors expected")
    for t in (state, times, hPolys):
        if t.dtype != torch.float64 or not t.is_contiguous():
            raise ValueError("float64 contiguous tensors expected")
    B, N, M, _ = hPolys.shape
    if state.shape != (B, 2, 3, 3) or times.shape != (B, N):
        raise ValueError("shape mismatch: state (B,2,3,3) [start PVA, end PVA; row = axis], times (B,N), hPolys (B,N,M,4)")
    dev = times.device
    D = 2 * order
    work = torch.empty(int(ctx.lib.anet_qp_solve_workspace(int(order), N, B, int(res), M)), device=dev, dtype=torch.float64)
    out = dict(coeffs=torch.empty((B, N, 3, D), device=dev, dtype=torch.float64), obj=torch.empty(B, device=dev, dtype=torch.float64),
               status=torch.empty(B, device=dev, dtype=torch.int32), iters=torch.empty(B, device=dev, dtype=torch.int32),
               residuals=torch.empty((B, 2), device=dev, dtype=torch.float64))
    return B, N, M, work, out


def qp_solve_dev(order, state, times, hPolys, res=20, max_vel=4.0, max_acc=6.0, m34=1400.0, settings=None, time_grad=False,
                 stream=None, ctx=None, launch_order=None):
    """anet_qp_solve[_time_grad]_dev: the batched QPSolver::solve with torch CUDA tensors in and out, nothing through the
    host, asynchronous on `stream` (default: torch's current stream).  state (B,2,3,3) = [start PVA, end PVA] (row = axis,
    columns p, v, a), times (B,N), hPolys (B,N,M,4) rows a.x <= b with zero rows as padding.  Returns the dict of
    `qp_solve` as device tensors (grad_T (B,N) with time_grad=True).
    launch_order: optional int32 CUDA tensor (B,), a permutation -- the problem each successive workgroup of the interior-point
    kernel takes (`launch_order_from_counts(previous["iters"])` when re-solving a similar batch: anet_qp_solve_ordered_dev);
    results do not depend on it."""
    import torch
    ctx = ctx or default_context(times.device.index or 0)
    B, N, M, work, out = _qp_dev_common(order, state, times, hPolys, res, ctx)
    st = stream if stream is not None else torch.cuda.current_stream(times.device).cuda_stream
    sp = ctypes.cast(ctypes.pointer(settings), ctypes.c_void_p) if settings is not None else None
    q = lambda t: ctypes.c_void_p(t.data_ptr())
    args = (ctx.handle, int(order), N, B, int(res), M, float(max_vel), float(max_acc), float(m34), q(state), q(times), q(hPolys),
            sp, q(work), q(out["coeffs"]), q(out["obj"]), q(out["status"]), q(out["iters"]), q(out["residuals"]))
    if launch_order is not None:
        if time_grad:
            raise ValueError("launch_order: the plain solve only")
        if not (launch_order.is_cuda and launch_order.dtype == torch.int32 and launch_order.is_contiguous() and
                launch_order.shape == (B,)):
            raise ValueError("launch_order: contiguous int32 CUDA tensor of shape (B,)")
        ctx.check(ctx.lib.anet_qp_solve_ordered_dev(*args[:13], q(launch_order), *args[13:], ctypes.c_void_p(st)))
    elif time_grad:
        out["grad_T"] = torch.empty((B, N), device=times.device, dtype=torch.float64)
        ctx.check(ctx.lib.anet_qp_solve_time_grad_dev(*args, q(out["grad_T"]), ctypes.c_void_p(st)))
    else:
        ctx.check(ctx.lib.anet_qp_solve_dev(*args, ctypes.c_void_p(st)))
    out["_work"] = work
    return out


def qp_solve_vjp_dev(order, state, times, hPolys, grad_z, res=20, max_vel=4.0, max_acc=6.0, m34=1400.0, settings=None,
                     stream=None, ctx=None):
    """anet_qp_solve_vjp_dev: solve + backward pass on the device (what a torch.autograd.Function around the layer calls
    in its backward): grad_z (B,N,3,2s) = d loss / d optimal coefficients -> grad_T (B,N).  Tensors as in `qp_solve_dev`."""
    import torch
    ctx = ctx or default_context(times.device.index or 0)
    B, N, M, work, out = _qp_dev_common(order, state, times, hPolys, res, ctx)
    if not (grad_z.is_cuda and grad_z.dtype == torch.float64 and grad_z.is_contiguous() and grad_z.shape == out["coeffs"].shape):
        raise ValueError("grad_z: float64 contiguous CUDA tensor of shape (B,N,3,2s)")
    st = stream if stream is not None else torch.cuda.current_stream(times.device).cuda_stream
    sp = ctypes.cast(ctypes.pointer(settings), ctypes.c_void_p) if settings is not None else None
    q = lambda t: ctypes.c_void_p(t.data_ptr())
    out["grad_T"] = torch.empty((B, N), device=times.device, dtype=torch.float64)
    ctx.check(ctx.lib.anet_qp_solve_vjp_dev(ctx.handle, int(order), N, B, int(res), M, float(max_vel), float(max_acc), float(m34),
                                            q(state), q(times), q(hPolys), sp, q(grad_z), q(work), q(out["coeffs"]), q(out["obj"]),
                                            q(out["status"]), q(out["iters"]), q(out["residuals"]), q(out["grad_T"]),
                                            ctypes.c_void_p(st)))
    out["_work"] = work
    return out


class QPConfig:
    """struct QPConfig (qp_solver.hpp:14-26) without the ros::NodeHandle: the three parameters it reads."""

    def __init__(self, MaxVelBox=4.0, MaxAccBox=6.0, ConstRes=20):
        self.MaxVelBox, self.MaxAccBox, self.ConstRes = float(MaxVelBox), float(MaxAccBox), int(ConstRes)


class QPSolver:
    """class QPSolver (qp_solver.hpp:28-366): setOrder, solve, getObjCost -- one trajectory per call,
    same acceptance rule (status Solved and -0.01 <= objective <= 5000, qp_solver.hpp:334-352)."""

    def __init__(self, conf, ctx=None):
        self.config = conf
        self.order_ = None
        self.obj_cost_ = -1.0
        self._ctx = ctx
        self._method = QP_METHOD_INTERIOR_POINT

    def setMethod(self, method):
        """Extension: QP_METHOD_INTERIOR_POINT (default since ABI 2) or QP_METHOD_ADMM (OSQP's algorithm and tolerances)."""
        if method not in (QP_METHOD_ADMM, QP_METHOD_INTERIOR_POINT):
            raise ValueError("unknown method")
        self._method = method

    def setOrder(self, order):
        if order not in (3, 4):
            raise ValueError("order must be 3 (jerk) or 4 (snap)")
        self.order_ = int(order)

    def getObjCost(self):
        return self.obj_cost_

    def solve(self, iniPVA, finPVA, hPolys, times):
        """hPolys: list of (m_i,4) arrays (rows a,b: a.x <= b).  Returns (ok, qp_solution) with
        qp_solution flattened piece -> axis -> coefficient like the reference's Eigen::VectorXd."""
        seg = len(hPolys)
        M = max(1, max(p.shape[0] for p in hPolys))
        hp = np.zeros((1, seg, M, 4))
        for i, p in enumerate(hPolys):
            hp[0, i, :p.shape[0]] = p
        t = np.asarray(times, dtype=np.float64)[:seg]
        out = qp_solve(self.order_, np.asarray(iniPVA)[None], np.asarray(finPVA)[None], hp, t[None],
                       res=self.config.ConstRes, max_vel=self.config.MaxVelBox, max_acc=self.config.MaxAccBox,
                       settings=qp_settings(method=self._method),
                       ctx=self._ctx)
        result = float(np.float32(out["obj"][0]))          # the reference reads the objective into a float
        if result > 5000 or result < -0.01 or out["status"][0] != 1:
            return False, None
        self.obj_cost_ = float(out["obj"][0])
        return True, out["coeffs"][0].reshape(-1)
