"""MINCO host mirror.

`minco.hpp` is not in the reference tree; the class below follows the
upstream GCOPTER method names the north star asks for -- setConditions / setParameters /
getCoeffs / getEnergy -- batched over B independent trajectories.  Data conventions are the
reference's (see include/allocnet_amd.h).  S3 = min-jerk (degree 5), S4 = min-snap (degree 7),
the reference's own order numbering (planner.yaml:23, learning_planner.hpp:203-233).
"""
import ctypes
import numpy as np

from .context import default_context


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _f64c(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != shape:
        raise ValueError(f"expected shape {shape}, got {a.shape}")
    return a


def recommended_ld(batch):
    """Row stride for the batch-minor layout that avoids power-of-two strides (anet_recommended_ld)."""
    from ._lib import load
    return int(load().anet_recommended_ld(int(batch)))


def minco_solve(head, tail, wps, T, s, want_coeffs=True, ctx=None):
    """Host (numpy, trajectory-major) entry point -> anet_minco_solve.

    head, tail : (B, 3, c);  wps : (B, N-1, 3);  T : (B, N)
    returns coeffs (B, N, 3, 2s) [or None] and energy (B,)
    """
    ctx = ctx or default_context()
    head = _f64c(head)
    B, three, c = head.shape
    if three != 3:
        raise ValueError("head must be (B, 3, c)")
    tail = _f64c(tail, (B, 3, c))
    T = _f64c(T)
    N = T.shape[1]
    if T.shape != (B, N):
        raise ValueError("T must be (B, N)")
    wps = _f64c(wps if wps is not None else np.zeros((B, 0, 3)), (B, N - 1, 3))
    coeffs = np.empty((B, N, 3, 2 * s)) if want_coeffs else None
    energy = np.empty(B)
    ctx.check(ctx.lib.anet_minco_solve(ctx.handle, s, c, N, B, _ptr(head), _ptr(tail), _ptr(wps),
                                       _ptr(T), _ptr(coeffs), _ptr(energy)))
    return coeffs, energy


def minco_solve_dev(head, tail, wps, T, s, c, N, B, coeffs=None, energy=None, stream=None, ctx=None):
    """Device entry point -> anet_minco_solve_dev.  All arguments are torch CUDA float64 tensors in
    the batch-minor layout (rows = per-trajectory fields, columns = batch, row stride ld):
    head/tail (3c, ld), wps ((N-1)*3, ld), T (N, ld), coeffs (N*3*2s, ld), energy (>=B,)."""
    import torch
    ctx = ctx or default_context(T.device.index or 0)
    ld = T.stride(0) if T.dim() == 2 else T.shape[-1]
    for t in (head, tail, T) + ((wps,) if N > 1 else ()) + ((coeffs,) if coeffs is not None else ()):
        if t.dtype != torch.float64 or not t.is_cuda or t.stride(-1) != 1 or (t.dim() == 2 and t.stride(0) != ld):
            raise ValueError("batch-minor float64 CUDA tensors with a common row stride expected")
    if stream is None:
        stream = torch.cuda.current_stream(T.device).cuda_stream
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    ctx.check(ctx.lib.anet_minco_solve_dev(ctx.handle, s, c, N, B, ld, p(head), p(tail),
                                           p(wps) if N > 1 else None, p(T), p(coeffs), p(energy),
                                           ctypes.c_void_p(stream)))
    return coeffs, energy


class BoundMincoSolve:
    """anet_minco_solve_dev with its arguments checked and converted ONCE (`bind_minco_solve`): calling the object launches the
    solve on the bound tensors and the bound stream and costs the ctypes trampoline plus the launch -- 4.5 us per 1024-trajectory
    launch on MI355X, which is the kernel's own duration, against 9.4 us through `minco_solve_dev`, whose per-call tensor checks,
    pointer conversions and stream lookup (5 us of Python) are then the bound of a loop of small launches.  The tensors are
    kept alive by the object; their CONTENTS may change between calls (that is the point), their storage may not."""

    def __init__(self, head, tail, wps, T, s, c, N, B, coeffs=None, energy=None, stream=None, ctx=None):
        import torch
        ctx = ctx or default_context(T.device.index or 0)
        ld = T.stride(0) if T.dim() == 2 else T.shape[-1]
        for t in (head, tail, T) + ((wps,) if N > 1 else ()) + ((coeffs,) if coeffs is not None else ()):
            if t.dtype != torch.float64 or not t.is_cuda or t.stride(-1) != 1 or (t.dim() == 2 and t.stride(0) != ld):
                raise ValueError("batch-minor float64 CUDA tensors with a common row stride expected")
        if energy is not None and (energy.dtype != torch.float64 or not energy.is_cuda or energy.numel() < B):
            raise ValueError("energy: a float64 CUDA tensor of at least B elements expected")
        if stream is None:
            stream = torch.cuda.current_stream(T.device).cuda_stream
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        self._keep = (head, tail, wps, T, coeffs, energy)
        self._ctx, self._fn = ctx, ctx.lib.anet_minco_solve_dev
        self._args = (ctx.handle, s, c, N, B, ld, p(head), p(tail), p(wps) if N > 1 else None, p(T), p(coeffs), p(energy))
        self._stream = ctypes.c_void_p(stream)
        self.coeffs, self.energy = coeffs, energy

    def with_stream(self, stream):
        """The same bound call on another stream (a raw hipStream_t handle, e.g. `torch.cuda.Stream.cuda_stream`)."""
        other = object.__new__(BoundMincoSolve)
        other.__dict__.update(self.__dict__)
        other._stream = ctypes.c_void_p(stream)
        return other

    def __call__(self):
        rc = self._fn(*self._args, self._stream)
        if rc:
            self._ctx.check(rc)


def bind_minco_solve(head, tail, wps, T, s, c, N, B, coeffs=None, energy=None, stream=None, ctx=None):
    """`minco_solve_dev`'s arguments, checked and converted once -> a callable that launches the solve (BoundMincoSolve)."""
    return BoundMincoSolve(head, tail, wps, T, s, c, N, B, coeffs=coeffs, energy=energy, stream=stream, ctx=ctx)


def minco_solve_wide_spread_dev(head, tail, wps, T, s, c, N, B, coeffs=None, energy=None, min_spread=50.0, stream=None,
                                ctx=None):
    """anet_minco_solve_wide_spread_dev: trajectories whose durations spread over more than `min_spread` are solved
    again by the pivoted collocation solve, their coeffs / energy overwritten (same tensors as minco_solve_dev)."""
    import torch
    ctx = ctx or default_context(T.device.index or 0)
    ld = T.stride(0) if T.dim() == 2 else T.shape[-1]
    if stream is None:
        stream = torch.cuda.current_stream(T.device).cuda_stream
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    ctx.check(ctx.lib.anet_minco_solve_wide_spread_dev(ctx.handle, s, c, N, B, ld, p(head), p(tail), p(wps) if N > 1 else None,
                                                       p(T), float(min_spread), p(coeffs), p(energy), ctypes.c_void_p(stream)))
    return coeffs, energy


def minco_sample_costs(head, tail, wps, T_samples, s, rho=0.0, ctx=None):
    """Time-allocation sampling for ONE problem (anet_minco_sample_costs): head / tail (3, c), wps (N-1, 3), T_samples
    (K, N) candidate duration vectors -> cost (K,) = int (p^(s))^2 + rho * sum T of each candidate, one launch."""
    ctx = ctx or default_context()
    head = _f64c(head); tail = _f64c(tail)
    T = _f64c(T_samples)
    K, N = T.shape
    c = head.shape[1]
    wps = _f64c(wps if N > 1 else np.zeros((0, 3)))
    if head.shape != (3, c) or tail.shape != (3, c) or wps.shape != (max(N - 1, 0), 3):
        raise ValueError("head / tail (3, c), wps (N-1, 3), T_samples (K, N)")
    cost = np.empty(K)
    ctx.check(ctx.lib.anet_minco_sample_costs(ctx.handle, int(s), c, N, K, _ptr(head), _ptr(tail), _ptr(wps), _ptr(T),
                                              float(rho), _ptr(cost)))
    return cost


def minco_sample_costs_dev(head, tail, wps, T, s, c, N, problems, samples_per_problem, rho=0.0, cost=None, stream=None,
                           ctx=None):
    """Device entry point -> anet_minco_sample_costs_dev: head / tail (3c, ldp), wps ((N-1)*3, ldp) per PROBLEM, T (N, ld)
    per SAMPLE (torch CUDA float64, batch-minor); sample b belongs to problem b // samples_per_problem.  Returns cost
    (problems * samples_per_problem,)."""
    import torch
    ctx = ctx or default_context(T.device.index or 0)
    total = problems * samples_per_problem
    cost = cost if cost is not None else torch.empty(total, device=T.device, dtype=torch.float64)
    if stream is None:
        stream = torch.cuda.current_stream(T.device).cuda_stream
    ctx.check(ctx.lib.anet_minco_sample_costs_dev(ctx.handle, int(s), int(c), int(N), int(problems), int(samples_per_problem),
                                                  T.stride(0), head.stride(0), _tptr(head), _tptr(tail),
                                                  _tptr(wps) if N > 1 else None, _tptr(T), float(rho), _tptr(cost),
                                                  ctypes.c_void_p(stream)))
    return cost


class MINCO:
    """Batched MINCO_S{s}NU mirror: setConditions -> setParameters -> getCoeffs/getEnergy."""

    def __init__(self, s, ctx=None):
        if s not in (2, 3, 4):
            raise ValueError("order s must be 2, 3 or 4")
        self.s = s
        self.ctx = ctx
        self.N = None
        self._coeffs = self._energy = None

    def setConditions(self, headState, tailState, pieceNum):
        """headState/tailState: (B, 3, c) or (3, c) (row = axis, cols p,v,a[,j])."""
        h = np.asarray(headState, dtype=np.float64)
        t = np.asarray(tailState, dtype=np.float64)
        self._single = h.ndim == 2
        if self._single:
            h, t = h[None], t[None]
        if h.shape != t.shape or h.shape[1] != 3 or not (1 <= h.shape[2] <= self.s):
            raise ValueError("boundary states must be (B,3,c) with 1 <= c <= s")
        self.head, self.tail, self.N = h, t, int(pieceNum)

    def setParameters(self, inPs, ts):
        """inPs: (B, N-1, 3) interior waypoints; ts: (B, N) durations."""
        if self.N is None:
            raise RuntimeError("setConditions first")
        ts = np.asarray(ts, dtype=np.float64)
        inPs = np.asarray(inPs, dtype=np.float64)
        if self._single:
            ts, inPs = ts[None], inPs[None]
        B = self.head.shape[0]
        if ts.shape != (B, self.N):
            raise ValueError("ts must be (B, N)")
        self.T = ts
        self._coeffs, self._energy = minco_solve(self.head, self.tail, inPs.reshape(B, self.N - 1, 3),
                                                 ts, self.s, ctx=self.ctx)

    def getCoeffs(self):
        return self._coeffs[0] if self._single else self._coeffs

    def getEnergy(self):
        return float(self._energy[0]) if self._single else self._energy


class MINCO_S2NU(MINCO):
    def __init__(self, ctx=None):
        super().__init__(2, ctx)


class MINCO_S3NU(MINCO):
    def __init__(self, ctx=None):
        super().__init__(3, ctx)


class MINCO_S4NU(MINCO):
    def __init__(self, ctx=None):
        super().__init__(4, ctx)


# ---------------------------------------------------------------------------------------------
# cost + analytic gradients (penalty functional on the reference's inequality rows)
# ---------------------------------------------------------------------------------------------
def make_penalty(rho=0.0, w_corridor=0.0, w_vel=0.0, w_acc=0.0, smooth_mu=1e-2, max_vel=4.0,
                 max_acc=6.0, res=20, poly_rows=0):
    """struct anet_penalty.  Defaults: MaxVelBox/MaxAccBox/ConstRes of config/planner.yaml:17-21 and
    the smoothing FIRI uses for its own smoothedL1 (firi.hpp:218)."""
    from ._lib import Penalty
    return Penalty(rho, w_corridor, w_vel, w_acc, smooth_mu, max_vel, max_acc, int(res), int(poly_rows))


def minco_cost_grad(head, tail, wps, T, s, hpolys=None, penalty=None, want_coeffs=False, ctx=None):
    """Host entry point -> anet_minco_cost_grad.
    head, tail (B,3,c); wps (B,N-1,3); T (B,N); hpolys (B,N,M,4) rows a.x<=b (zero rows = padding).
    Returns cost (B,), gradP (B,N-1,3), gradT (B,N) [, coeffs (B,N,3,2s)]."""
    ctx = ctx or default_context()
    head = _f64c(head)
    B, _, c = head.shape
    tail = _f64c(tail, (B, 3, c))
    T = _f64c(T)
    N = T.shape[1]
    wps = _f64c(wps if wps is not None else np.zeros((B, 0, 3)), (B, N - 1, 3))
    pen = penalty
    if hpolys is not None:
        hpolys = _f64c(hpolys)
        if pen is None or hpolys.shape != (B, N, pen.poly_rows, 4):
            raise ValueError("hpolys must be (B, N, penalty.poly_rows, 4)")
    cost = np.empty(B); gradP = np.empty((B, N - 1, 3)); gradT = np.empty((B, N))
    coeffs = np.empty((B, N, 3, 2 * s)) if want_coeffs else None
    ctx.check(ctx.lib.anet_minco_cost_grad(
        ctx.handle, s, c, N, B, _ptr(head), _ptr(tail), _ptr(wps), _ptr(T),
        _ptr(hpolys) if hpolys is not None else None,
        ctypes.cast(ctypes.pointer(pen), ctypes.c_void_p) if pen is not None else None,
        _ptr(cost), _ptr(gradP), _ptr(gradT), _ptr(coeffs)))
    return (cost, gradP, gradT, coeffs) if want_coeffs else (cost, gradP, gradT)


def _tptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def minco_cost_grad_launches(s, N, B, penalty=None, ctx=None, c=3):
    """anet_minco_cost_grad_launches: 1 if a cost + gradient evaluation of this shape (order s, boundary count c, N pieces) runs
    as ONE launch (k_minco_cost_grad_fused) on this context's device, 3 for solve -> piece gradients -> adjoint."""
    ctx = ctx or default_context(0)
    n = ctx.lib.anet_minco_cost_grad_launches(ctx.handle, s, c, N, B, ctypes.cast(ctypes.pointer(penalty), ctypes.c_void_p)
                                              if penalty is not None else None)
    if n < 0:
        ctx.check(n)
    return n


def minco_piece_grad_shape(s, N, B, penalty=None, ctx=None):
    """anet_minco_piece_grad_shape: the launch shape of the penalty / energy-gradient kernel for this batch -- 0 a lane per
    (trajectory, piece), 1 / 2 the small-batch shapes, 3 k_piece_grad_mx (the basis-table contractions on the FP64 matrix
    instructions: large batches, orders 3 and 4, 20 samples per piece)."""
    ctx = ctx or default_context(0)
    n = ctx.lib.anet_minco_piece_grad_shape(ctx.handle, s, N, B, ctypes.cast(ctypes.pointer(penalty), ctypes.c_void_p)
                                            if penalty is not None else None)
    if n < 0:
        ctx.check(n)
    return n


def minco_cost_grad_dev(head, tail, wps, T, s, c, N, B, hpolys=None, penalty=None, work=None, cost=None,
                        gradP=None, gradT=None, coeffs=None, stream=None, ctx=None):
    """Device entry point -> anet_minco_cost_grad_dev (torch CUDA float64, batch-minor, common ld)."""
    import torch
    ctx = ctx or default_context(T.device.index or 0)
    ld = T.stride(0)
    dev = T.device
    if work is None:
        work = torch.empty(ctx.lib.anet_minco_cost_grad_workspace(s, N, ld), device=dev, dtype=torch.float64)
    cost = cost if cost is not None else torch.empty(ld, device=dev, dtype=torch.float64)
    gradP = gradP if gradP is not None else torch.empty(max(3 * (N - 1), 1), ld, device=dev, dtype=torch.float64)
    gradT = gradT if gradT is not None else torch.empty(N, ld, device=dev, dtype=torch.float64)
    if stream is None:
        stream = torch.cuda.current_stream(dev).cuda_stream
    ctx.check(ctx.lib.anet_minco_cost_grad_dev(
        ctx.handle, s, c, N, B, ld, _tptr(head), _tptr(tail), _tptr(wps) if N > 1 else None, _tptr(T),
        _tptr(hpolys), ctypes.cast(ctypes.pointer(penalty), ctypes.c_void_p) if penalty is not None else None,
        _tptr(work), _tptr(cost), _tptr(gradP), _tptr(gradT), _tptr(coeffs), ctypes.c_void_p(stream)))
    return cost, gradP, gradT, work
