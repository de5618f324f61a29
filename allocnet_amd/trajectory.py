"""Host mirror of the reference's Piece<D> / Trajectory<D> containers
(src/planner/include/gcopter/trajectory.hpp:37-645) and of network/utils/trajectory.py.

Structure (durations, coefficient matrices, junctions) is host bookkeeping, exactly like the
reference; everything that evaluates polynomials or the control-effort cost runs in the HIP
kernels through anet_traj_eval / anet_traj_cost.  A Trajectory here may hold one trajectory
(reference semantics) -- batches go through `traj_eval` / `traj_cost` directly.
"""
import ctypes
import numpy as np

from .context import default_context


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def traj_eval(coeffs, T, tq, deriv, ctx=None):
    """coeffs (B,N,3,D), T (B,N), tq (B,nq) absolute times -> (B,nq,3)."""
    ctx = ctx or default_context()
    coeffs = np.ascontiguousarray(coeffs, dtype=np.float64)
    T = np.ascontiguousarray(T, dtype=np.float64)
    tq = np.ascontiguousarray(tq, dtype=np.float64)
    B, N, three, D = coeffs.shape
    if three != 3 or T.shape != (B, N) or tq.shape[0] != B:
        raise ValueError("shape mismatch")
    nq = tq.shape[1]
    out = np.empty((B, nq, 3))
    ctx.check(ctx.lib.anet_traj_eval(ctx.handle, D // 2, N, B, _ptr(coeffs), _ptr(T), nq, _ptr(tq),
                                     int(deriv), _ptr(out)))
    return out


def traj_cost(coeffs, T, order=None, m34=1400.0, ctx=None):
    """Trajectory::getTrajCost batched: coeffs (B,N,3,D), T (B,N) -> (B,)."""
    ctx = ctx or default_context()
    coeffs = np.ascontiguousarray(coeffs, dtype=np.float64)
    T = np.ascontiguousarray(T, dtype=np.float64)
    B, N, _, D = coeffs.shape
    s = D // 2
    if order is not None and order != s:
        raise ValueError("order must be D/2 (3 for degree 5, 4 for degree 7)")
    out = np.empty(B)
    ctx.check(ctx.lib.anet_traj_cost(ctx.handle, s, N, B, _ptr(coeffs), _ptr(T), float(m34), _ptr(out)))
    return out


def traj_cost_grad_T(coeffs, T, m34=1400.0, ctx=None):
    """d getTrajCost / dT at fixed coefficients, (B,N): the time gradient the reference's training
    propagates (layers.py:143-147 with the QP solution detached, :121)."""
    ctx = ctx or default_context()
    coeffs = np.ascontiguousarray(coeffs, dtype=np.float64)
    T = np.ascontiguousarray(T, dtype=np.float64)
    B, N, _, D = coeffs.shape
    out = np.empty((B, N))
    ctx.check(ctx.lib.anet_traj_cost_grad_T(ctx.handle, D // 2, N, B, _ptr(coeffs), _ptr(T), float(m34), _ptr(out)))
    return out


def traj_max_rate(coeffs, T, which, ctx=None):
    """Per-piece maximum of ||v|| (which=1) or ||a|| (which=2): Piece::getMaxVelRate/getMaxAccRate batched.
    coeffs (B,N,3,D), T (B,N) -> (B,N)."""
    ctx = ctx or default_context()
    coeffs = np.ascontiguousarray(coeffs, dtype=np.float64)
    T = np.ascontiguousarray(T, dtype=np.float64)
    B, N, _, D = coeffs.shape
    out = np.empty((B, N))
    ctx.check(ctx.lib.anet_traj_max_rate(ctx.handle, D // 2, N, B, _ptr(coeffs), _ptr(T), int(which), _ptr(out)))
    return out


def piece_normalized_coeffs(coeffs, T, deriv, ctx=None):
    """Piece::normalizePosCoeffMat / normalizeVelCoeffMat / normalizeAccCoeffMat (trajectory.hpp:135-171) for a batch of pieces:
    coefficients of position (deriv 0), velocity (1) or acceleration (2) in normalised time.  coeffs (P,3,D), T (P,) -> (P,3,D-deriv)."""
    ctx = ctx or default_context()
    coeffs = np.ascontiguousarray(coeffs, dtype=np.float64)
    T = np.ascontiguousarray(T, dtype=np.float64)
    P, _, D = coeffs.shape
    out = np.empty((P, 3, D - int(deriv)))
    ctx.check(ctx.lib.anet_piece_normalized_coeffs(ctx.handle, D // 2, P, _ptr(coeffs), _ptr(T), int(deriv), _ptr(out)))
    return out


class Piece:
    """Piece<D> (trajectory.hpp:37-316): duration + 3 x (D+1) coefficient matrix, highest power first."""

    def __init__(self, dur, cMat, ctx=None):
        self.duration = float(dur)
        self.coeffMat = np.array(cMat, dtype=np.float64)
        if self.coeffMat.ndim != 2 or self.coeffMat.shape[0] != 3:
            raise ValueError("coefficient matrix must be 3 x (D+1)")
        self._ctx = ctx

    def getDim(self):
        return 3

    def getDegree(self):
        return self.coeffMat.shape[1] - 1

    def getDuration(self):
        return self.duration

    def getCoeffMat(self):
        return self.coeffMat

    def _eval(self, t, d):
        # a one-piece trajectory with a huge duration: no piece location involved
        co = self.coeffMat[None, None]
        return traj_eval(co, np.array([[np.inf]]), np.array([[float(t)]]), d, ctx=self._ctx)[0, 0]

    def getPos(self, t):
        return self._eval(t, 0)

    def getVel(self, t):
        return self._eval(t, 1)

    def getAcc(self, t):
        return self._eval(t, 2)

    def getJer(self, t):
        return self._eval(t, 3)

    # trajectory.hpp:135-171
    def normalizePosCoeffMat(self):
        return piece_normalized_coeffs(self.coeffMat[None], np.array([self.duration]), 0, ctx=self._ctx)[0]

    def normalizeVelCoeffMat(self):
        return piece_normalized_coeffs(self.coeffMat[None], np.array([self.duration]), 1, ctx=self._ctx)[0]

    def normalizeAccCoeffMat(self):
        return piece_normalized_coeffs(self.coeffMat[None], np.array([self.duration]), 2, ctx=self._ctx)[0]

    # trajectory.hpp:177-314
    def getMaxVelRate(self):
        return float(traj_max_rate(self.coeffMat[None, None], np.array([[self.duration]]), 1, ctx=self._ctx)[0, 0])

    def getMaxAccRate(self):
        return float(traj_max_rate(self.coeffMat[None, None], np.array([[self.duration]]), 2, ctx=self._ctx)[0, 0])

    def checkMaxVelRate(self, maxVelRate):
        return self.getMaxVelRate() < maxVelRate

    def checkMaxAccRate(self, maxAccRate):
        return self.getMaxAccRate() < maxAccRate


class Trajectory:
    """Trajectory<D> (trajectory.hpp:317-645)."""

    def __init__(self, durs=None, cMats=None, ctx=None):
        self.pieces = []
        self._ctx = ctx
        if durs is not None:
            for d, c in zip(durs, cMats):       # min(durs.size(), cMats.size()) like the reference
                self.pieces.append(Piece(d, c, ctx))

    # --- container interface --------------------------------------------------------------
    def getPieceNum(self):
        return len(self.pieces)

    def getDurations(self):
        return np.array([p.duration for p in self.pieces])

    def getTotalDuration(self):
        return float(sum(p.duration for p in self.pieces))

    def __getitem__(self, i):
        return self.pieces[i]

    def __iter__(self):
        return iter(self.pieces)

    def clear(self):
        self.pieces.clear()

    def reserve(self, n):
        return None

    def emplace_back(self, *args):
        if len(args) == 1:
            self.pieces.append(args[0])
        else:
            self.pieces.append(Piece(args[0], args[1], self._ctx))

    def append(self, traj):
        self.pieces.extend(traj.pieces)

    def _arrays(self):
        if not self.pieces:
            raise RuntimeError("empty trajectory")
        co = np.stack([p.coeffMat for p in self.pieces])[None]
        return co, self.getDurations()[None]

    # --- evaluation (GPU) -----------------------------------------------------------------
    def _eval(self, t, d):
        co, T = self._arrays()
        t = np.atleast_1d(np.asarray(t, dtype=np.float64))
        out = traj_eval(co, T, t[None], d, ctx=self._ctx)[0]
        return out[0] if out.shape[0] == 1 else out

    def getPos(self, t):
        return self._eval(t, 0)

    def getVel(self, t):
        return self._eval(t, 1)

    def getAcc(self, t):
        return self._eval(t, 2)

    def getJer(self, t):
        return self._eval(t, 3)

    def getTrajCost(self, order, m34=1400.0):
        co, T = self._arrays()
        return float(traj_cost(co, T, order, m34, ctx=self._ctx)[0])

    # --- dynamic feasibility (trajectory.hpp:576-630): maximum over the pieces
    def getMaxVelRate(self):
        co, T = self._arrays()
        return float(traj_max_rate(co, T, 1, ctx=self._ctx).max())

    def getMaxAccRate(self):
        co, T = self._arrays()
        return float(traj_max_rate(co, T, 2, ctx=self._ctx).max())

    def checkMaxVelRate(self, maxVelRate):
        return self.getMaxVelRate() < maxVelRate

    def checkMaxAccRate(self, maxAccRate):
        return self.getMaxAccRate() < maxAccRate

    # --- junctions (trajectory.hpp:540-574): direct coefficient reads except at the very end
    def getPositions(self):
        N = self.getPieceNum()
        D = self.pieces[0].getDegree()
        pos = np.zeros((3, N + 1))
        for i in range(N):
            pos[:, i] = self.pieces[i].coeffMat[:, D]
        pos[:, N] = self.pieces[-1].getPos(self.pieces[-1].duration)
        return pos

    def getJuncPos(self, j):
        D = self.pieces[0].getDegree()
        if j != self.getPieceNum():
            return self.pieces[j].coeffMat[:, D].copy()
        return self.pieces[j - 1].getPos(self.pieces[j - 1].duration)

    def getJuncVel(self, j):
        D = self.pieces[0].getDegree()
        if j != self.getPieceNum():
            return self.pieces[j].coeffMat[:, D - 1].copy()
        return self.pieces[j - 1].getVel(self.pieces[j - 1].duration)

    def getJuncAcc(self, j):
        D = self.pieces[0].getDegree()
        if j != self.getPieceNum():
            return self.pieces[j].coeffMat[:, D - 2] * 2.0
        return self.pieces[j - 1].getAcc(self.pieces[j - 1].duration)

    def locatePieceIdx(self, t):
        """Returns (idx, local t) -- the reference mutates t in place (trajectory.hpp:496-514)."""
        N = self.getPieceNum()
        idx = 0
        while idx < N and t > self.pieces[idx].duration:
            t -= self.pieces[idx].duration
            idx += 1
        if idx == N:
            idx -= 1
            t += self.pieces[idx].duration
        return idx, t
