"""Host mirrors of the reference's training-side QP objects:

  MinTrajOpt  network/utils/min_traj_opt.py:19-178   (params dict, update(...) -> .params=[Q,A,b,G1,h1,G2,h2])
  OsqpLayer   network/utils/learning/layers.py:35-247 (forward / forward4lstm: solve + loss terms)

Assembly and solve run on the GPU (anet_qp_assemble / anet_qp_solve).  There is no autograd tape here;
what the reference's backward pass delivers to the segment times is returned explicitly: in layers.py
the QP solution is a detached leaf (:121,222), so d(objc)/dTimes flows only through Q(T) -- that is
`OsqpLayer.time_grad` (anet_traj_cost_grad_T / path_length), plus 1/segments from the mean-time term.
"""
import numpy as np

from . import qp as _qp
from .trajectory import traj_cost_grad_T


class MinTrajOpt:
    def __init__(self, params, ctx=None):
        if not isinstance(params, dict):
            raise ValueError("pass the parameter dict (the reference reads utils/params.yaml when given [])")
        self.params_cfg = params
        self.order = params["planning"]["order"]
        self.state_dim = params["planning"]["state_dim"]
        self.dim = params["planning"]["dim"]
        self.res = params["planning"]["res"]
        self.D = 2 * self.order
        self.use_time_factor = params["planning"]["use_time_factor"]      # false in the reference configuration (params.yaml:24)
        self.phy_limits = [params["physical_limits"][k] for k in ("max_vel", "max_acc", "max_jerk")]
        self.phase1_phy_limits = [params["phase1_physical_limits"][k] for k in ("max_vel", "max_acc", "max_jerk", "inf_dis")]
        self._ctx = ctx

    def update(self, state, hpolys, time_factor, phase=1, traj_times=None, seq_len=5):
        """state (9,2): rows px,vx,ax,py,..; col 0 start, col 1 end.  hpolys (rows,4,seq_len) zero padded.
        time_factor (seq_len,): segment times (network output).  min_traj_opt.py:68-178."""
        state = np.asarray(state, dtype=np.float64)
        hpolys = np.asarray(hpolys, dtype=np.float64)
        tf = np.asarray(time_factor, dtype=np.float64)
        self.hpolys = []
        for i in range(seq_len):
            poly = hpolys[:, :, i]
            if np.linalg.norm(poly) <= 1.0:          # :78-79 (float32 norm in the reference; same threshold)
                break
            for j in range(poly.shape[0]):           # :82-85 strip the zero-padded rows
                if np.linalg.norm(poly[j, :]) <= 0.0:
                    poly = poly[0:j, :]
                    break
            self.hpolys.append(poly)
        self.seg = len(self.hpolys)
        if self.seg == 0:
            raise ValueError("no polytope survives the zero-padding test")
        self.start_state = state[:, 0]
        self.end_state = state[:, 1]
        self.start = np.array([self.start_state[0], self.start_state[3], self.start_state[6]])
        self.goal = np.array([self.end_state[0], self.end_state[3], self.end_state[6]])
        self.var_num = self.seg * self.dim * self.D
        self.eq_num = (2 * self.state_dim + self.order * (self.seg - 1)) * self.dim
        const_num = sum(p.shape[0] for p in self.hpolys)
        self.ineq_num1 = self.res * const_num
        self.ineq_num2 = self.res * 4 * self.dim * self.seg
        self.ineq_num = self.ineq_num1 + self.ineq_num2
        if self.use_time_factor:
            # min_traj_opt.py:113-136: waypoints from the deepest common points of consecutive polytopes, a lower bound
            # on every segment time from the limits, and the network output as a factor on top of it
            self.inner_pts = self.get_inner_pts()
            if self.inner_pts is None:
                raise ValueError("consecutive polytopes have no common interior point")
            self.waypts = np.vstack([self.start] + ([self.inner_pts] if len(self.inner_pts) else []) + [self.goal])
            self.path_length = float(sum(np.linalg.norm(self.waypts[i + 1] - self.waypts[i]) for i in range(len(self.waypts) - 1)))
            self.time_lb = self.getT_lbs(self.waypts, self.phy_limits[0], self.phy_limits[1])
            self.Times = self.time_lb + self.time_lb * tf
            if traj_times is not None:
                with np.errstate(divide="ignore", invalid="ignore"):
                    self.ref_time_factor = np.asarray(traj_times, dtype=np.float64) / self.time_lb
        else:
            self.Times = tf
            self.path_length = float(np.linalg.norm(self.goal - self.start))
            if traj_times is not None:
                self.ref_time_factor = np.asarray(traj_times, dtype=np.float64)
        tf = self.Times
        lim = self.phase1_phy_limits if phase == 1 else self.phy_limits
        ini = self.start_state.reshape(3, 3)
        fin = self.end_state.reshape(3, 3)
        Q, A, b, G, h = _qp.qp_assemble(self.order, ini, fin, self.hpolys, tf[:self.seg], res=self.res,
                                        max_vel=lim[0], max_acc=lim[1], row_order=_qp.ORDER_PYTHON, ctx=self._ctx)
        n1 = self.ineq_num1
        self._limits = (lim[0], lim[1])
        self.params = [Q, A, b, G[:n1], h[:n1], G[n1:], h[n1:]]


    # ---- use_time_factor branch (min_traj_opt.py:185-296): host-side preprocessing, as in the reference ----
    def getT_lbs(self, pts, maxv, maxa):
        """min_traj_opt.py:195-211.  The reference fills a float32 tensor of five entries from float32 limits; the
        same roundings are applied here so that Times agrees to the last bit."""
        times = np.zeros(5, dtype=np.float32)
        mv, ma = np.float32(maxv), np.float32(maxa)
        for i in range(pts.shape[0] - 1):
            dis = (pts[i + 1] - pts[i])
            vel_t = np.abs((dis / mv))
            acc_t = np.abs((2 * dis / ma))
            times[i] = max(float(vel_t.max()), float(np.sqrt(acc_t.max())))
        return times.astype(np.float64)

    def get_inner_pts(self):
        """min_traj_opt.py:251-275: one waypoint per pair of consecutive polytopes."""
        n = self.seg
        if n <= 1:
            return np.zeros((0, 3))
        if n == 2:
            return (0.5 * (self.start + self.goal)).reshape(1, 3)
        pts = []
        for i in range(n - 1):
            pt = self.get_inner_points(np.vstack((self.hpolys[i], self.hpolys[i + 1])), 0.01)
            if pt is None:
                return None
            pts.append(pt)
        return np.array(pts)

    @staticmethod
    def get_inner_points(hpoly, eps=0.001):
        """min_traj_opt.py:279-296: the point of largest common slack, the 4-variable LP  max d  s.t.  A x + d <= b,
        d >= 0, through scipy.optimize.linprog exactly as the reference calls it (host-side preprocessing of a branch
        the reference configuration disables; the batched corridor code has its own deepest-point kernel)."""
        import scipy.optimize
        A = np.hstack((hpoly[:, 0:3], np.ones((hpoly.shape[0], 1))))
        res = scipy.optimize.linprog([0, 0, 0, -1], A_ub=A, b_ub=hpoly[:, 3],
                                     bounds=[(-np.inf, np.inf)] * 3 + [(0, np.inf)])
        if res.fun is None:
            return None
        return res.x[0:3]


class OsqpLayer:
    """Solve + loss terms of layers.py:51-151 (forward) and :153-247 (forward4lstm)."""

    def __init__(self, ctx=None, method=_qp.QP_METHOD_INTERIOR_POINT):
        """method: QP_METHOD_INTERIOR_POINT (default: the optimum to 1e-6, an order of magnitude faster, solves every
        problem either method can) or QP_METHOD_ADMM (OSQP's algorithm and tolerances, as layers.py:77-81 runs them)."""
        self._ctx = ctx
        self._settings = _qp.qp_settings(method=method)
        self.time_grad = None             # what the reference's autograd delivers (z detached): 1/2 z'(dQ/dT)z
        self.implicit_time_grad = None    # d(optimal objc)/dTimes through the QP (anet_qp_solve_time_grad)

    def _solve(self, qp_traj):
        M = max(p.shape[0] for p in qp_traj.hpolys)
        hp = np.zeros((1, qp_traj.seg, M, 4))
        for i, p in enumerate(qp_traj.hpolys):
            hp[0, i, :p.shape[0]] = p
        ini = qp_traj.start_state.reshape(1, 3, 3)
        fin = qp_traj.end_state.reshape(1, 3, 3)
        T = qp_traj.Times[:qp_traj.seg][None]
        out = _qp.qp_solve(qp_traj.order, ini, fin, hp, T, res=qp_traj.res, max_vel=qp_traj._limits[0],
                           max_acc=qp_traj._limits[1], settings=self._settings, time_grad=True, ctx=self._ctx)
        return out, T

    def forward(self, qp_traj):
        segments = qp_traj.seg
        Times = qp_traj.Times
        out, T = self._solve(qp_traj)
        curr_obj1_val = float(np.sum(Times[:segments]) / (1.0 * segments))
        zero_segments = 5 - segments
        curr_padding_loss = float(np.mean(Times[segments:] ** 2)) if zero_segments != 0 else 0.0
        if out["status"][0] != 1:
            curr_objt_val = None
            if hasattr(qp_traj, "ref_time_factor"):
                curr_objt_val = float(np.mean((Times[:segments] - qp_traj.ref_time_factor[:segments]) ** 2) / segments
                                      + curr_padding_loss)
            self.time_grad = None
            self.implicit_time_grad = None
            return None, curr_obj1_val, curr_objt_val, None, curr_padding_loss
        z = out["coeffs"][0].reshape(-1)
        curr_objc_val = float(out["obj"][0] / qp_traj.path_length)
        g = traj_cost_grad_T(out["coeffs"], T, m34=1400.0, ctx=self._ctx)[0] / qp_traj.path_length
        self.time_grad = np.zeros_like(Times)
        self.time_grad[:segments] = g
        self.implicit_time_grad = np.zeros_like(Times)
        self.implicit_time_grad[:segments] = out["grad_T"][0] / qp_traj.path_length
        return z, curr_obj1_val, None, curr_objc_val, curr_padding_loss

    def backward(self, qp_traj, grad_z):
        """The backward pass the reference's KKT hook is after (layers.py:129-141, 230-243), carried through to the
        durations: grad_z = d loss / d z (the flat solution `forward` returns, or (N,3,2s)) -> d loss / d Times (zeros
        beyond the used segments).  None if the QP is not solved.  Any loss of the optimal coefficients can be
        differentiated; for the reference's own objc loss use `implicit_time_grad`, which needs no extra solve."""
        seg = qp_traj.seg
        M = max(p.shape[0] for p in qp_traj.hpolys)
        hp = np.zeros((1, seg, M, 4))
        for i, p in enumerate(qp_traj.hpolys):
            hp[0, i, :p.shape[0]] = p
        ini = qp_traj.start_state.reshape(1, 3, 3); fin = qp_traj.end_state.reshape(1, 3, 3)
        T = np.asarray(qp_traj.Times[:seg], dtype=np.float64)[None]
        out = _qp.qp_solve_vjp(qp_traj.order, ini, fin, hp, T, np.asarray(grad_z, dtype=np.float64).reshape(1, seg, 3, -1),
                               res=qp_traj.res, max_vel=qp_traj._limits[0], max_acc=qp_traj._limits[1],
                               settings=_qp.qp_settings(method=_qp.QP_METHOD_INTERIOR_POINT), ctx=self._ctx)
        if out["status"][0] != 1:
            return None
        g = np.zeros_like(np.asarray(qp_traj.Times, dtype=np.float64))
        g[:seg] = out["grad_T"][0]
        return g

    def backward_batch(self, qp_trajs, grad_zs):
        """`backward` for a minibatch: one anet_qp_solve_vjp per (order, segment count, res, limits) group, as
        forward_batch groups its solves.  grad_zs[i] may be None (sample skipped).  Returns a list of d loss / d Times
        arrays (None where skipped or unsolved)."""
        n = len(qp_trajs)
        out = [None] * n
        groups = {}
        for idx, q in enumerate(qp_trajs):
            if grad_zs[idx] is not None:
                groups.setdefault((q.order, q.seg, q.res, q._limits[0], q._limits[1]), []).append(idx)
        st = _qp.qp_settings(method=_qp.QP_METHOD_INTERIOR_POINT)
        for (order, seg, res, vmax, amax), ids in groups.items():
            M = max(max(p.shape[0] for p in qp_trajs[i].hpolys) for i in ids)
            B = len(ids)
            hp = np.zeros((B, seg, M, 4)); ini = np.zeros((B, 3, 3)); fin = np.zeros((B, 3, 3)); T = np.zeros((B, seg))
            gz = np.zeros((B, seg, 3, 2 * order))
            for r, i in enumerate(ids):
                q = qp_trajs[i]
                for k, pl in enumerate(q.hpolys):
                    hp[r, k, :pl.shape[0]] = pl
                ini[r] = q.start_state.reshape(3, 3); fin[r] = q.end_state.reshape(3, 3); T[r] = q.Times[:seg]
                gz[r] = np.asarray(grad_zs[i], dtype=np.float64).reshape(seg, 3, 2 * order)
            res_ = _qp.qp_solve_vjp(order, ini, fin, hp, T, gz, res=res, max_vel=vmax, max_acc=amax, settings=st, ctx=self._ctx)
            for r, i in enumerate(ids):
                if res_["status"][r] == 1:
                    g = np.zeros_like(np.asarray(qp_trajs[i].Times, dtype=np.float64))
                    g[:seg] = res_["grad_T"][r]
                    out[i] = g
        return out

    def forward_batch(self, qp_trajs):
        """Extension: the minibatch of the training loop in ONE solve per (order, segment count, res, limits) group
        instead of one `forward` call per sample (minsnap_network_conv_lstm.py:340-352 loops in Python).
        Returns a list with the 5-tuple `forward` returns for each sample, and lists `time_grads`,
        `implicit_time_grads` (None where the QP was not solved)."""
        n = len(qp_trajs)
        results, tg, itg = [None] * n, [None] * n, [None] * n
        groups = {}
        for idx, q in enumerate(qp_trajs):
            groups.setdefault((q.order, q.seg, q.res, q._limits[0], q._limits[1]), []).append(idx)
        for (order, seg, res, vmax, amax), ids in groups.items():
            M = max(max(p.shape[0] for p in qp_trajs[i].hpolys) for i in ids)
            B = len(ids)
            hp = np.zeros((B, seg, M, 4)); ini = np.zeros((B, 3, 3)); fin = np.zeros((B, 3, 3)); T = np.zeros((B, seg))
            for r, i in enumerate(ids):
                q = qp_trajs[i]
                for k, pl in enumerate(q.hpolys):
                    hp[r, k, :pl.shape[0]] = pl
                ini[r] = q.start_state.reshape(3, 3); fin[r] = q.end_state.reshape(3, 3); T[r] = q.Times[:seg]
            out = _qp.qp_solve(order, ini, fin, hp, T, res=res, max_vel=vmax, max_acc=amax, settings=self._settings,
                               time_grad=True, ctx=self._ctx)
            eff = traj_cost_grad_T(out["coeffs"], T, m34=1400.0, ctx=self._ctx)
            for r, i in enumerate(ids):
                q = qp_trajs[i]
                Times = q.Times
                obj1 = float(np.sum(Times[:seg]) / (1.0 * seg))
                pad = float(np.mean(Times[seg:] ** 2)) if 5 - seg != 0 else 0.0
                if out["status"][r] != 1:
                    objt = None
                    if hasattr(q, "ref_time_factor"):
                        objt = float(np.mean((Times[:seg] - q.ref_time_factor[:seg]) ** 2) / seg + pad)
                    results[i] = (None, obj1, objt, None, pad)
                    continue
                results[i] = (out["coeffs"][r].reshape(-1), obj1, None, float(out["obj"][r] / q.path_length), pad)
                tg[i] = np.zeros_like(Times); tg[i][:seg] = eff[r] / q.path_length
                itg[i] = np.zeros_like(Times); itg[i][:seg] = out["grad_T"][r] / q.path_length
        return results, tg, itg

    def forward4lstm(self, qp_traj, pred_stop_tokens, seq_len=5):
        segments = qp_traj.seg
        Times = qp_traj.Times
        out, T = self._solve(qp_traj)
        curr_obj1_val = float(np.sum(Times[:segments]) / (1.0 * segments))
        pred = np.asarray(pred_stop_tokens, dtype=np.float64)
        gt = np.concatenate([np.zeros(segments - 1), np.ones(seq_len - segments + 1)])
        end_penalty, thresh = 5.0, 0.42                                     # layers.py:190-191
        premature = float(np.sum((pred > thresh) & (gt < thresh)) * end_penalty)
        late = float(np.sum((pred < thresh) & (gt > thresh)) * end_penalty)
        with np.errstate(divide="ignore"):               # nn.BCELoss: mean reduction, each log clamped at -100
            bce = float(-np.mean(gt * np.maximum(np.log(pred), -100.0) + (1 - gt) * np.maximum(np.log(1 - pred), -100.0)))
        stop_token_loss = bce + premature + late
        if out["status"][0] != 1:
            curr_objt_val = None
            if hasattr(qp_traj, "ref_time_factor"):
                curr_objt_val = float(np.mean((Times[:segments] - qp_traj.ref_time_factor[:segments]) ** 2) / segments)
            self.time_grad = None
            self.implicit_time_grad = None
            return None, curr_obj1_val, curr_objt_val, None, stop_token_loss
        z = out["coeffs"][0].reshape(-1)
        curr_objc_val = float(out["obj"][0] / qp_traj.path_length)
        g = traj_cost_grad_T(out["coeffs"], T, m34=1400.0, ctx=self._ctx)[0] / qp_traj.path_length
        self.time_grad = np.zeros_like(Times)
        self.time_grad[:segments] = g
        self.implicit_time_grad = np.zeros_like(Times)
        self.implicit_time_grad[:segments] = out["grad_T"][0] / qp_traj.path_length
        return z, curr_obj1_val, None, curr_objc_val, stop_token_loss
