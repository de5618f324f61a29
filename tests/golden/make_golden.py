#!/usr/bin/env python3
"""Generate golden fixtures by IMPORTING the reference's Python QP twin.

Runs only in the build container (needs /root/reference).  Nothing from the
reference is copied: this script imports `network/utils/min_traj_opt.py`
(MinTrajOpt.update -> Q,A,b,G1,h1,G2,h2; reference lines :68-178, :377-697) and
`network/utils/trajectory.py` (Trajectory.get_pos/get_vel/get_acc, :47-98), feeds them
seeded inputs and stores inputs + outputs as .npz data.

The reference module imports `cvxpy`, `osqp` and `memory_profiler` at module top
(min_traj_opt.py:3,12,16).  None is installed here and none is touched by the assembly
code path, so inert placeholder modules are registered for the import to succeed.  They
carry no arithmetic.

Extra derived vectors (computed with numpy from the REFERENCE-ASSEMBLED matrices, so they
pin the solve against the reference's own formulation):
  z_eq / e_eq        : solution + 0.5 z'Qz of the equality-constrained QP (reference Q: m_34=1400)
  z_wp_* / e_wp_*    : same with waypoint rows appended (the rows the reference keeps commented
                       out at min_traj_opt.py:434-437) and Q patched to the true integral
                       (1440) -> this is what a MINCO solve must reproduce.
                       *_c3: only p,v,a fixed at the ends (reference convention)
                       *_cs: p,v,a,(j) fixed at the ends (MINCO convention; extra rows appended)
"""
import os, sys, types, io, contextlib
import numpy as np

REF = "/root/reference/network"
OUT = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    for name in ("cvxpy", "osqp", "memory_profiler"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            if name == "memory_profiler":
                m.profile = lambda f: f
            if name == "osqp":
                m.OSQP = object
            sys.modules[name] = m
    sys.path.insert(0, REF)
    import torch  # noqa
    from utils.min_traj_opt import MinTrajOpt
    from utils.trajectory import Trajectory
    return MinTrajOpt, Trajectory


def make_params(order, res, vmax=5.0, amax=7.0, vmax1=5.0, amax1=8.0, use_time_factor=False):
    return {
        "physical_limits": {"max_vel": vmax, "max_acc": amax, "max_jerk": 12.0},
        "phase1_physical_limits": {"max_vel": vmax1, "max_acc": amax1, "max_jerk": 10.0,
                                   "inf_dis": 0.1},
        "planning": {"order": order, "state_dim": 3, "dim": 3, "res": res, "seg": 5,
                     "var_num": 120, "use_time_factor": use_time_factor},
    }


def synth_problem(rng, N, M_max=12, rest=True):
    """Random-walk waypoints + box-ish corridors in a.x<=b form (normalised rows)."""
    pts = [np.array([0.0, 0.0, 1.0])]
    for _ in range(N):
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        p = pts[-1] + d * rng.uniform(1.0, 3.0)
        p[2] = min(max(p[2], 0.0), 5.0)
        pts.append(p)
    pts = np.array(pts)
    T = rng.uniform(0.5, 2.0, size=N)
    state = np.zeros((9, 2))
    state[0::3, 0] = pts[0]
    state[0::3, 1] = pts[-1]
    if not rest:
        state[1::3, 0] = rng.normal(size=3) * 0.5
        state[2::3, 0] = rng.normal(size=3) * 0.3
        state[1::3, 1] = rng.normal(size=3) * 0.5
        state[2::3, 1] = rng.normal(size=3) * 0.3
    hpolys = np.zeros((50, 4, N))
    for i in range(N):
        lo = np.minimum(pts[i], pts[i + 1]) - rng.uniform(0.5, 3.0, size=3)
        hi = np.maximum(pts[i], pts[i + 1]) + rng.uniform(0.5, 3.0, size=3)
        rows = []
        for ax in range(3):
            e = np.zeros(3); e[ax] = 1.0
            rows.append(np.r_[e, hi[ax]])
            rows.append(np.r_[-e, -lo[ax]])
        k = int(rng.integers(0, M_max - 6 + 1))
        mid = 0.5 * (pts[i] + pts[i + 1])
        for _ in range(k):
            a = rng.normal(size=3); a /= np.linalg.norm(a)
            rows.append(np.r_[a, a @ mid + rng.uniform(1.0, 3.0)])
        rows = np.array(rows)
        hpolys[: rows.shape[0], :, i] = rows
    return state, hpolys, T, pts


def kkt_solve(Q, A, b):
    n, m = Q.shape[0], A.shape[0]
    K = np.block([[Q, A.T], [A, np.zeros((m, m))]])
    rhs = np.r_[np.zeros(n), b]
    sol = np.linalg.solve(K, rhs)
    z = sol[:n]
    res = np.linalg.norm(K @ sol - rhs, np.inf)
    return z, 0.5 * z @ Q @ z, res, np.linalg.cond(K)


def main():
    import torch
    MinTrajOpt, Trajectory = _import_reference()
    cases = [
        # name, order, N, res, seed, rest, phase
        ("snap_n5_r20", 4, 5, 20, 11, True, 2),
        ("snap_n8_r5", 4, 8, 5, 12, False, 2),
        ("snap_n8_r20", 4, 8, 20, 13, True, 1),
        ("jerk_n5_r20", 3, 5, 20, 14, False, 2),
        ("jerk_n16_r4", 3, 16, 4, 15, True, 1),
        ("snap_n1_r4", 4, 1, 4, 16, False, 2),
        ("jerk_n2_r3", 3, 2, 3, 17, False, 2),
    ]
    for name, order, N, res, seed, rest, phase in cases:
        rng = np.random.default_rng(seed)
        state, hpolys, T, pts = synth_problem(rng, N, rest=rest)
        D = 2 * order
        opt = MinTrajOpt(make_params(order, res))
        with contextlib.redirect_stdout(io.StringIO()):
            opt.update(torch.tensor(state), torch.tensor(hpolys), torch.tensor(T),
                       phase=phase, seq_len=N)
        Q, A, b, G1, h1, G2, h2 = [p.detach().numpy().astype(np.float64) for p in opt.params]
        assert opt.seg == N
        n = 3 * D * N
        assert Q.shape == (n, n) and A.shape[1] == n
        # compact inequality storage: per row keep only the piece's 3*D columns
        m_rows = [int(np.sum(np.linalg.norm(hpolys[:, :, i], axis=1) > 0)) for i in range(N)]
        G1c = np.zeros((G1.shape[0], 3 * D)); G2c = np.zeros((G2.shape[0], D))
        r = 0
        for i in range(N):
            for _ in range(res):
                blk = G1[r:r + m_rows[i]]
                G1c[r:r + m_rows[i]] = blk[:, i * 3 * D:(i + 1) * 3 * D]
                z = blk.copy(); z[:, i * 3 * D:(i + 1) * 3 * D] = 0
                assert not z.any()
                r += m_rows[i]
        assert r == G1.shape[0]
        r = 0
        for i in range(N):
            for _ in range(res):
                for j in range(3):
                    blk = G2[r:r + 4]
                    c0 = i * 3 * D + j * D
                    G2c[r:r + 4] = blk[:, c0:c0 + D]
                    z = blk.copy(); z[:, c0:c0 + D] = 0
                    assert not z.any()
                    r += 4
        assert r == G2.shape[0]

        out = dict(order=order, N=N, res=res, phase=phase, state=state,
                   hpolys=hpolys[:16], m_rows=np.array(m_rows), T=T, pts=pts,
                   Q=Q, A=A, b=b, G1c=G1c, h1=h1, G2c=G2c, h2=h2,
                   G1_sum=G1.sum(), G2_sum=G2.sum(),
                   path_length=float(opt.path_length))
        assert not hpolys[16:].any()

        # (1) equality-constrained QP with the reference's own Q (1400)
        z, e, resid, cond = kkt_solve(Q, A, b)
        out.update(z_eq=z, e_eq=e, kkt_resid=resid, kkt_cond=cond)

        # (2) waypoint-pinned problems on the reference-assembled A,b (Q -> true integral)
        Q2 = Q.copy()
        if order == 4:
            for blk in range(3 * N):
                c0 = blk * D
                Q2[c0 + 2, c0 + 3] *= 1440.0 / 1400.0
                Q2[c0 + 3, c0 + 2] *= 1440.0 / 1400.0
        rows, rhs = [], []
        for i in range(1, N):          # start position of piece i == waypoint i
            for ax in range(3):
                rw = np.zeros(n); rw[i * 3 * D + ax * D + D - 1] = 1.0
                rows.append(rw); rhs.append(pts[i, ax])
        Aw = np.vstack([A] + rows) if rows else A
        bw = np.r_[b, rhs] if rows else b
        z3, e3, r3, _ = kkt_solve(Q2, Aw, bw)
        out.update(z_wp_c3=z3, e_wp_c3=e3, wp_resid_c3=r3)
        if order == 4:                 # MINCO convention: jerk also fixed (=0) at both ends
            rows2, rhs2 = [], []
            jh = rng.normal(size=3) * (0.0 if rest else 0.4)
            jt = rng.normal(size=3) * (0.0 if rest else 0.4)
            TN = T[-1]
            for ax in range(3):
                rw = np.zeros(n); rw[ax * D + 4] = 6.0
                rows2.append(rw); rhs2.append(jh[ax])
                rw = np.zeros(n); c0 = (N - 1) * 3 * D + ax * D
                rw[c0:c0 + 5] = [210 * TN**4, 120 * TN**3, 60 * TN**2, 24 * TN, 6.0]
                rows2.append(rw); rhs2.append(jt[ax])
            As = np.vstack([Aw] + rows2); bs = np.r_[bw, rhs2]
            zs, es, rs, _ = kkt_solve(Q2, As, bs)
            out.update(z_wp_cs=zs, e_wp_cs=es, wp_resid_cs=rs, jerk_head=jh, jerk_tail=jt)

        # (3) reference trajectory.py evaluation of z_eq
        coeffs = [z.reshape(N, 3, D)[i] for i in range(N)]
        with contextlib.redirect_stdout(io.StringIO()):
            traj = Trajectory(coeffs, list(T))
            ts = np.linspace(0.0, float(np.sum(T)) * 0.999, 17)
            pos = np.array([traj.get_pos(t) for t in ts])
            vel = np.array([traj.get_vel(t) for t in ts])
            acc = np.array([traj.get_acc(t) for t in ts])
        out.update(eval_t=ts, eval_pos=pos, eval_vel=vel, eval_acc=acc)

        # (4) the time gradient the reference's training back-propagates: d(1/2 z'Q(T)z)/dT with the
        #     solution z detached (layers.py:121,143-147), taken by autograd THROUGH THE REFERENCE'S OWN Q(T)
        Tt = torch.tensor(T, dtype=torch.float64, requires_grad=True)
        opt2 = MinTrajOpt(make_params(order, res))
        with contextlib.redirect_stdout(io.StringIO()):
            opt2.update(torch.tensor(state), torch.tensor(hpolys), Tt, phase=phase, seq_len=N)
        zt = torch.tensor(z)
        loss = 0.5 * zt @ opt2.params[0] @ zt
        (gT,) = torch.autograd.grad(loss, opt2.Times)
        out.update(dcost_dT=gT.detach().numpy())

        path = os.path.join(OUT, f"qp_{name}.npz")
        np.savez_compressed(path, **out)
        print(f"{name}: n={n} me={A.shape[0]} mg={G1.shape[0]}+{G2.shape[0]} "
              f"kkt_resid={resid:.1e} cond={cond:.1e} e_eq={e:.6g} e_wp_c3={e3:.6g} "
              f"size={os.path.getsize(path)/1024:.0f}KB")


def main_time_factor():
    """use_time_factor = True (min_traj_opt.py:113-136, 185-296; disabled in the reference's params.yaml but part of
    MinTrajOpt.update): waypoints from scipy's LP over consecutive polytopes, the float32 time lower bounds, Times =
    time_lb (1 + factor), ref_time_factor, the path length along the waypoints -- and the matrices assembled with them."""
    import torch
    MinTrajOpt, _ = _import_reference()
    for name, order, N, res, seed, phase in [("tf_snap_n3", 4, 3, 5, 31, 2), ("tf_jerk_n5", 3, 5, 4, 32, 1), ("tf_snap_n2", 4, 2, 4, 33, 2)]:
        rng = np.random.default_rng(seed)
        state, hpolys, T, pts = synth_problem(rng, N, rest=False)
        factor = np.zeros(5); factor[:N] = rng.uniform(0.2, 1.5, size=N)
        ref_times = np.zeros(5); ref_times[:N] = rng.uniform(0.8, 2.5, size=N)
        hp5 = np.zeros((50, 4, 5)); hp5[:, :, :N] = hpolys
        opt = MinTrajOpt(make_params(order, res, vmax=4.5, amax=7.0, use_time_factor=True))
        with contextlib.redirect_stdout(io.StringIO()):
            opt.update(torch.tensor(state), torch.tensor(hp5), torch.tensor(factor), phase=phase,
                       traj_times=torch.tensor(ref_times), seq_len=5)
        Q, A, b, G1, h1, G2, h2 = [p.detach().numpy().astype(np.float64) for p in opt.params]
        out = dict(order=order, N=N, res=res, phase=phase, state=state, hpolys=hp5[:16], factor=factor, ref_times=ref_times,
                   Times=opt.Times.detach().numpy().astype(np.float64), time_lb=opt.time_lb.detach().numpy().astype(np.float64),
                   waypts=np.asarray(opt.waypts, dtype=np.float64), path_length=float(opt.path_length),
                   ref_time_factor=opt.ref_time_factor.detach().numpy().astype(np.float64),
                   Q=Q, A=A, b=b, G1_sum=G1.sum(), h1=h1, G2_sum=G2.sum(), h2=h2)
        assert opt.seg == N and not hp5[16:].any()
        path = os.path.join(OUT, f"{name}.npz")
        np.savez_compressed(path, **out)
        print(f"{name}: Times={out['Times'][:N]} path_length={out['path_length']:.6f} size={os.path.getsize(path)/1024:.0f}KB")


def main_vjp():
    """Backward pass through the inequality QP (layers.py:120-147, 225-243): d loss / d Times for a seeded smooth loss of
    the optimal coefficients, loss(z) = w1.z + 1/2 sum w2 z^2.

    The reference installs the hook  grad <- -J^-1 grad  with
        J = [[Q, G'diag(lam), A'], [G, diag(Gz - h), 0], [A, 0, 0]]                     (layers.py:129-134)
    on y = (z, lam, nu).  J is the TRANSPOSE of the Jacobian dF/dy of the KKT residual
        F(y; T) = (Qz + G'lam + A'nu,  lam * (Gz - h),  Az - b),
    so  w = -J^-1 [dloss/dz; 0; 0]  is the adjoint vector and  d loss / d T = w' dF/dT  at fixed y (implicit function
    theorem: F(y*(T), T) = 0).  The reference stops at the detached leaf z; carried through, the contraction with dF/dT
    is taken here by torch.autograd THROUGH THE REFERENCE'S OWN Q(T), A(T), b(T), G(T), h(T) (MinTrajOpt.update with Times
    on the tape).  The optimum (z, lam, nu) of the reference-assembled matrices comes from the float64 interior-point
    oracle (oracle/qp_np.py), everything else is the reference's matrices and layers.py's J."""
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle.qp_np import qp_ipm
    MinTrajOpt, _ = _import_reference()
    for name, order, N, res, seed, phase in [("vjp_snap_n3", 4, 3, 6, 45, 2), ("vjp_jerk_n4", 3, 4, 5, 41, 1),
                                             ("vjp_snap_n2", 4, 2, 8, 41, 2), ("vjp_snap_n3b", 4, 3, 8, 50, 1)]:
        rng = np.random.default_rng(seed)
        state, hpolys, T, pts = synth_problem(rng, N, M_max=9, rest=False)
        seglen = np.linalg.norm(np.diff(pts, axis=0), axis=1)
        T = seglen / rng.uniform(2.0, 3.0, size=N)          # brisk enough for active limits, slow enough to be feasible
        D = 2 * order
        Tt = torch.tensor(T, dtype=torch.float64, requires_grad=True)
        opt = MinTrajOpt(make_params(order, res))
        with contextlib.redirect_stdout(io.StringIO()):
            opt.update(torch.tensor(state), torch.tensor(hpolys), Tt, phase=phase, seq_len=N)
        Qt, At, bt, G1t, h1t, G2t, h2t = opt.params
        Gt = torch.vstack([G1t, G2t]); ht = torch.hstack([h1t, h2t])          # layers.py:68-70
        Q, A, b, G, h = [x.detach().numpy().astype(np.float64) for x in (Qt, At, bt, Gt, ht)]
        z, lam, nu, obj, it = qp_ipm(Q, A, b, G, h, tol=1e-12, max_iter=300)
        assert it < 300, (name, "QP oracle did not converge")
        g = G @ z - h
        active = int((lam > 1e-6).sum())
        assert active >= 2, (name, active, "too few active inequalities: the fixture would not exercise the inequality rows")
        n, mg, me = Q.shape[0], G.shape[0], A.shape[0]
        w1 = rng.normal(size=n); w2 = rng.uniform(0.0, 1.0, size=n)
        gz = w1 + w2 * z
        J = np.block([[Q, G.T * lam[None, :], A.T],
                      [G, np.diag(g), np.zeros((mg, me))],
                      [A, np.zeros((me, mg + me))]])                                # layers.py:129-134
        rhs = np.r_[gz, np.zeros(mg + me)]
        w = -np.linalg.solve(J, rhs)                                                 # layers.py:139
        zt, lt, nt, wt = (torch.tensor(x) for x in (z, lam, nu, w))
        F = torch.hstack([Qt @ zt + Gt.T @ lt + At.T @ nt, lt * (Gt @ zt - ht), At @ zt - bt])
        (gT,) = torch.autograd.grad(wt @ F, opt.Times)
        m_rows = [int(np.sum(np.linalg.norm(hpolys[:, :, i], axis=1) > 0)) for i in range(N)]
        out = dict(order=order, N=N, res=res, phase=phase, state=state, hpolys=hpolys[:16], m_rows=np.array(m_rows), T=T,
                   w1=w1, w2=w2, z=z, lam_max=float(lam.max()), n_active=active, obj=float(obj), hook_grad_z=w[:n],
                   dloss_dT=gT.detach().numpy().astype(np.float64), kkt_cond=float(np.linalg.cond(J)))
        assert not hpolys[16:].any()
        path = os.path.join(OUT, f"{name}.npz")
        np.savez_compressed(path, **out)
        print(f"{name}: n={n} mg={mg} active={active} obj={obj:.6g} cond(J)={out['kkt_cond']:.1e} dloss_dT={out['dloss_dT']} "
              f"size={os.path.getsize(path)/1024:.0f}KB")


def main_layers():
    """The reference's OWN post-solve code executed: network/utils/learning/layers.py OsqpLayer.forward (:51-151) and
    forward4lstm (:153-247) are imported and run -- their loss terms (mean time, padding MSE, reference-time MSE, stop-token
    BCE + 5.0 penalties at 0.42, objc = 1/2 z'Qz / path_length), their KKT hook (:129-141) and torch.autograd's d/dTimes
    through the reference's Q(T) -- on qp_traj objects the reference's MinTrajOpt.update built.

    What is NOT the reference's: OSQP.  `osqp` is not in this image; the module's `osqp.OSQP` is replaced by an injector with
    OSQP's setup / solve interface that returns the optimum (x, y = [nu; lam]) of the matrices it is handed from the float64
    interior-point oracle (oracle/qp_np.py), or status 'maximum iterations reached' when told to fail (the unsolved branches).
    So these fixtures pin everything layers.py does AROUND the solve, at the exact optimum; they do not pin OSQP's iterates."""
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle.qp_np import qp_ipm
    MinTrajOpt, _ = _import_reference()
    import importlib
    layers = importlib.import_module("utils.learning.layers")

    class Injector:
        fail = False

        def setup(self, P, q, A, l, u, **kw):
            self.P = np.asarray(P.todense(), dtype=np.float64)
            A = np.asarray(A.todense(), dtype=np.float64)
            eq = np.isfinite(l)
            self.A, self.b, self.G, self.h = A[eq], u[eq], A[~eq], u[~eq]
            self.n_eq = int(eq.sum())
            assert eq[:self.n_eq].all()                    # layers.py stacks the equality rows first

        def solve(self):
            info = types.SimpleNamespace(status="maximum iterations reached" if Injector.fail else "solved")
            if Injector.fail:
                return types.SimpleNamespace(x=None, y=None, info=info)
            z, lam, nu, obj, it = qp_ipm(self.P, self.A, self.b, self.G, self.h, tol=1e-12, max_iter=300)
            assert it < 300
            return types.SimpleNamespace(x=z, y=np.r_[nu, lam], info=info)
    layers.osqp.OSQP = Injector
    cases = [("layers_snap_n3", 4, 3, 6, 45, 2, [0.1, 0.2, 0.9, 0.95, 0.99]),
             ("layers_jerk_n4", 3, 4, 5, 41, 1, [0.5, 0.2, 0.9, 0.3, 0.99]),
             ("layers_snap_n5", 4, 5, 4, 52, 2, [0.43, 0.9, 0.41, 0.1, 0.2])]
    for name, order, N, res, seed, phase, pred in cases:
        rng = np.random.default_rng(seed)
        state, hpolys, T, pts = synth_problem(rng, N, M_max=9, rest=False)
        seglen = np.linalg.norm(np.diff(pts, axis=0), axis=1)
        T5 = np.zeros(5); T5[:N] = seglen / rng.uniform(2.0, 3.0, size=N)
        T5[N:] = rng.uniform(0.1, 0.6, size=5 - N)           # what the network emits for unused segments: the padding loss's input
        ref5 = np.zeros(5); ref5[:N] = T5[:N] * rng.uniform(0.8, 1.3, size=N)
        hp5 = np.zeros((50, 4, 5)); hp5[:, :, :N] = hpolys
        out = dict(order=order, N=N, res=res, phase=phase, state=state, hpolys=hp5[:16], Times=T5, ref_times=ref5,
                   pred_stop_tokens=np.array(pred))
        assert not hp5[16:].any()
        for mode in ("forward", "forward4lstm"):
            for fail in (False, True):
                Injector.fail = fail
                Tt = torch.tensor(T5, dtype=torch.float64, requires_grad=True)
                opt = MinTrajOpt(make_params(order, res))
                layer = layers.OsqpLayer()
                with contextlib.redirect_stdout(io.StringIO()):
                    opt.update(torch.tensor(state), torch.tensor(hp5), Tt, phase=phase, traj_times=torch.tensor(ref5), seq_len=5)
                    if mode == "forward":
                        r = layer.forward(opt)
                    else:
                        r = layer.forward4lstm(opt, torch.tensor(pred, dtype=torch.float32), seq_len=5)
                z, obj1, objt, objc, last = r
                key = f"{mode}_{'unsolved' if fail else 'solved'}"
                assert opt.seg == N
                out[key + "_obj1"] = float(obj1)
                out[key + "_last"] = float(last)              # padding loss (forward) / stop-token loss (forward4lstm)
                (g1,) = torch.autograd.grad(obj1, Tt, retain_graph=True)
                out[key + "_dobj1_dT"] = g1.numpy().astype(np.float64)
                if fail:
                    assert z is None and objc is None
                    out[key + "_objt"] = float(objt)
                    (gt,) = torch.autograd.grad(objt, Tt)
                    out[key + "_dobjt_dT"] = gt.numpy().astype(np.float64)
                else:
                    assert objt is None
                    out[key + "_objc"] = float(objc)
                    out[key + "_z"] = z.detach().numpy().astype(np.float64)
                    # the leaf z behind hstack((z, lam, nu))[0:n]: its .grad after backward is what the reference's hook leaves
                    leaf = z.grad_fn.next_functions[0][0].next_functions[0][0].variable
                    objc.backward()
                    out[key + "_dobjc_dT"] = Tt.grad.numpy().astype(np.float64)
                    out[key + "_hook_grad_z"] = leaf.grad.numpy().astype(np.float64)
                    out["path_length"] = float(opt.path_length)
        if "forward_solved_last" in out and N < 5:
            (gp,) = torch.autograd.grad(torch.nn.MSELoss()(Tt[N:], torch.zeros(5 - N, dtype=torch.float64)), Tt)
            out["dpadding_dT"] = gp.numpy().astype(np.float64)
        path = os.path.join(OUT, f"{name}.npz")
        np.savez_compressed(path, **out)
        print(f"{name}: objc={out['forward_solved_objc']:.6g} obj1={out['forward_solved_obj1']:.6g} pad={out['forward_solved_last']:.6g} "
              f"stop={out['forward4lstm_solved_last']:.6g} objt(unsolved)={out['forward_unsolved_objt']:.6g} / "
              f"{out['forward4lstm_unsolved_objt']:.6g} dobjc_dT={out['forward_solved_dobjc_dT']} size={os.path.getsize(path)/1024:.0f}KB")


if __name__ == "__main__":
    if "--only-layers" in sys.argv:
        main_layers()
        sys.exit(0)
    if "--only-vjp" not in sys.argv:
        main()
        main_time_factor()
    main_vjp()
    main_layers()
